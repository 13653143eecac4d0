// acx_ppm_stream4.hip — k_ppm_stream4: the position-parallel `iter` scan (automaton_search_iter_next,
// src/AutomatonSearchIter.c:243-300; what it reports at a position: automaton_build_output, :157-197) for the
// batch shape BASELINE.json's config 2 names: haystacks of one fixed length (a stride below 2048) over a FOUR-LETTER
// alphabet that a shift tells apart (ACGT: symbol = (byte >> 1) & 3), image with C = 9, F = 10 (any dictionary of
// 2-bit symbols with keys of ten letters or more) and keys of at most 33 letters.  Everything else stays with
// k_ppm_stream (acx_ppm_kernels.hip), which this kernel is a specialisation of: the same tiles, the same queue and
// rounds, the same record streams, grants and block sums (k_ppm_gather_pos moves its records as it moves those of
// k_ppm_stream).  What is different, and why (profiles/r4_*): the kernel is bound by instruction issue — every
// instruction of every kind costs its SIMD an issue slot — so
//   * the geometry is constant: C, F, the halo, the LDS layout (2 KiB per wave, so that an LDS address is an OR) and
//     the tile size are compile-time numbers, the kernel reads a dozen arguments instead of fifty (16-18 SGPRs still spill to lanes
//     of a VGPR: derived scalars, read back mostly on rare paths — tools/resource_usage.py);
//   * the hot cell is `hot4` (include/acx_blob.h): its second word is the VALUE of the shallowest key that ends here —
//     what nearly every candidate that is no false alarm needs — and the id of the depth-C node only where no key
//     ends; one 16-bit field says "go deeper" for the next two symbols.  The top level of a slot is a dozen
//     instructions (k_ppm_stream: thirty, and a second gather for a third of the matching positions);
//   * a queue entry is the position + 33, the number its window, its offset in the haystack and its global position
//     are all one add away from;
//   * the haystack of a tile is staged by straight-line code (whole tiles: two 16-byte loads per lane, no bounds
//     checks; the last tile of a batch: a copy of the loop body that checks).
//   * round 6: the waits for loads are where the loads are needed (profiles/r6_s4_issue_budget.md §3: the compiler's placement had put them
//     behind the record stores and in front of the filter), the filter's neighbour word comes over the DPP network.
// Integer only, no MFMA: there is no contraction on this path.
#include "acx_kernels.h"
#include "acx_ppm_layout.h"

#define PPM_GRANT 1024u            // records a wave takes from the scratch pool at a time (as k_ppm_stream)
#define PPM_MAX_GRANTS 16u
#define PPM_DESC_WORDS 40u         // per wave: total, n_grants, 16 x base, 16 x count (+ pad)

#include "acx_ppm_device.h"

// development builds: -DACX_S4_EXP=<bits> switches parts off FOR TIMING ONLY (results are wrong): 1 hot cells from one
// line (no L2 latency), 2 no deeper walks, 4 no record stores, 8 no rounds, 16 no queue
#ifndef ACX_S4_EXP
#define ACX_S4_EXP 0
#endif
// -DACX_S4_PHASES: every wave adds up the clock ticks (s_memtime, 100 MHz) it spends in the five steps of a trip of the loop
// (acx_ppm_args.phase_out: 8 sums over all waves; read and printed by scan_ppm under ACX_PPM_PHASES in -DACX_TUNING builds)
#ifdef ACX_S4_PHASES
#define S4_PH(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tph; tph = t_; } while (0)
#else
#define S4_PH(i) do { } while (0)
#endif
// -DACX_S4_TSEG=<16 a + b>: every wave adds up the clock ticks (s_memtime, 100 MHz) between points a and b of a trip of the loop —
// ONE pair of clock reads per trip, waited for at the trip's end, so that the kernel runs at (nearly) its own speed; a build per segment
// (tools/r6_phase_profile.sh).  phase_out[0]: ticks, [1]: trips that passed both points, [7]: waves.
#ifdef ACX_S4_TSEG
// (the clock value goes straight into a VGPR: the kernel has no scalar registers to spare — a live SGPR pair more and it spills to scratch)
#define S4_TP(k) do { if ((ACX_S4_TSEG >> 4) == (k)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); tp_v = (uint32_t)t_; tp_have = 1u; } \
                      if ((ACX_S4_TSEG & 15) == (k) && tp_have == 1u) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); tp_sum_v += (uint32_t)t_ - tp_v; tp_n_v += 1u; tp_have = 0u; } } while (0)
#else
#define S4_TP(k) do { } while (0)
#endif
#ifdef ACX_S4_MARK
#define S4_MARK(x) asm volatile("; MARK " #x)
#else
#define S4_MARK(x) do { } while (0)
#endif

namespace {

constexpr uint32_t S4_C = 9, S4_F = 10;                // symbols of a hot cell's code, of the filter's
constexpr uint32_t S4_HP = 32;                         // halo: staged positions in front of a tile (>= longest - 1)
constexpr uint32_t S4_TPOS = 2048;                     // positions per tile: 32 per lane
constexpr int      S4_NE = 6;                          // queue entries per lane and round
constexpr uint32_t S4_QCAP = 64u * S4_NE;              // entries of a round
constexpr uint32_t S4_G_BYTES = 4u << (2 * S4_F - 5);  // the filter bitmap: 4^F bits = 128 KiB, at LDS address 0
constexpr uint32_t S4_WAVE_BYTES = 2048;               // LDS of one wave
// byte offsets inside a wave's LDS.  Two symbol buffers: the round that is worked on may belong to the tile before the
// one whose candidates are being fetched.  A buffer: words 0, 1: pad; 2, 3: halo (32 positions); 4 .. 131: the tile; 132: pad
constexpr uint32_t S4_SYMB = 528;                      // bytes from one symbol buffer to the next (word 0 of the second, never read, is word 132 of the first, read for nothing)
constexpr uint32_t S4_QUEUE = 1064;                    // uint16 entries; the hand-over of the deeper walks lives in the same memory
// one bit per staged position: a byte of no key (word 0: halo, 1 .. 64: tile).  At the END of the wave's LDS: a tile without
// such bytes around (the usual one) does not look at them, and its queue may grow into them — 384 entries, a whole round;
// with them the queue ends where they begin (362 entries: a tile that holds more takes two rounds)
constexpr uint32_t S4_OBITS = S4_WAVE_BYTES - 260;
constexpr uint32_t S4_QCAP_OTHER = (S4_OBITS - S4_QUEUE) / 2;
static_assert(S4_QUEUE % 8 == 0 && S4_QUEUE >= S4_SYMB + 133 * 4 && S4_QUEUE + 2 * S4_QCAP <= S4_WAVE_BYTES && S4_QUEUE + 648 <= S4_OBITS &&
              S4_QCAP_OTHER >= 256 && S4_G_BYTES + 16 * S4_WAVE_BYTES <= ACX_PPM_LDS_BYTES, "LDS plan");

// LDS by byte address (the kernel's dynamic LDS starts at address 0: it has no static LDS).  A generic pointer built from
// `smem` costs an add of the array's base — zero, but a link-time zero the compiler does not fold — per access: one
// instruction per probe of the filter.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"     // (the HOST pass sees 64-bit pointers here; LDS pointers of the device are 32 bits)
__device__ __forceinline__ uint32_t lds_rd32(uint32_t byte_addr) { return *(const lds_u32_t*)byte_addr; }
__device__ __forceinline__ uint32_t lds_rd16(uint32_t byte_addr) { return *(const lds_u16_t*)byte_addr; }
__device__ __forceinline__ void lds_wr16(uint32_t byte_addr, uint32_t v) { *(lds_u16_t*)byte_addr = (uint16_t)v; }
#pragma clang diagnostic pop

__device__ __forceinline__ uint32_t top_base4(uint32_t d) { return 0x55555555u & ((1u << (2u * d)) - 1u); }   // (4^d - 1) / 3: top_base[d] of a four-symbol image

// H12: the image's hot4 cells are 12 bytes — { eowmask | go << 16, value of the shallowest key (0: none), deep id of the depth-C node } —
// (include/acx_blob.h; the dictionaries of iter_long, acx_long.cpp: there nearly every cell that sends a walk deeper also ends a key, so
// with 8-byte cells nearly every walker waits for cid[] first — one more round trip to the L2 in the chain of every pass)
// a slot's hot cell in registers AS ITS LOAD FILLS THEM: two words, or — 12-byte cells — three (a third word kept in a register of its own
// is a copy of a loaded register, i.e. a wait for the gather in the trip that issues it: tools/s4_waits.sh)
typedef uint32_t u32x3a __attribute__((ext_vector_type(3), aligned(4)));
template <bool H12> struct S4Cell { typedef u32x2 type; };
template <> struct S4Cell<true> { typedef u32x3a type; };
__device__ __forceinline__ uint32_t s4_cell_id(const u32x2& c) { return c.y; }      // (8-byte cells: the second word is the id where it is no value)
__device__ __forceinline__ uint32_t s4_cell_id(const u32x3a& c) { return c.z; }

// OFFS: the batch is an offsets batch (acx_ppm_args.off) instead of a fixed stride.  Then the symbols a key may use end at the START of the
// position's haystack, and the starts arrive as a bitmap (start_bits: bit p = a haystack starts at byte p; one word per lane and tile, loaded
// with the tile's bytes) that shares a word array with the bytes of no key: a bit at position x of `obits` says "what a key that ends at or
// behind x may use begins at x" — a start at x, or a byte of no key at x - 1 — and the limit of an entry is its distance to the last such bit
// (other_limit; fixed strides keep the cheaper form: a multiply for the offset in the haystack, the bits only where bytes of no key are
// around).  The records carry global positions as ever; k_ppm_gather_pos<true> finds their haystacks in the offsets.
template <bool H12, bool OFFS>
__global__ void __launch_bounds__(ACX_PPM_BLOCK) k_ppm_stream4(const acx_ppm_args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    {
        const u32x4* g4 = (const u32x4*)a.g;
        u32x4* s4 = (u32x4*)smem;
#pragma unroll
        for (uint32_t i = 0; i < S4_G_BYTES / 16 / ACX_PPM_BLOCK; i++) s4[threadIdx.x + i * ACX_PPM_BLOCK] = g4[threadIdx.x + i * ACX_PPM_BLOCK];
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t* const lds = (uint8_t*)smem;
    const uint32_t wbase = S4_G_BYTES + wid * S4_WAVE_BYTES;          // this wave's LDS, as a byte address
    uint32_t* const obits = (uint32_t*)(lds + wbase + S4_OBITS);
    uint32_t* const obits_tile = obits + 1;
    uint16_t* const queue = (uint16_t*)(lds + wbase + S4_QUEUE);
    Ppm<2, true, false> P(a);                                          // windows and symbols for the walks below the cells and for the rare general enumeration
    P.s_g = smem; P.s_map = nullptr; P.s_sym = (uint32_t*)(lds + wbase) + 1;
    P.T.q0 = S4_HP; P.T.halo = 0; P.T.idx_first = 0; P.T.ndw = 0; P.T.abase = nullptr; P.T.e0 = 0; P.T.npos = 0; P.has_other = 0;
    for (uint32_t i = lane; i < S4_WAVE_BYTES / 4; i += 64) ((uint32_t*)(lds + wbase))[i] = 0;

    // the batch: H bytes, cut into tiles; a wave takes a contiguous run of them (k_ppm_gather_pos counts on exactly this cut)
    const uint32_t stride = (uint32_t)a.stride, m24 = a.m24;
    uint32_t H;                                                        // (the launcher checks the size)
    if (OFFS) { const int64_t hb = a.off[a.n_hay]; H = (uint32_t)(hb < a.hay_cap ? hb : a.hay_cap); } else H = (uint32_t)(a.n_hay * a.stride - 1) + 1u;
    const uint32_t lo_pos = OFFS ? (uint32_t)a.off[0] : 0u;            // (bytes in front of the first haystack belong to none)
    const uint32_t n_tiles = OFFS ? (uint32_t)a.n_items : (uint32_t)(((int64_t)H + S4_TPOS - 1) / S4_TPOS);      // (offsets: the host's count — it sized the grid and the gather's runs by the buffer)
    const uint32_t n_waves = gridDim.x * ACX_PPM_WAVES;
    const uint32_t tpw = (n_tiles + n_waves - 1) / n_waves;
    const uint32_t wave_id = blockIdx.x * ACX_PPM_WAVES + wid;
#ifdef ACX_S4_WAVETIME             // development: when every wave starts and ends (100 MHz clock; phase_out: 2 x 4096 words), printed by scan_ppm
    const unsigned long long wt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (unequal runs: acx_ppm_slot_first_tile, acx_ppm_layout.h)
    const uint32_t blk_first = blockIdx.x * ACX_PPM_WAVES * tpw;
    const uint64_t t_begin = (uint64_t)blk_first + acx_ppm_slot_first_tile((uint32_t)wid, tpw, a.share_a, a.share_b);
    const uint32_t t_run = acx_ppm_slot_first_tile((uint32_t)wid + 1u, tpw, a.share_a, a.share_b) - acx_ppm_slot_first_tile((uint32_t)wid, tpw, a.share_a, a.share_b);
    uint32_t* const desc = a.wave_desc + (size_t)wave_id * PPM_DESC_WORDS;
    wave_sync();
    if (t_begin >= n_tiles) { if (lane == 0) { desc[0] = 0; desc[1] = 0; } return; }
    uint32_t tiles_left = n_tiles - (uint32_t)t_begin < t_run ? n_tiles - (uint32_t)t_begin : t_run;

    uint32_t ar_shift = a.sym_arith - 1u, ar_lut = a.sym_lut;
    asm volatile("" : "+s"(ar_shift), "+s"(ar_lut));                  // (kept in registers: the stage reads them for every dword)
    // which of the four bytes of a dword are none of the four letters (rare path)
    auto nib_of = [&](uint32_t w) -> uint32_t {
        const uint32_t x = (w >> ar_shift) & 0x03030303u;
        const uint32_t d = __builtin_amdgcn_perm(0u, ar_lut, x) ^ w;
        return ((d & 0xFFu) ? 1u : 0u) | ((d & 0xFF00u) ? 2u : 0u) | ((d & 0xFF0000u) ? 4u : 0u) | ((d >> 24) ? 8u : 0u);
    };
    // symbols that exist going back from staged position q when bytes of no key (OFFS: or haystack starts) are around.  A key has at most
    // 33 letters (acx_ppm_stream4_eligible): the word of q and the one in front of it hold every bit that can matter — no loop over words
    // (an offsets batch asks this of EVERY entry, and its last start lies a haystack's length back: the loop was a fifth of the kernel).
    // q >= 32 (a tile's positions start behind the halo word), so the word in front exists.
    auto other_limit = [&](uint32_t q) -> uint32_t {
        const uint32_t w = q >> 5;
        const uint32_t m1 = obits[w - 1u];
        const uint32_t m0 = obits[w] & (0xFFFFFFFFu >> (31u - (q & 31u)));
        const unsigned long long v = ((unsigned long long)m0 << 32) | m1;
        // the last bit at or in front of q: position 32 (w - 1) + 63 - clz; OFFS: that position is the first usable one, else the one behind it
        const uint32_t last = 32u * (w - 1u) + 63u - (uint32_t)__builtin_clzll(v | 1ull) + (OFFS ? 0u : 1u);
        return v ? q + 1u - last : 64u;
    };

    // ---- prologue: the halo of the run's first tile ---------------------------------------------------
    uint32_t e0 = (uint32_t)t_begin * S4_TPOS;
    uint32_t any_prev = 0;
    if (e0 > 0) {                                                      // (a multiple of 2048: 32 bytes in front of it exist)
        uint32_t nib = 0;
        if (lane < S4_HP / 4) {
            const uint32_t w = *(const uint32_t*)(a.hay + (e0 - S4_HP) + 4u * lane);
            const uint32_t x = (w >> ar_shift) & 0x03030303u;
            ((uint8_t*)(lds + wbase + 8))[lane] = (uint8_t)((x * 0x01041040u) >> 24);     // (the halo words of buffer 0)
            nib = nib_of(w);
            if (nib) atomicOr(&obits[0], nib << (4u * lane));
        }
        if (__any(nib != 0)) any_prev = 1;
    }
    uint32_t o_carry = 0;                                              // OFFS: which bytes of the 32 in front of the tile occur in no key (bit 31: the byte in front of the tile's first)
    if (OFFS) {
        wave_sync();
        if (e0 > 0) {
            o_carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)obits[0]);
            wave_sync();
            if (lane == 0) obits[0] = (o_carry << 1) | a.start_bits[(e0 >> 5) - 1u];
        }
        any_prev = 1;
    }
    uint32_t r_tile = 0;
    if (!OFFS) { uint32_t rr0; (void)div_magic(e0, a.stride_magic, stride, rr0); r_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)rr0); }
    const uint32_t step_r = OFFS ? 0u : S4_TPOS % stride;

    // a lane's 32 bytes of a tile (read once: they need not stay in the caches)
    u32x4 wn0 = {0, 0, 0, 0}, wn1 = {0, 0, 0, 0};                      // (two register quadruples, as the two loads fill them: tools/s4_waits.sh — eight scalars made the compiler copy a loaded register, i.e. wait for the load it had just issued)
    auto load_lane_full = [&](uint32_t b) {
        const u32x4a v0 = __builtin_nontemporal_load((const u32x4a*)(a.hay + b));
        const u32x4a v1 = __builtin_nontemporal_load((const u32x4a*)(a.hay + b + 16u));
        wn0 = v0; wn1 = v1;
    };
    auto load_lane = [&](uint32_t b) {                                 // the tile may end inside the buffer's last bytes
        if ((int64_t)b + 32 <= a.hay_cap) load_lane_full(b);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = (int64_t)b + 4 * j + 4 <= a.hay_cap ? *(const uint32_t*)(a.hay + b + 4u * j) : load_dw_tail(a.hay, a.hay_cap, b + 4u * j);
                if (j < 4) wn0[j] = w; else wn1[j - 4] = w;
            }
        }
    };
    // (a tile whose 2048 bytes lie inside the buffer: no checks)
    const uint32_t full_end = a.hay_cap >= (int64_t)S4_TPOS ? (uint32_t)((a.hay_cap < 0xFFFFFFFFll ? a.hay_cap : 0xFFFFFFFFll) - S4_TPOS) : 0u;
    const bool any_full = a.hay_cap >= (int64_t)S4_TPOS;
    uint32_t wsb = 0;                                                  // OFFS: the lane's word of the start bitmap, requested with the tile's bytes
    auto load_tile = [&](uint32_t e) {
        if (any_full && e <= full_end) load_lane_full(e + 32u * lane); else load_lane(e + 32u * lane);
        if (OFFS) wsb = a.start_bits[(e >> 5) + lane];
    };
    load_tile(e0);

    // the wave's record stream
    uint32_t run_off = 0;                                              // records so far
    uint32_t g_base = 0, g_size = 0, g_used = 0, ng = 0;               // current grant of the pool
    bool dead = false;                                                 // pool or grant list exhausted: keep counting, stop writing
    const uint32_t pool_x = blockIdx.x % a.n_pools;
    const uint32_t longest = a.longest;
    const uint32_t qaddr = wbase + S4_QUEUE + 2u * lane;               // this lane's entry of slot 0
    const uint32_t nm24 = 0u - m24;

    // ---- the pipeline -----------------------------------------------------------------------------------
    // A ROUND is at most 352 candidates of one tile (all of them, unless the tile holds more).  Every trip of the loop
    //   1. works off the round fetched one trip earlier (set O): top levels, deeper walks — every load that is waited for
    //      lies here;
    //   2. finishes it: placement, records (stores);
    //   3. turns the bytes of the next tile into symbols (registers only) when this trip hands over the last candidates
    //      of the current one, and requests the bytes of the tile after that;
    //   4. pushes the candidates of the next round into the queue and FETCHES: where they sit, their windows, the
    //      requests for their hot cells — into the same slot registers;
    //   5. stages the next tile: symbols -> the OTHER symbol buffer, the filter, the prefix sum of the pass words.
    // Loads return in order: a wait for a later load is a wait for the hot cells too.  So nothing between 4 and the next
    // trip's 1 waits for a load: the gathers of a round are in flight while the address unit of the CU serves them
    // (about one lane per ns and CU, profiles/r4_*) and the wave does step 5.  What a wave's time per round is made of
    // is the chain of its DEPENDENT round trips to the L2 (profiles/r4_experiments.md): every one taken out of the chain —
    // a value requested early and looked at late, an id that rides behind another gather — made the kernel faster,
    // instructions taken out did not.
    uint32_t cur = 0;                                                  // symbol buffer of the tile whose candidates are being fetched
#ifdef ACX_S4_TSEG
    uint32_t tp_v = 0, tp_sum_v = 0, tp_n_v = 0, tp_have = 0;
#endif
    bool tile_ok = true;
    uint32_t pw = 0, x_ex = 0, x_tot = 0, seg_lo = 0, use_other = 0, any_cur = 0;
    uint32_t Wc1 = 0, Wc2 = 0, anyo_c = 0;                             // the symbols of the tile that is being staged, which of its bytes occur in no key
    uint32_t Bc = 0;                                                   // OFFS: its word of `obits` (starts | bytes of no key, one position on)
    // ONE set of slot registers: step 1 is the last reader of a round's hot cells, step 3 loads the next round's into the
    // same registers (a second set would have to be copied into the first, and a copy of a loaded register is a wait)
    typedef typename S4Cell<H12>::type cell_t;
    cell_t hcO[S4_NE];                                                 // (H12: .z = the id of the depth-C node)
    uint32_t ppO[S4_NE];                                               // entry (position + 33) | symbols that exist << 12 | the seven symbols below the cell's << 18
    uint32_t nO = 0, nN = 0, cgO = 0, cgN = 0, symO = wbase, symN = wbase;
    bool haveO = false;
#pragma unroll
    for (int e = 0; e < S4_NE; e++) { hcO[e] = (cell_t)(0u); ppO[e] = 0; }

    // bytes of the tile in wn0, wn1 -> symbols (registers); the bytes of the tile at e_next are requested
    auto convert = [&](bool more, uint32_t e_next) {
        S4_MARK(M_STAGE);
        uint32_t diff = 0, pr[8];
        anyo_c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint32_t w = j < 4 ? wn0[j] : wn1[j - 4];
            const uint32_t x = (w >> ar_shift) & 0x03030303u;
            diff |= __builtin_amdgcn_perm(0u, ar_lut, x) ^ w;            // the letters permuted by the symbols give the bytes back iff all four are letters
            pr[j] = x * 0x01041040u;                                      // one multiply gathers the four 2-bit fields into the top byte
        }
        Wc1 = __builtin_amdgcn_perm(pr[1], pr[0], 0x0c0c0703u) | __builtin_amdgcn_perm(pr[3], pr[2], 0x07030c0cu);
        Wc2 = __builtin_amdgcn_perm(pr[5], pr[4], 0x0c0c0703u) | __builtin_amdgcn_perm(pr[7], pr[6], 0x07030c0cu);
        if (__any(diff != 0u)) {                                        // some byte of the tile is none of the four letters
#pragma unroll
            for (int j = 0; j < 8; j++) anyo_c |= nib_of(j < 4 ? wn0[j] : wn1[j - 4]) << (4 * j);
        }
        if (OFFS) {
            // the byte of no key in front of a lane's first position: the last bit of the lane before it (lane 0: of the tile before)
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp((int)o_carry, (int)anyo_c, 0x138, 0xF, 0xF, false);
            o_carry = (uint32_t)__builtin_amdgcn_readlane((int)anyo_c, 63);
            Bc = (anyo_c << 1) | (before >> 31) | wsb;
        }
        if (more) load_tile(e_next);
    };
    wave_sync();
    uint32_t halo_w = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_rd32(wbase + 3u * 4u));     // (the prologue wrote the halo words of buffer 0 in front of a wave_sync)
    // the tile at e0 (its symbols in Wc1, Wc2): symbols -> buffer `cur`, the filter, the prefix sum
    auto stage_rest = [&]() {
        uint32_t* const sym_tile = (uint32_t*)(lds + wbase + cur * S4_SYMB) + 4;
        const uint32_t left = H - e0;
        const uint32_t npos = left < S4_TPOS ? left : S4_TPOS;
        const uint32_t W1 = Wc1, W2 = Wc2, anyo = anyo_c;
        { u32x2 v; v.x = W1; v.y = W2; *(u32x2*)(sym_tile + 2u * lane) = v; }
        any_cur = OFFS ? 1u : (__any(anyo != 0) ? 1u : 0u);
        use_other = any_cur | any_prev;
        if (OFFS) obits_tile[lane] = Bc;
        else if (use_other) {
            obits_tile[lane] = anyo;
            if (!any_prev && lane == 0) obits[0] = 0;                    // (the tile before left no such bits, and a queue may have been there)
        }
        // the 16 symbols in front of the lane's own: the second word of the lane before it, over the DPP network (wave_shr:1; lane 0: the
        // last word of the tile before, kept in a scalar) — no LDS round trip in front of the filter
        const uint32_t W0 = (uint32_t)__builtin_amdgcn_update_dpp((int)halo_w, (int)W2, 0x138, 0xF, 0xF, false);
        halo_w = (uint32_t)__builtin_amdgcn_readlane((int)W2, 63);
        S4_TP(8);
        S4_MARK(M_FILTER);
        // the filter: every lane asks the bitmap about its own 32 positions, windows in registers (as k_ppm_stream)
        {
            constexpr uint32_t FB = 2 * S4_F, ush = 32u - FB, amask = ((1u << (FB - 5u)) - 1u) << 2;
            uint32_t U[5];
            U[0] = __builtin_amdgcn_alignbit(W1, W0, ush); U[1] = __builtin_amdgcn_alignbit(W2, W1, ush); U[2] = W2 >> ush; U[3] = 0; U[4] = 0;
            uint32_t acc = 0;
#pragma unroll
            for (int i0 = 0; i0 < 32; i0 += 16) {
                uint32_t gw[16], bs[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t b = 2u * (uint32_t)(i0 + i + 1), k = b >> 5, sh = b & 31u;
                    bs[i] = sh ? __builtin_amdgcn_alignbit(U[k + 1], U[k], sh) : U[k];
                    const uint32_t A = (bs[i] >> 3) & amask;
                    gw[i] = lds_rd32(A);
                }
                // ONE wait for the sixteen reads: all of them are operands of the (empty) statement below, so the compiler waits
                // for the last before it and for none after it (a wait per read is an instruction per read, and every
                // instruction is an issue slot)
                asm volatile("" : "+v"(gw[0]), "+v"(gw[1]), "+v"(gw[2]), "+v"(gw[3]), "+v"(gw[4]), "+v"(gw[5]), "+v"(gw[6]), "+v"(gw[7]),
                                  "+v"(gw[8]), "+v"(gw[9]), "+v"(gw[10]), "+v"(gw[11]), "+v"(gw[12]), "+v"(gw[13]), "+v"(gw[14]), "+v"(gw[15]));
#pragma unroll
                for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_alignbit(gw[i] >> (bs[i] & 31u), acc, 1u);
            }
            pw = acc;
        }
        if (use_other | (npos < S4_TPOS ? 1u : 0u)) {
            const uint32_t lp = 32u * lane;
            const uint32_t nv = npos > lp ? (npos - lp < 32u ? npos - lp : 32u) : 0u;
            pw &= (nv >= 32u ? 0xFFFFFFFFu : (1u << nv) - 1u) & ~anyo;      // (a byte of no key ends no key)
            if (OFFS && e0 < lo_pos) {                                   // (bytes in front of the first haystack)
                const uint32_t cut = lo_pos - e0;
                pw &= cut >= lp + 32u ? 0u : (cut > lp ? ~((1u << (cut - lp)) - 1u) : 0xFFFFFFFFu);
            }
        }
        S4_TP(9);
        S4_MARK(M_PREFIX);
        x_ex = wave_excl_scan((uint32_t)__popc(pw), x_tot);
        if (ACX_S4_EXP & 16) x_tot = 0;
        seg_lo = 0;
    };
    convert(tiles_left > 1, e0 + S4_TPOS);
    stage_rest();

#ifdef ACX_S4_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph = __builtin_amdgcn_s_memtime();
#endif
#ifdef ACX_S4_TSEG
    tp_have = 0;
#endif
#define S4_SLOTS(e, ...) _Pragma("unroll") for (int g_ = 0; g_ < S4_NE; g_ += 2) { if (g_ == 0 || (uint32_t)g_ < k) { _Pragma("unroll") for (int e = g_; e < g_ + 2; e++) { __VA_ARGS__ } } }
    for (;;) {
        S4_PH(7);
        S4_TP(0);
        // Every load of the trip before is waited for HERE — the hot cells of the round that step 1 looks at next (the youngest of them:
        // nothing is lost) and the bytes of the next tile.  Left to the compiler, which cannot know that a fetched round is always worked
        // off in the next trip, the waits land where those registers are next WRITTEN — behind the record stores of step 2 and the
        // loads of step 3, as waits for the stores' completion and for the loads just issued (tools/s4_waits.sh lists them).
        asm volatile("" : "+v"(wn0), "+v"(wn1));
        if (OFFS) asm volatile("" : "+v"(wsb));
#pragma unroll
        for (int e = 0; e < S4_NE; e++) asm volatile("" : "+v"(hcO[e]));
        // ---- 1. the round fetched one trip earlier: top levels, deeper walks -----------------------------------------------
        const uint32_t k = (nO + 63u) >> 6;                              // slots of set O that hold entries
        uint32_t cn[S4_NE];
        int32_t va[S4_NE];
        int32_t vb1 = 0; uint32_t vbe = S4_NE;                          // one second value per lane and round, with the slot it belongs to (as k_ppm_stream)
        auto set_vb = [&](uint32_t e, int32_t v) { if (vbe == (uint32_t)S4_NE) { vb1 = v; vbe = e; } };
#pragma unroll
        for (int e = 0; e < S4_NE; e++) { cn[e] = 0; va[e] = 0; }
        if (haveO) {
            P.s_sym = (uint32_t*)(lds + symO) + 1;
            S4_MARK(M_TOP);
            // top levels: how many keys end here (the cell holds the value of the shallowest), whether the walk goes deeper
            uint32_t n_go = 0, gomask = 0, cmax = 0;
            S4_SLOTS(e,
                const uint32_t hw = hcO[e].x, pk = ppO[e];
                const uint32_t L = __builtin_amdgcn_ubfe(pk, 12u, 6u);
                const uint32_t Lc = L < S4_C ? L : S4_C;
                const uint32_t m = hw & ((1u << Lc) - 1u);
                cn[e] = (uint32_t)__popc(m);
                // the walk below the cell: bit 16 + (next two symbols) — asked whatever L is: a walk that runs out of symbols ends at once
                // (a cell whose node has one child holds that child's unbranched path — up to seven symbols — instead of the 16 bits: the walk goes deeper
                //  iff the text agrees with all of them; three futile walkers in four came from such cells, the text agreeing with two symbols by chance)
                const uint32_t t14 = pk >> 18;
                const uint32_t g_many = __builtin_amdgcn_ubfe(hw, 16u + (t14 >> 10), 1u);
                const uint32_t g_one = ((t14 ^ (hw >> 16)) >> __builtin_amdgcn_ubfe(hw, 9u, 4u)) == 0u ? 1u : 0u;      // (bits 30, 31 of the cell's word are zero there)
                uint32_t g = (hw & 0x2000u) ? g_one : g_many;
                if (ACX_S4_EXP & 8192) g &= (((pk & 0xFFFu) * 2654435761u) >> 30) == 0u ? 1u : 0u;      // (timing only: three walkers in four dropped — what a go field over three symbols could give)
                gomask |= g << e; n_go += g;
                va[e] = (int32_t)hcO[e].y;
                cmax = cn[e] > cmax ? cn[e] : cmax;
            )
            S4_TP(1);
            S4_MARK(M_SECOND);
            // a second key within the cell's levels (0.6 % of the matching positions of config 2): the value of the first such
            // slot of the lane is REQUESTED here and looked at behind the deeper walks — a wait here would be one more
            // round trip to the L2 in the chain of every round
            uint32_t vbte = S4_NE, vbt_at = 0;
            int32_t vbt = 0;
            if (__any(cmax > 1u)) {
#pragma unroll
                for (int e = 0; e < S4_NE; e++) {
                    if (cn[e] > 1u && vbte == (uint32_t)S4_NE) {
                        const uint32_t L = __builtin_amdgcn_ubfe(ppO[e], 12u, 6u), Lc = L < S4_C ? L : S4_C;
                        const uint32_t m = hcO[e].x & ((1u << Lc) - 1u), m2 = m & (m - 1u);
                        const uint32_t d2 = (uint32_t)__ffs(m2);
                        const uint32_t X = P.window((ppO[e] & 0xFFFu) - 1u);
                        vbte = (uint32_t)e; vbt_at = top_base4(d2) + (X >> (32u - 2u * d2));
                    }
                }
                if (vbte < (uint32_t)S4_NE) vbt = a.top_val[vbt_at];
            }
            S4_MARK(M_DEEP);
            // deeper levels: the entries that go on, 64 at a time, one per lane; one 16-byte record per step, selects instead of
            // branches (as k_ppm_stream).  The hand-over: {entry | L << 12 | .. | eowmask << 23, second word of the cell}.
            uint32_t n_deep;
            const uint32_t d_base = wave_excl_scan(n_go, n_deep);
            if (ACX_S4_EXP & 3) n_deep = 0;
            uint32_t* const dq = (uint32_t*)queue;                       // [0..127] hand-over, then {first, second value}; [128..129] the dump slot; then 64 counts
            uint16_t* const dcnt = (uint16_t*)(dq + 130);
            for (uint32_t d0 = 0; d0 < n_deep; d0 += 64u) {
                {
                    uint32_t rnk = d_base - d0;                          // (unsigned: ranks below d0 wrap far beyond 64)
                    S4_SLOTS(e,
                        const uint32_t g = (gomask >> e) & 1u;
                        const uint32_t slot = (g != 0u && rnk < 64u) ? rnk : 64u;       // (slot 64: nobody reads it)
                        u32x2 v; v.x = (ppO[e] & 0x7FFFFFu) | (hcO[e].x << 23); v.y = s4_cell_id(hcO[e]);
                        *(u32x2*)(dq + 2 * slot) = v;
                        rnk += g;
                    )
                }
                wave_sync();
                {
                    bool go = d0 + lane < n_deep;
                    const uint32_t pk = go ? dq[2 * lane] : S4_HP + 1u;
                    uint32_t did = go ? dq[2 * lane + 1] : 0u;
                    const uint32_t wpq = (pk & 0xFFFu) - 1u, wL = (pk >> 12) & 63u;
                    // the cell's second word is a value (a key ends within the cell's levels as well: one walker in twelve): the id comes
                    // from cid[] — requested here, looked at when the others have taken their first step (their record gather is
                    // issued behind this load and waited for first: no round trip of its own)
                    const bool pend = !H12 && go && (pk >> 23) != 0u;         // (12-byte cells hand the id over whatever else they hold)
                    uint32_t cidv = 0;
                    if (!H12 && pend) cidv = a.cid[(ACX_S4_EXP & 256) ? ((P.window(wpq) >> (32u - 2u * S4_C)) & 15u) : (P.window(wpq) >> (32u - 2u * S4_C))];   // (256, timing only: cid[] from one line)
                    uint32_t dd = S4_C, wc = 0;
                    int32_t wa = 0, wb = 0;
                    uint32_t s1 = P.sym_at(wpq - S4_C);
                    go = go && !pend && did != 0u;
                    bool first_step = true;
                    for (;;) {
                        const uint32_t single = did >> 31, first = single ^ 1u;
                        uint32_t off = single ? a.single_off + (did << 4) : a.row_off + ((did + s1) << 4);   // (bit 31 shifts out; a row's id is a record index)
                        off = go ? off : a.row_off;
                        if ((ACX_S4_EXP & 512) && !first_step) off = a.row_off + (off & 112u);   // (timing only: the records behind the first from one line)
                        if ((ACX_S4_EXP & 64) && first_step) off = a.row_off + (off & 112u);     // (timing only: the first record gather from one line)
                        const u32x4 rec = *(const u32x4*)(a.deep_base + off);
                        const uint32_t g = go ? 1u : 0u;
                        const uint32_t len = rec.y & 0xFFu;
                        const uint32_t dn = dd + first + len;
                        const uint32_t wq = go ? wpq - dd - first : S4_HP;      // (a walker that is done reads a harmless window)
                        const uint32_t diff = P.label_diff(wq, len, rec.x, rec.y);
                        const uint32_t ok = g & (rec.y >> 9) & (wL >= dn ? 1u : 0u) & ((len == 0u ? 1u : 0u) | (diff == 0u ? 1u : 0u));
                        const uint32_t hit = ok & (rec.y >> 8) & 1u;
                        wa = (hit & (wc == 0u ? 1u : 0u)) ? (int32_t)rec.z : wa;
                        wb = (hit & (wc == 1u ? 1u : 0u)) ? (int32_t)rec.z : wb;
                        wc += hit;
                        dd = dn;
                        did = rec.w;
                        const uint32_t g2 = ok & 1u & (rec.w != 0u ? 1u : 0u) & ((rec.w >> 31) | (wL > dn ? 1u : 0u));
                        go = g2 != 0u;
                        if (first_step) {                                // (wave-uniform) the walkers whose id has arrived meanwhile join
                            first_step = false;
                            if (pend) { dd = S4_C; did = cidv; go = cidv != 0u; }
                        }
                        if (!__any(go)) break;
                        if (ACX_S4_EXP & 32) break;                      // (timing only: one step per walker)
                        s1 = go ? P.sym_at(wpq - dd) : 0u;
                    }
                    dq[2 * lane] = (uint32_t)wa; dq[2 * lane + 1] = (uint32_t)wb; dcnt[lane] = (uint16_t)wc;
                }
                wave_sync();
                {
                    // what the walks found: the count and both values of every slot's walker in one pass (no second look at the
                    // hand-over's memory, no branches: selects)
                    uint32_t rnk = d_base - d0;
                    S4_SLOTS(e,
                        const uint32_t g = (gomask >> e) & 1u;
                        const bool mine = g != 0u && rnk < 64u;
                        const uint32_t ix = mine ? rnk : 64u;
                        const uint32_t c2 = mine ? (uint32_t)dcnt[ix] : 0u;
                        const u32x2 dv = *(const u32x2*)(dq + 2 * ix);
                        const uint32_t ct = cn[e];
                        va[e] = (c2 != 0u && ct == 0u) ? (int32_t)dv.x : va[e];
                        // the second value of the position: the walk's first behind ONE key of the cell, its second behind none
                        const bool has2 = (ct == 0u && c2 > 1u) || (ct == 1u && c2 != 0u);
                        const bool take = has2 && vbe == (uint32_t)S4_NE;
                        vb1 = take ? (ct == 0u ? (int32_t)dv.y : (int32_t)dv.x) : vb1;
                        vbe = take ? (uint32_t)e : vbe;
                        cn[e] = ct + c2;
                        rnk += g;
                    )
                }
                wave_sync();
            }
            // (the requested value is waited for HERE, behind the walks' own waits: left to the compiler, the wait lands behind the
            //  record stores of step 2 — `s_waitcnt vmcnt(2)`, i.e. a wait for the STORES' completion, a quarter of a wave's time)
            asm volatile("" : "+v"(vbt));
            if (vbte < (uint32_t)S4_NE) set_vb(vbte, vbt);
        }

        S4_PH(0);
        S4_TP(2);
        // ---- 2. set O: placement, records ------------------------------------------------------------------------------------
        if (haveO) {
            const uint32_t cg = cgO;
            uint32_t rr[S4_NE];
#pragma unroll
            for (int e = 0; e < S4_NE; e++) rr[e] = (ppO[e] & 0xFFFu) + cg;   // the global positions
            S4_MARK(M_PLACE);
            // place: entry e * 64 + lane; the records of a slot follow those of the slots below it (two slots per prefix
            // sum, 16 bits each: a slot has at most 64 x longest < 65536 records)
            // (the three prefix sums are independent: their steps are interleaved, one chain of six dependent DPP adds instead of
            //  three; slots beyond the round's last count nothing)
            uint32_t ex[S4_NE], rt = 0;
            {
                static_assert(S4_NE == 6, "three pairs of slots");
                const uint32_t v0 = cn[0] | (cn[1] << 16), v1 = cn[2] | (cn[3] << 16), v2 = cn[4] | (cn[5] << 16);
                uint32_t x0 = v0, x1 = v1, x2 = v2;
#define S4_DPP3(ctrl, rmask) do { \
                    const uint32_t a0_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x0, ctrl, rmask, 0xF, false); \
                    const uint32_t a1_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x1, ctrl, rmask, 0xF, false); \
                    const uint32_t a2_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x2, ctrl, rmask, 0xF, false); \
                    x0 += a0_; x1 += a1_; x2 += a2_; } while (0)
                S4_DPP3(0x111, 0xF); S4_DPP3(0x112, 0xF); S4_DPP3(0x114, 0xF); S4_DPP3(0x118, 0xF);     // row_shr 1, 2, 4, 8
                S4_DPP3(0x142, 0xA); S4_DPP3(0x143, 0xC);                                               // row_bcast 15 -> rows 1, 3; 31 -> rows 2, 3
#undef S4_DPP3
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, 63), t1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, 63),
                               t2 = (uint32_t)__builtin_amdgcn_readlane((int)x2, 63);
                x0 -= v0; x1 -= v1; x2 -= v2;
                ex[0] = rt + (x0 & 0xFFFFu); rt += t0 & 0xFFFFu; ex[1] = rt + (x0 >> 16); rt += t0 >> 16;
                ex[2] = rt + (x1 & 0xFFFFu); rt += t1 & 0xFFFFu; ex[3] = rt + (x1 >> 16); rt += t1 >> 16;
                ex[4] = rt + (x2 & 0xFFFFu); rt += t2 & 0xFFFFu; ex[5] = rt + (x2 >> 16); rt += t2 >> 16;
            }
            if (rt && !dead) {
                if (g_used + rt + 1u > g_size) {                       // this round (and the spare slot behind it) does not fit the current grant: open the next one
                    if (ng == PPM_MAX_GRANTS) dead = true;
                    else {
                        uint32_t need = PPM_GRANT << (ng < 10 ? ng : 10);
                        if (need < rt + 1u) need = rt + 1u;
                        unsigned long long oo = 0;
                        if (lane == 0) oo = atomicAdd(a.heads + pool_x, (unsigned long long)need);
                        oo = __shfl(oo, 0, 64);
                        if (oo + need > a.pool_records) dead = true;
                        else {
                            if (lane == 0) { if (ng) desc[18 + ng - 1] = g_used; desc[2 + ng] = (uint32_t)((unsigned long long)pool_x * a.pool_records + oo); }
                            g_base = (uint32_t)((unsigned long long)pool_x * a.pool_records + oo); g_size = need; g_used = 0; ng++;
                        }
                    }
                    if (dead && lane == 0) *a.overflow = 1;
                }
            }
            const bool wr = rt && !dead && !(ACX_S4_EXP & 4);
            S4_MARK(M_REC);
            // records, longest key of a position first.  Slot rt of the round (one past its last record; the grant has the
            // room) takes the stores of the lanes that have no first / second record.
            uint32_t slow = 0, slow1 = 0;                                // slots with more than two records (slow1: or two, the second not in vb1): the general enumeration
            uint8_t* const out8 = (uint8_t*)(a.scratch + g_base + g_used);
            uint2* const out = (uint2*)out8;
            if (wr) {
                uint32_t two = 0;
                S4_SLOTS(e,
                    const uint32_t c = cn[e]; const uint32_t oe = ex[e] + c - 1u;
#ifdef ACX_S4_PRED_STORES
                    if (c) *(uint2*)(out8 + (oe << 3)) = make_uint2(rr[e], (uint32_t)va[e]);
#else
#ifdef ACX_S4_NT_STORES
                    { u32x2 rv_; rv_.x = rr[e]; rv_.y = (uint32_t)va[e]; __builtin_nontemporal_store(rv_, (u32x2*)(out8 + ((c ? oe : rt) << 3))); }
#else
                    *(uint2*)(out8 + ((c ? oe : rt) << 3)) = make_uint2(rr[e], (uint32_t)va[e]);
#endif
#endif
                    two |= (c > 1u ? 1u : 0u) << e;
                    slow |= (c > 2u ? 1u : 0u) << e;
                )
                if (__any(two != 0u)) {
                    const uint32_t mine = vbe < (uint32_t)S4_NE ? 1u << vbe : 0u;   // the slot whose second value vb1 holds
                    slow1 = two & ~mine; two &= mine;
                    S4_SLOTS(e,
                        if ((two >> e) & 1u) *(uint2*)(out8 + ((ex[e] + cn[e] - 2u) << 3)) = make_uint2(rr[e], (uint32_t)vb1);
                    )
                }
                slow |= slow1;
                while (__any(slow != 0u)) {                              // rare: one slot per lane and pass, from the 32-byte cell
                    if (slow) {
                        const uint32_t se = (uint32_t)__ffs(slow) - 1u;
                        slow &= slow - 1u;
                        const uint32_t from = (slow1 >> se) & 1u ? 1u : 2u;
                        typename Ppm<2, true, false>::Ent E;
                        uint32_t oe = 0;
                        E.p = 0; E.X = 0; E.L = 0; E.idx = 0;
#pragma unroll
                        for (int e = 0; e < S4_NE; e++) if (se == (uint32_t)e) { E.p = (ppO[e] & 0xFFFu) - (S4_HP + 1u); E.L = __builtin_amdgcn_ubfe(ppO[e], 12u, 6u); E.idx = rr[e]; oe = ex[e] + cn[e] - 1; }
                        E.X = P.window(S4_HP + E.p);
                        const u32x4* cell = (const u32x4*)((const uint8_t*)a.cells + ((E.X >> (32u - 2u * S4_C)) << 5));
                        E.c0 = cell[0]; E.c1 = cell[1];
                        asm volatile("" : "+v"(E.c0), "+v"(E.c1));       // (waited for here, on the rare path: a load that may still be pending where the paths join costs the COMMON path a wait for everything)
                        const uint32_t idx = E.idx;
                        P.matches(E, from, 0xFFFFFFFFu, [&](uint32_t kk2, int32_t v) { out[oe - kk2] = make_uint2(idx, (uint32_t)v); });
                    }
                }
            }
            if (rt && !dead) g_used += rt;
            run_off += rt;
            wave_sync();

        }

        S4_TP(3);
        // ---- 3. this trip hands over the last candidates of the current tile: the next tile's bytes -> symbols ------------------
        const uint32_t ex_lo = (tile_ok && seg_lo) ? (uint32_t)__builtin_amdgcn_readlane((int)x_ex, (int)seg_lo) : 0u;
        const uint32_t qcap = use_other ? S4_QCAP_OTHER : S4_QCAP;
        const bool will_adv = tile_ok && x_tot - ex_lo <= qcap;
        const bool st = will_adv && tiles_left > 1u && e0 + S4_TPOS < H;
        if (st) convert(tiles_left > 2u, e0 + 2u * S4_TPOS);

        S4_PH(1);
        S4_TP(4);
        // ---- 4. the next round of this tile: its candidates -> the queue, in position order; fetch -------------------------------
        bool haveN = false;
        if (tile_ok && x_tot != 0u) {
            S4_MARK(M_PUSH);
            uint32_t seg_hi = 64u, n = x_tot - ex_lo;
            if (n > qcap) { seg_hi = seg_lo + 8u; n = (seg_hi < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)x_ex, (int)seg_hi) : x_tot) - ex_lo; }   // (eight lanes: 256 positions)
            {
                uint32_t w = (lane >= seg_lo && lane < seg_hi) ? pw : 0u;
                uint32_t ja = wbase + S4_QUEUE + 2u * (x_ex - ex_lo);
                const uint32_t lp = 32u * lane + S4_HP + 1u;
                while (w) {
                    const uint32_t b = (uint32_t)__builtin_ctz(w);
                    w &= w - 1u;
                    lds_wr16(ja, lp + b);
                    ja += 2u;
                }
            }
            seg_lo = seg_hi;
            wave_sync();
            S4_TP(5);
            if (n && !(ACX_S4_EXP & 8)) {
                S4_MARK(M_FETCH);
                // where the entries sit, their windows, the requests for their hot cells.  Straight-line over the slots (lane l
                // owns entries l, 64 + l, ..; slots are worked in pairs, a pair beyond the queue's end is skipped): their LDS
                // reads and gathers overlap.  A slot beyond the queue's end asks the spare cell (all zero: nothing ends
                // there, nothing goes deeper).
                const uint32_t k = (n + 63u) >> 6;                   // (slots of set N: shadows the count of set O)
                const uint32_t cx1 = r_tile - 32u;                  // entry + cx1 = 1 + (offset of the tile's first byte in its haystack + position)
                const uint32_t sbase = wbase + cur * S4_SYMB;
                uint32_t LL[S4_NE];
#pragma unroll
                for (int e = 0; e < S4_NE; e++) LL[e] = longest;           // (slots beyond the round's last keep what they held: nobody reads them)
                if (use_other) {                                    // bytes of no key around (rare): the symbols that exist going back from every entry
                    S4_SLOTS(e,
                        const uint32_t qi = 64u * (uint32_t)e + lane;
                        const uint32_t lo2 = other_limit(qi < n ? (uint32_t)queue[qi] - 1u : S4_HP);
                        if (lo2 < LL[e]) LL[e] = lo2;
                    )
                }
                S4_SLOTS(e,
                    const uint32_t qi = 64u * (uint32_t)e + lane;
                    const uint32_t ent = lds_rd16(qaddr + 128u * (uint32_t)e);      // position + 33
                    // 1 + the offset in its haystack = x1 - stride * floor((x1 - 1) / stride), the quotient by a 24-bit multiply
                    // (exact while x1 - 1 < stride + 2048)
                    uint32_t L = LL[e];
                    if (!OFFS) {
                        const uint32_t x1 = ent + cx1;
                        const uint32_t q = ((uint32_t)__umul24(x1, m24) + nm24) >> 23;
                        const uint32_t L0 = x1 - (uint32_t)__umul24(q, stride);
                        L = L0 < LL[e] ? L0 : LL[e];
                    }
                    // the 32 bits of symbols that end with the entry's position: words (ent >> 4) + 1 and + 2 of the symbol buffer
                    const uint32_t wa = ((ent >> 2) & 0x7FCu) + sbase;
                    const uint32_t X = __builtin_amdgcn_alignbit(lds_rd32(wa + 8u), lds_rd32(wa + 4u), ent << 1);
                    const uint32_t off = (X >> (32u - 2u * S4_C - 3u)) & ((8u << (2u * S4_C)) - 8u);
                    if constexpr (H12) {
                        const uint32_t o12 = qi < n ? off + (off >> 1) : (12u << (2u * S4_C));      // (cell x 12 bytes; the spare cell behind the last)
                        hcO[e] = *(const u32x3a*)((const uint8_t*)a.hot4 + o12);
                    } else
                    hcO[e] = *(const u32x2*)((const uint8_t*)a.hot4 + ((ACX_S4_EXP & 1) ? ((qi < n ? off : 0u) & 56u) : (qi < n ? off : (8u << (2u * S4_C)))));
                    const uint32_t t14 = __builtin_amdgcn_ubfe(X, 32u - 2u * (S4_C + 7u), 14u);         // the seven symbols below the cell's (the first in the top bits): what the cell's "go deeper" field is asked about
                    ppO[e] = ent | (L << 12) | (t14 << 18);
                )
                wave_sync();                                         // (the queue's memory is free from here on)
                nN = n; cgN = e0 - 33u; symN = sbase; haveN = true;
            }
        }

        S4_PH(3);
        S4_TP(6);
        // ---- the round that was fetched is the next trip's set O; a tile that has handed over all its candidates makes room for the next ---------------------
        S4_MARK(M_TAIL);
        nO = nN; cgO = cgN; symO = symN; haveO = haveN;
        if (will_adv) {
            // the tail of this tile is the halo of the next (which is staged into the other buffer)
            const uint32_t* const sc = (const uint32_t*)(lds + wbase + cur * S4_SYMB);
            uint32_t* const sn = (uint32_t*)(lds + wbase + (cur ^ 1u) * S4_SYMB);
            if (lane < 2u) sn[2u + lane] = sc[130u + lane];
            if (use_other && lane == 0u) obits[0] = obits[64];
            any_prev = any_cur;
            e0 += S4_TPOS;
            r_tile += step_r;
            if (r_tile >= stride) r_tile -= stride;
            cur ^= 1u;
            tiles_left--;
            tile_ok = st;
            wave_sync();
            S4_TP(7);
            // ---- 5. stage the next tile ------------------------------------------------------------------------------------
            if (st) stage_rest();
        }
        S4_PH(4);
        S4_TP(10);
#ifdef ACX_S4_TSEG
        tp_have = 0;
#endif
        if (!tile_ok && !haveO) break;
    }
#undef S4_SLOTS
#ifdef ACX_S4_PHASES
    if (lane == 0 && a.phase_out) for (int i = 0; i < 8; i++) atomicAdd(a.phase_out + i, ph[i]);
#endif
#ifdef ACX_S4_TSEG
    if (lane == 0 && a.phase_out) { atomicAdd(a.phase_out + 0, (unsigned long long)tp_sum_v); atomicAdd(a.phase_out + 1, (unsigned long long)tp_n_v); atomicAdd(a.phase_out + 7, 1ull); }
#endif
#ifdef ACX_S4_WAVETIME
    if (lane == 0 && a.phase_out && wave_id < 4096u) { a.phase_out[8 + 2 * wave_id] = wt0; a.phase_out[8 + 2 * wave_id + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (lane == 0) {
        if (ACX_S4_EXP & 5) { run_off = 0; ng = 0; }                   // (timing-only builds: nothing for the gather to move)
        desc[0] = run_off; desc[1] = ng; if (ng) desc[18 + ng - 1] = g_used;
        // the records of the 16 waves of this block, summed where k_ppm_gather_pos finds them (relaxed: no fences, see k_ppm_stream)
        if (a.block_sum && run_off) __hip_atomic_fetch_add(a.block_sum + blockIdx.x, run_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

// Does k_ppm_stream4 take this scan?  (acx_ppm_args as scan_ppm filled them for the stream kernels.)
static bool s4_image_ok(const acx_ppm_args& a) {
    return a.fast && a.hot4 && (a.cid || a.hot12) && !a.skip &&
           a.sym_bits == 2 && a.pow2 && a.sym_arith != 0 && a.K == 4 && !a.g_global && !a.F2 && a.nsub == 8 &&
           a.C == S4_C && a.F == S4_F && a.halo_pos == S4_HP && a.longest <= S4_HP + 1u && a.g_words * 4u == S4_G_BYTES;
}
bool acx_ppm_stream4_eligible(const acx_ppm_args& a) {
    if (a.off) return a.start_bits != nullptr && s4_image_ok(a);      // (an offsets batch: scan_ppm made the start bitmap because acx_ppm_stream4_offs_ok said yes)
    return s4_image_ok(a) && a.m24 && a.stride >= 8 && a.stride < 2048;
}
// ... an offsets batch (before its start bitmap exists)?
bool acx_ppm_stream4_offs_ok(const acx_ppm_args& a) { return a.off != nullptr && s4_image_ok(a); }

hipError_t acx_launch_ppm_stream4(const acx_ppm_args& a, int64_t blocks, hipStream_t s) {
    const size_t lds_bytes = S4_G_BYTES + 16u * S4_WAVE_BYTES;
    auto kernel = a.off ? (a.hot12 ? k_ppm_stream4<true, true> : k_ppm_stream4<false, true>) : (a.hot12 ? k_ppm_stream4<true, false> : k_ppm_stream4<false, false>);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(ACX_PPM_BLOCK), lds_bytes, s, a);
    return hipGetLastError();
}
