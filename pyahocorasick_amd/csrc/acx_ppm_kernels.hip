// acx_ppm_kernels.hip — position-parallel scan (ACX_SCAN_ALL) for gfx950.  Layout of the image:
// include/acx_blob.h "ppm"; builder: acx_ppm.cpp; CPU restatement: oracle/ppm_walk.c.
//
// The reference's search loop (automaton_search_iter_next, src/AutomatonSearchIter.c:243-300) carries
// a state from byte to byte.  What it REPORTS at position e — the keys that are suffixes of the
// text up to e, longest first (automaton_build_output, :157-197) — depends on the last
// longest_word bytes only, so here every position is its own unit of work:
//
//   k_ppm_stream   the default (further down): runs of tiles per wave, register windows, a ring queue, wide rounds;
//                  then k_ppm_wave_scan + k_ppm_gather.
//   k_ppm_scan     the general form (any stride, chunks, unaligned buffers):
//                  a WAVE owns a tile of 256 end positions.  It loads the tile and its left halo
//                  (longest_word-1 bytes) with coalesced dword loads, turns bytes into packed symbols
//                  in LDS, and every lane tests 4 positions against the filter bitmap G (LDS: "the F
//                  newest symbols may end a key").  Positions that pass are compacted into a queue
//                  (ballot + mbcnt); the queue is worked off 64 entries at a time: one 32-byte cell
//                  (global, indexed by the C newest symbols), then the dense child rows of the
//                  reversed-key trie while the walk goes on.  Matches are counted, placed with a
//                  wave prefix sum and written as final {end_index, value} records into a scratch
//                  pool (the global position of a tile's records is not known yet).
//   scan           exclusive prefix sum of the per-tile counts (acx_kernels.hip).
//   k_ppm_compact  copies every tile's records to their final place and turns tile-local offsets of
//                  the haystack starts into match_off[].
//
// Integer only, no MFMA: there is no contraction on this path.  Bound by instruction issue (one LDS
// bitmap probe per position costs 8 instructions) and by the dependent gathers of the positions that
// pass the filter (a cell, then deep records).
#include "acx_kernels.h"
#include "acx_ppm_layout.h"

#include <type_traits>

#define PPM_TILE  ACX_PPM_TILE
#define PPM_GRANT 1024u            // records a wave takes from the scratch pool at a time
#define PPM_NOBASE 0xFFFFFFFFu
#ifndef ACX_PPM_NE
#define ACX_PPM_NE 6               // k_ppm_stream: queue entries per lane and round (at most)
#endif

#include "acx_ppm_device.h"

namespace {

template <int SB, bool POW2, bool CHUNK>
__global__ void __launch_bounds__(ACX_PPM_BLOCK) k_ppm_scan(const acx_ppm_args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    for (uint32_t i = threadIdx.x; i < a.g_words; i += blockDim.x) smem[a.lds.g_off + i] = a.g[i];
    if (threadIdx.x < 256) ((uint8_t*)(smem + a.lds.map_off))[threadIdx.x] = a.symtab[threadIdx.x];   // (0xFF: a byte of no key)
    __syncthreads();

    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    Ppm<SB, POW2, CHUNK> P(a);
    P.s_g = smem + a.lds.g_off;
    P.s_map = (const uint8_t*)(smem + a.lds.map_off);
    uint32_t* wbase = smem + a.lds.wave_off + (uint32_t)wid * a.lds.wave_words;
    P.s_sym = wbase;
    P.s_oth = (uint8_t*)(wbase + a.lds.sym_words);
    P.s_last = (uint16_t*)(P.s_oth + ((a.lds.dwords + 3u) & ~3u));
    P.s_queue = (uint8_t*)(wbase + a.lds.sym_words + a.lds.oth_words);
    P.s_qoff = wbase + a.lds.sym_words + a.lds.oth_words + a.lds.queue_words;
    if (lane == 0) { P.s_sym[0] = 0; }

    const int64_t n_items = CHUNK ? *a.n_items_dev : a.n_items;
    const int64_t wave0 = (int64_t)wid * gridDim.x + blockIdx.x;      // block-minor: a partial last round spreads over all CUs
    const int64_t n_waves = (int64_t)gridDim.x * ACX_PPM_WAVES;
    const uint32_t pool_x = blockIdx.x % a.n_pools;
    uint32_t g_cur = 0, g_end = 0;                                    // this wave's current grant of the scratch pool (wave-uniform)
    const int64_t H = CHUNK ? a.hay_cap : a.n_hay * a.stride;
    const uint8_t* const hay_end = a.hay + a.hay_cap;

    // geometry of a tile from its descriptor (CHUNK) or its number (STRIDE)
    auto geo_of = [&](int64_t tile, const acx_chunk_desc& d) -> Geo {
        Geo G;
        int64_t g0; uint32_t halo;
        if (CHUNK) {
            g0 = d.start + d.emit; G.npos = d.len - d.emit; halo = (uint32_t)d.emit; G.halo = halo;
            G.idx_first = (uint32_t)(d.idx0 + d.emit); G.e0 = 0;
        } else {
            g0 = tile * PPM_TILE;
            const int64_t left = H - g0;
            G.npos = left < PPM_TILE ? (int32_t)left : PPM_TILE;
            const uint32_t want = a.longest - 1;
            halo = g0 < (int64_t)want ? (uint32_t)g0 : want;
            G.halo = 0; G.idx_first = 0; G.e0 = (uint32_t)g0;
        }
        const uint8_t* first = a.hay + (g0 - halo);
        const uint32_t shift = (uint32_t)((uintptr_t)first & 3u);
        G.abase = first - shift;
        G.q0 = halo + shift;
        G.ndw = (shift + halo + (uint32_t)G.npos + 3u) >> 2;
        return G;
    };
    auto load_desc = [&](int64_t tile) -> acx_chunk_desc {
        acx_chunk_desc d;
        d.start = 0; d.emit = 0; d.len = 0; d.idx0 = 0; d.hay = 0; d.flags = 0; d.pad = 0;
        if (CHUNK && tile < n_items) d = a.ck[tile];
        return d;
    };
    // dword i of a tile's staging area; bytes outside [hay, hay + hay_cap) read as 0
    auto load_dw = [&](const Geo& G, uint32_t i) -> uint32_t {
        uint32_t w = 0;
        if (i < G.ndw) {
            const uint8_t* pw = G.abase + 4 * (size_t)i;
            if (pw >= a.hay && pw + 4 <= hay_end) w = *(const uint32_t*)pw;
            else {
#pragma nounroll
                for (int k = 0; k < 4; k++) if (pw + k >= a.hay && pw + k < hay_end) w |= (uint32_t)pw[k] << (8 * k);
            }
        }
        return w;
    };

    // software pipeline: the haystack dwords of the NEXT tile are requested while this one is worked on
    acx_chunk_desc d_n1 = load_desc(wave0 + n_waves);
    Geo g_cur_tile = geo_of(wave0, load_desc(wave0));
    uint32_t pre0 = 0, pre1 = 0;
    if (wave0 < n_items) { pre0 = load_dw(g_cur_tile, lane); pre1 = load_dw(g_cur_tile, 64 + lane); }

    for (int64_t tile = wave0; tile < n_items; tile += n_waves) {
        P.T = g_cur_tile;
        const Geo& T = P.T;
        const uint32_t w0 = pre0, w1 = pre1;
        // next tile: geometry now (its descriptor arrived an iteration ago), dwords requested now, descriptor after next
        const bool has_next = tile + n_waves < n_items;
        Geo g_next = geo_of(tile + n_waves, d_n1);
        if (has_next) { pre0 = load_dw(g_next, lane); pre1 = load_dw(g_next, 64 + lane); }
        d_n1 = load_desc(tile + 2 * n_waves);
        g_cur_tile = g_next;

        // ---- stage: bytes -> packed symbols ----------------------------------------------
        uint32_t any_other = 0, carry = 0;
        for (uint32_t i0 = 0; i0 < T.ndw; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint32_t w = i0 == 0 ? w0 : (i0 == 64 ? w1 : load_dw(T, i));
            uint32_t packed = 0, nib = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t s = P.s_map[(w >> (8 * k)) & 0xFFu];
                const bool oth = a.has_other && s == 0xFFu;
                nib |= (oth ? 1u : 0u) << k;
                packed |= (oth ? 0u : s) << (SB * k);
            }
            if (i < T.ndw) {
                if (SB == 2) ((uint8_t*)(P.s_sym + 1))[i] = (uint8_t)packed;
                else if (SB == 4) ((uint16_t*)(P.s_sym + 1))[i] = (uint16_t)packed;
                else P.s_sym[1 + i] = packed;
                if (a.has_other) P.s_oth[i] = (uint8_t)nib;
            } else nib = 0;
            if (a.has_other && __any(nib != 0)) any_other = 1;
        }
        if (lane == 0) {                                              // the funnel shift of the newest window reads one word further
            const uint32_t lastw = 1 + ((T.ndw * 4 * SB + 31) >> 5);
            P.s_sym[lastw] = 0;
        }
        P.has_other = any_other;
        if (any_other) {
            // last "other" position (+1) at or before the end of every staged dword: running maximum
            wave_sync();
            for (uint32_t i0 = 0; i0 < T.ndw; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint32_t nib = i < T.ndw ? P.s_oth[i] : 0u;
                uint32_t last = nib ? 4 * i + (31 - __clz(nib)) + 1 : 0u;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(last, d, 64); if (lane >= d && t > last) last = t; }
                if (carry > last) last = carry;
                if (i < T.ndw) P.s_last[i] = (uint16_t)last;
                carry = __shfl(last, 63, 64);
            }
        }
        wave_sync();

        // ---- filter: 4 consecutive end positions per lane; the queue keeps position order --------
        uint32_t pm = 0;                                              // which of this lane's 4 positions passed
        uint32_t hs_r0 = 0, hs_h0 = 0;                                // STRIDE: haystack and offset of the lane's first position
        {
            if (!CHUNK) hs_h0 = div_magic(T.e0 + 4 * lane, a.stride_magic, (uint32_t)a.stride, hs_r0);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t p = 4 * lane + k;
                uint32_t r = 0;
                if (!CHUNK) { r = hs_r0 + k; while (r >= (uint32_t)a.stride) r -= (uint32_t)a.stride; }
                const uint32_t L = P.limit(p, r);
                if (L >= a.min_len) {
                    const uint32_t X = P.window(T.q0 + p);
                    uint32_t cC, cF;
                    P.codes_CF(X, L, cC, cF);
                    pm |= ((P.s_g[cF >> 5] >> (cF & 31u)) & 1u) << k;
                }
            }
        }
        uint32_t nq;
        const uint32_t qb = wave_excl_scan((uint32_t)__popc(pm), nq);
        {
            uint32_t j = qb;
#pragma unroll
            for (int k = 0; k < 4; k++) if ((pm >> k) & 1u) P.s_queue[j++] = (uint8_t)(4 * lane + k);
        }
        wave_sync();

        // ---- count: one queue entry per lane and round; the first two matches keep their values ----
        typename Ppm<SB, POW2, CHUNK>::Ent E0;
        uint32_t c0 = 0, off0 = 0, total = 0;
        int32_t v0a = 0, v0b = 0;
        for (uint32_t b = 0; b < nq; b += 64) {
            const bool act = b + lane < nq;
            uint32_t c = 0;
            if (act) {
                const auto E = P.load_ent(P.s_queue[b + lane]);
                int32_t va = 0, vb = 0;
                c = P.matches(E, 0u, 2u, [&](uint32_t k, int32_t v) { if (k == 0) va = v; else vb = v; });
                if (b == 0) { E0 = E; c0 = c; v0a = va; v0b = vb; }
            }
            uint32_t rt;
            const uint32_t ex = wave_excl_scan(c, rt);
            if (act) P.qoff_set(b + lane, total + ex);
            if (b == 0) off0 = ex;
            total += rt;
        }
        wave_sync();

        if (!CHUNK) {                                                 // haystacks that start in this tile: their tile-local offset
            uint32_t r0 = hs_r0, h = hs_h0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (r0 >= (uint32_t)a.stride) { r0 -= (uint32_t)a.stride; h++; }
                if (r0 == 0 && (int32_t)(4 * lane + k) < T.npos) {
                    const uint32_t qi = qb + (uint32_t)__popc(pm & ((1u << k) - 1u));
                    a.hay_local[h] = (int32_t)(qi < nq ? P.qoff_get(qi) : total);
                }
                r0++;
            }
        }
        // where this tile's records go in the scratch pool
        uint32_t base = PPM_NOBASE;
        if (total) {
            if (g_cur + total > g_end || g_end == 0) {
                const uint32_t need = total > PPM_GRANT ? total : PPM_GRANT;
                unsigned long long o = 0;
                if (lane == 0) o = atomicAdd(a.heads + pool_x, (unsigned long long)need);
                o = __shfl(o, 0, 64);
                if (o + need > a.pool_records) { if (lane == 0) *a.overflow = 1; g_cur = 0; g_end = 0; }
                else { g_cur = (uint32_t)((unsigned long long)pool_x * a.pool_records + o); g_end = g_cur + need; }
            }
            if (g_end) { base = g_cur; g_cur += total; }
        }
        if (lane == 0) { a.counts[tile] = (int32_t)total; a.scr_off[tile] = base; }

        // ---- emit: final records, longest match of a position first ------------------------------------
        if (total && base != PPM_NOBASE) {
            uint2* out = a.scratch + base;
            if ((uint32_t)lane < nq && c0) {
                const uint32_t o = off0 + c0 - 1, idx = E0.idx;
                out[o] = make_uint2(idx, (uint32_t)v0a);
                if (c0 > 1) out[o - 1] = make_uint2(idx, (uint32_t)v0b);
                if (c0 > 2) P.matches(E0, 2u, 0xFFFFFFFFu, [&](uint32_t k, int32_t v) { out[o - k] = make_uint2(idx, (uint32_t)v); });
            }
            for (uint32_t b = 64; b < nq; b += 64) {
                if (b + lane < nq) {
                    const auto E = P.load_ent(P.s_queue[b + lane]);
                    const uint32_t c = P.matches(E, 0u, 0u, [](uint32_t, int32_t) {});
                    if (c) {
                        const uint32_t o = P.qoff_get(b + lane) + c - 1, idx = E.idx;
                        P.matches(E, 0u, 0xFFFFFFFFu, [&](uint32_t k, int32_t v) { out[o - k] = make_uint2(idx, (uint32_t)v); });
                    }
                }
            }
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------------
// Fast path (k_ppm_stream): fixed-stride batches and offset batches whose haystacks are at least 8 bytes long, a
// 4-byte aligned buffer below 4 GiB, a halo of at most 256 positions.  Same image as k_ppm_scan.  What bounds it
// (profiles/r3a_*): instruction issue — an integer instruction costs 1.2 (add, shift, and/or/xor) to 2 ns (everything
// else) per wave and SIMD — and the rate at which the L2s take requests: every divergent lane of a gather is one,
// ~260 G requests/s over the chip while the table fits an XCD's 4 MiB L2, half of that at 8 MiB.  So the structure is
// built for few instructions per position, ONE 8-byte gather per candidate from a table that stays L2 resident, and
// little code (the loop of a wave must stay in the instruction cache it shares with its neighbours):
//   * a wave owns a contiguous RUN of tiles of NSUB x 256 positions; the left halo of a tile is what the previous
//     tile left in LDS, so no byte is staged twice;
//   * a lane owns 4 * NSUB CONTIGUOUS positions.  Four-letter alphabets: bytes -> 2-bit symbols by shift, mask and
//     one multiply per dword (no table lookup); a byte that is none of the four letters shows up as a mismatch against
//     a byte permute of the letters;
//   * the symbols stay in the lane's registers; a filter probe is two funnel shifts by constants, one LDS read and
//     three more instructions; the filter is asked about the symbols as they stand (no offsets, masks or haystack
//     boundaries: the rounds bound every match by the symbols that exist); the outcome is a pass word per lane;
//   * the set bits of the pass words and of the words of haystack starts go into a ring queue in position order; the
//     queue is worked off NE x 64 entries at a time (a ROUND).  An entry asks the 8-byte hot cell of its C newest
//     symbols: which of them end keys, the value of the shallowest, whether the walk can go deeper.  The cells of the
//     NEXT round are requested before this round's deeper walks start: their latencies hide behind each other;
//   * each round counts, places (DPP prefix sums) and writes its records at once: the wave's records form one ordered
//     stream, appended to grants of the scratch pool that the wave lists in its descriptor; k_ppm_gather copies the
//     streams to their final place.  The exclusive prefix of an entry that is a haystack start IS the record offset
//     of its haystack.
// ---------------------------------------------------------------------------------------------------
#define PPM_QCAP_MAX 384u          // queue entries (uint16) the LDS layout has room for: 64 x NE, NE <= 6 — a round takes them all
#define PPM_DESC_WORDS 40u         // per wave: total, n_grants, 16 x base, 16 x count (+ pad)
#define PPM_MAX_GRANTS 16u
#define PPM_GPOS_THREADS 64        // k_ppm_gather_pos: one wave per block, no LDS

// OFFS:  the batch is given by a device offsets array instead of a fixed stride (ragged packets, one long
//        haystack).  Contract: off[0] = 0 and no haystack shorter than 8 bytes (acx_scan_params.min_hay_len), as with a
//        stride of at least 8: a tile holds at most TPOS / 8 starts.  A tile that has more raises a.short_hay and the
//        host issues the scan again on the general kernel.  The starts come from first_h[] (k_ppm_first_h: one binary
//        search per tile) and live in LDS as a bitmap with per-word "last start" and "starts before" tables.
// POW2:  codes are plain bit fields of the window; otherwise Horner in radix K over its symbols.
// GG:    the filter bitmap is read from global memory (wide alphabets: one level deeper than LDS could hold).
// ARITH: SB == 2, four letters that a shift tells apart: symbol = (byte >> shift) & 3.
// M24:   fixed stride below 2048: a position's haystack by a 24-bit multiply; else (stride >= 2048) by a compare.
// NE:    queue entries per lane and round.
// (Wave-uniform facts that the slots of a round depend on are template parameters or select a copy of the round: a
//  branch between two slots, even one that never diverges, keeps the scheduler from overlapping their LDS reads.)
template <int SB, int NSUB, bool POW2, bool OFFS, bool GG, bool ARITH, bool M24, int NE>
__global__ void __launch_bounds__(ACX_PPM_BLOCK) k_ppm_stream(const acx_ppm_args a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    static_assert(!ARITH || (SB == 2 && POW2), "arithmetic symbols: four letters");
    if (!GG) for (uint32_t i = threadIdx.x; i < a.g_words; i += blockDim.x) smem[a.lds.g_off + i] = a.g[i];
    if (GG && a.gh) for (uint32_t i = threadIdx.x; i < ACX_PPM_GH_WORDS; i += blockDim.x) smem[a.lds.g_off + i] = a.gh[i];    // the hashed copy of the global filter
    if (threadIdx.x < 256) {
        // a byte of no key: 0xFF with 8-bit symbols (staged as symbol 0 by convert), 0x80 with narrower ones — its
        // symbol bits are 0 then, which keeps every staged field below K (the filter reads windows unmasked)
        const uint32_t sy = a.symtab[threadIdx.x];
        ((uint8_t*)(smem + a.lds.map_off))[threadIdx.x] = sy == 0xFFu ? (SB == 8 ? 0xFFu : 0x80u) : (uint8_t)sy;
    }
    if (threadIdx.x < ACX_PPM_MAX_C + 2) smem[a.lds.map_off + 64 + threadIdx.x] = a.top_base[threadIdx.x];
    __syncthreads();

    constexpr uint32_t SPW = 32 / SB;                                  // symbols per word
    constexpr uint32_t TPOS = NSUB * 256u;                             // positions per tile
    constexpr uint32_t TW = TPOS / SPW;                                // words of a tile's symbols
    constexpr uint32_t BW = TPOS / 32;                                 // words of the start bitmap
    constexpr uint32_t SMASK = (1u << SB) - 1u;
    constexpr uint32_t PPL = TPOS / 64;                                // positions per lane: 32 or 16, contiguous
    constexpr uint32_t DPL = PPL / 4;                                  // haystack dwords per lane
    constexpr uint32_t OWN = PPL * SB;                                 // bits of packed symbols a lane makes: 32 .. 256
    constexpr uint32_t OW = OWN / 32;                                  // ... in words
    constexpr uint32_t LPS = 256 / PPL;                                // lanes of a sub-step (256 positions)
    static_assert(OWN >= 32 && DPL >= 4, "tiles of 1024 or 2048 positions");
    // Fixed-stride batches: a record carries the GLOBAL position of its match; k_ppm_gather, which moves every record
    // anyway and is bound by memory, turns it into the index inside its haystack and finds the record offset of every
    // haystack from the positions.  Offset batches: haystack starts travel through the queue as entries of their own
    // (the exclusive record count in front of one IS the record offset of its haystack), records carry the final index.
    constexpr bool STARTQ = OFFS;
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform, and the compiler is told so)
    Ppm<SB, POW2, false> P(a);
    P.s_g = GG ? a.g : smem + a.lds.g_off;
    P.s_map = (const uint8_t*)(smem + a.lds.map_off);
    const uint32_t* const s_tb = smem + a.lds.map_off + 64;            // top_base[]
    uint32_t* wbase = smem + a.lds.wave_off + (uint32_t)wid * a.lds.wave_words;
    uint32_t* const sym = wbase;                                       // [0,1] pad, halo words, tile words, pad
    P.s_sym = wbase + 1;                                               // (Ppm counts one pad word)
    const uint32_t HP = a.halo_pos, HW = HP / SPW;                     // halo: positions (multiple of SPW and 4), words
    uint32_t* const obits = wbase + a.lds.sym_words;                   // one bit per staged position: a byte that occurs in no key
    uint32_t* const obits_tile = obits + HP / 32;                      // (the halo is a whole number of words)
    uint16_t* const queue = (uint16_t*)(wbase + a.lds.sym_words + a.lds.oth_words);
    uint32_t* const sbits = wbase + a.lds.sym_words + a.lds.oth_words + a.lds.queue_words;   // OFFS: haystack starts of the tile, one bit per position
    // (byte tables since round 5 — a tile of 2048 positions with its two uint16 tables did not fit beside a 128 KiB filter, and the offsets
    //  batches of a four-letter dictionary ran on tiles of 1024: 0.334 against 0.287 ms per 150 MB on fixed strides, profiles/r5_experiments.md §8)
    uint8_t* const sback = (uint8_t*)(sbits + BW);                     //       bitmap words back from word w to the last word with a start at or before it (255: none)
    uint8_t* const scnt = sback + BW;                                  //       starts before each bitmap word (at most 255 per tile: more raise short_hay)
    uint8_t* const sym_tile_bytes = (uint8_t*)(sym + 2 + HW);
    for (uint32_t i = lane; i < 2 + HW + TW + 1; i += 64) sym[i] = 0;
    for (uint32_t i = lane; i < (HP + TPOS) / 32 + 1; i += 64) obits[i] = 0;
    P.T.q0 = HP; P.T.halo = 0; P.T.idx_first = 0; P.T.ndw = 0; P.T.abase = nullptr; P.T.e0 = 0; P.T.npos = 0; P.has_other = 0;

    // the batch: H bytes, cut into tiles; a wave takes a contiguous run of them
    const uint32_t stride = (uint32_t)a.stride, m24 = a.m24;
    uint32_t H;
    if (OFFS) H = (uint32_t)a.off[a.n_hay]; else H = (uint32_t)(a.n_hay * a.stride - 1) + 1u;   // (the launcher checks the size)
    const int64_t n_tiles = OFFS ? a.n_items : ((int64_t)H + TPOS - 1) / TPOS;
    const int64_t n_waves = (int64_t)gridDim.x * ACX_PPM_WAVES;
    const int64_t tpw = (n_tiles + n_waves - 1) / n_waves;            // tiles per wave: a contiguous run
    const int64_t wave_id = (int64_t)blockIdx.x * ACX_PPM_WAVES + wid;
    // (the waves of a block may take unequal runs — acx_ppm_slot_first_tile, acx_ppm_layout.h; the gather kernels are told the
    //  same shares)
    const int64_t blk_first = (int64_t)blockIdx.x * ACX_PPM_WAVES * tpw;
    const int64_t t_begin = blk_first + acx_ppm_slot_first_tile((uint32_t)wid, (uint32_t)tpw, a.share_a, a.share_b);
    const int64_t t_stop = blk_first + acx_ppm_slot_first_tile((uint32_t)wid + 1u, (uint32_t)tpw, a.share_a, a.share_b);
    const int64_t t_end = t_stop < n_tiles ? t_stop : n_tiles;
    uint32_t* const desc = a.wave_desc + (size_t)wave_id * PPM_DESC_WORDS;
    const uint32_t pool_x = blockIdx.x % a.n_pools;
    const uint32_t step_q = OFFS ? 0u : TPOS / stride, step_r = OFFS ? 0u : TPOS % stride;
    wave_sync();
    if (t_begin >= t_end || (uint64_t)t_begin * TPOS >= H) { if (lane == 0) { desc[0] = 0; desc[1] = 0; } return; }

    // x = offset-in-haystack of the tile's first position + a position of the tile: haystacks crossed, new offset
    auto divmod = [&](uint32_t x, uint32_t& r) -> uint32_t {
        uint32_t q;
        if (M24) { q = (uint32_t)__umul24(x, m24) >> 23; r = x - (uint32_t)__umul24(q, stride); }   // stride < 2048: exact for x < stride + 2048
        else { q = x >= stride ? 1u : 0u; r = x - (q ? stride : 0u); }   // stride >= 2048: one haystack start per tile at most
        return q;                                                      // (24-bit multiplies: full rate; the cast: a logical shift)
    };
    auto load_dw = [&](uint32_t b) -> uint32_t {                      // the dword at byte b of the buffer (b is a multiple of 4)
        if ((int64_t)b + 4 <= a.hay_cap) return *(const uint32_t*)(a.hay + b);
        return load_dw_tail(a.hay, a.hay_cap, b);
    };
    // bytes -> packed symbols (4 x SB bits); nib: which of the four bytes occur in no key
    const uint32_t ar_shift = ARITH ? a.sym_arith - 1u : 0u, ar_lut = a.sym_lut;
    auto convert = [&](uint32_t w, uint32_t& nib) -> uint32_t {
        nib = 0;
        if (ARITH) {
            // symbol = (byte >> shift) & 3; the four letters permuted by the symbols give the bytes back iff all four are
            // letters; one multiply gathers the four 2-bit fields into the top byte (no two partial products overlap)
            const uint32_t x = (w >> ar_shift) & 0x03030303u;
            const uint32_t d = __builtin_amdgcn_perm(0u, ar_lut, x) ^ w;
            if (d) nib = ((d & 0xFFu) ? 1u : 0u) | ((d & 0xFF00u) ? 2u : 0u) | ((d & 0xFF0000u) ? 4u : 0u) | ((d >> 24) ? 8u : 0u);
            return (x * 0x01041040u) >> 24;
        }
        uint32_t packed = 0, seen = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sv = P.s_map[(w >> (8 * k)) & 0xFFu];
            if (SB == 8) { const bool o = a.has_other && sv == 0xFFu; nib |= (o ? 1u : 0u) << k; packed |= (o ? 0u : sv) << (SB * k); }
            else { seen |= sv; packed |= (sv & SMASK) << (SB * k); }
        }
        if (SB != 8 && (seen & 0x80u)) {                               // (symbols are < 16 here: bit 7 means "no key has it")
#pragma unroll
            for (int k = 0; k < 4; k++) if (P.s_map[(w >> (8 * k)) & 0xFFu] == 0x80u) nib |= 1u << k;
        }
        return packed;
    };
    auto put_sym = [&](uint8_t* base, uint32_t i, uint32_t packed) {
        if (SB == 2) base[i] = (uint8_t)packed; else if (SB == 4) ((uint16_t*)base)[i] = (uint16_t)packed; else ((uint32_t*)base)[i] = packed;
    };
    // symbols available going back from staged position q when bytes of no key are around: the distance to the last
    // one at or before q (the halo is as long as the longest key, so running out of bitmap means "far enough")
    auto other_limit = [&](uint32_t q) -> uint32_t {
        uint32_t w = q >> 5;
        uint32_t m = obits[w] & (0xFFFFFFFFu >> (31u - (q & 31u)));
        while (m == 0u && w > 0u) m = obits[--w];
        const uint32_t last = m ? 32u * w + (31u - (uint32_t)__clz(m)) + 1u : 0u;
        return q + 1 - last;
    };
    // code of the n newest symbols of a window
    auto code_n = [&](uint32_t Xm, uint32_t n) -> uint32_t {
        if (POW2) return Xm >> (32 - SB * n);
        uint32_t c = 0;
        for (uint32_t i = 1; i <= n; i++) c = (uint32_t)__umul24(c, a.K) + __builtin_amdgcn_ubfe(Xm, 32 - SB * i, (uint32_t)SB);   // (codes stay below 2^27 / K < 2^24 before the last step: a full-rate 24-bit multiply)
        return c;
    };

#ifdef ACX_PPM_DEV
    // experiment: waves of one SIMD (w, w + 4, w + 8, w + 12) start a quarter of a tile period apart
    { const uint32_t st = (a.dbg >> 8) & 0xFFu; for (uint32_t i = 0; i < st * ((uint32_t)wid >> 2); i++) __builtin_amdgcn_s_sleep(32); }
#endif
    // ---- prologue: the halo of the run's first tile -----------------------------------------
    uint32_t e0 = (uint32_t)(t_begin * TPOS);
    uint32_t any_prev = 0;
    if (e0 > 0) {
        const uint32_t have = e0 < HP ? e0 : HP;                       // bytes in front of the run (a multiple of 4)
        uint32_t nib = 0, packed = 0;
        if (4u * lane < have) {
            packed = convert(*(const uint32_t*)(a.hay + (e0 - have) + 4u * lane), nib);
            put_sym(sym_tile_bytes - (size_t)(have / 4) * (SB / 2), lane, packed);      // the halo ends where the tile begins
            if (nib) { const uint32_t qs = HP - have + 4u * lane; atomicOr(&obits[qs >> 5], nib << (qs & 31u)); }
        }
        if (__any(nib != 0)) any_prev = 1;
    }
    uint32_t h_tile = 0, r_tile = 0;                                   // STRIDE: haystack of the tile's first byte, its offset in it
    if (!OFFS) {
        uint32_t rr0; h_tile = div_magic(e0, a.stride_magic, stride, rr0); r_tile = rr0;
        h_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)h_tile); r_tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)r_tile);
    }
    // a lane's PPL bytes of a tile (contiguous: 16-byte loads; read once: they need not stay in the caches)
    // (register quadruples, as the loads fill them: scalars made the compiler copy loaded registers — a wait for a load just issued, tools/isa_waits.py)
    auto load_lane = [&](uint32_t b, u32x4 (&w)[DPL / 4]) {
        if ((int64_t)b + PPL <= a.hay_cap) {
#pragma unroll
            for (int j = 0; j < (int)DPL / 4; j++) { const u32x4a v = __builtin_nontemporal_load((const u32x4a*)(a.hay + b + 16u * j)); w[j] = v; }
        } else {
#pragma unroll
            for (int j = 0; j < (int)DPL; j++) w[j / 4][j % 4] = load_dw(b + 4u * j);
        }
    };
    u32x4 wnext[DPL / 4];
    load_lane(e0 + PPL * (uint32_t)lane, wnext);
    int64_t fh_next = OFFS ? a.first_h[t_begin] : 0;

    // the wave's record stream
    uint32_t run_off = 0;                                              // records so far
    uint32_t g_base = 0, g_size = 0, g_used = 0, ng = 0;               // current grant of the pool
    bool dead = false;                                                 // pool or grant list exhausted: keep counting, stop writing
    uint32_t qcount = 0;
    constexpr uint32_t PPM_QCAP = 64u * NE;                            // a round takes the whole queue
    static_assert(PPM_QCAP <= PPM_QCAP_MAX && PPM_QCAP >= 256u, "a sub-step (256 positions) fits the empty queue");
    const uint32_t Cn = a.C;
    const uint32_t cmask = (1u << Cn) - 1u;                            // (C <= 16)
    const uint32_t longest = a.longest;

#ifdef ACX_PPM_PHASES
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tph = __builtin_amdgcn_s_memtime();
#define PH(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[i] += t_ - tph; tph = t_; } while (0)
#elif defined(ACX_PPM_MARK)
#define PH(i) asm volatile("; MARK " #i)
#else
#define PH(i) do { } while (0)
#endif
    for (int64_t tile = t_begin; tile < t_end; tile++) {
        if (e0 >= H) break;
        PH(7);
        const uint32_t left = H - e0;
        const uint32_t npos = left < TPOS ? left : TPOS;
        // ---- stage (and request the next tile's bytes) ---------------------------------------------
        // W[0]: the 32 bits of symbols in front of the lane's own (read back from LDS below), W[1..]: its own
        uint32_t anyo = 0;                                             // which of the lane's positions hold a byte of no key
        uint32_t W[OW + 2];
#pragma unroll
        for (int k = 0; k <= (int)OW + 1; k++) W[k] = 0;
        if (ARITH) {
            // four dwords (16 positions) make one word of symbols: the top bytes of the four products
            uint32_t diff = 0;
#pragma unroll
            for (int g = 0; g < (int)DPL / 4; g++) {
                uint32_t pr[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t w = wnext[g][j];
                    const uint32_t x = (w >> ar_shift) & 0x03030303u;
                    diff |= __builtin_amdgcn_perm(0u, ar_lut, x) ^ w;
                    pr[j] = x * 0x01041040u;
                }
                W[1 + g] = __builtin_amdgcn_perm(pr[1], pr[0], 0x0c0c0703u) | __builtin_amdgcn_perm(pr[3], pr[2], 0x07030c0cu);
            }
            if (__any(diff != 0u)) {                                   // some byte of the tile is none of the four letters
#pragma unroll
                for (int j = 0; j < (int)DPL; j++) { uint32_t nb; (void)convert(wnext[j / 4][j % 4], nb); anyo |= nb << (4 * j); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < (int)DPL; j++) {
                uint32_t nb;
                const uint32_t packed = convert(wnext[j / 4][j % 4], nb);
                W[1 + (4 * SB * j) / 32] |= packed << ((4 * SB * j) & 31);
                anyo |= nb << (4 * j);
            }
        }
#pragma unroll
        for (int k = 0; k < (int)OW; k++) ((uint32_t*)sym_tile_bytes)[OW * lane + k] = W[1 + k];
        if (tile + 1 < t_end) load_lane(e0 + TPOS + PPL * (uint32_t)lane, wnext);
        // OFFS: the haystack starts of this tile -> bitmap, last-start and count tables
        uint32_t hbase = 0, base_r = 0;                                // haystack covering the tile's first byte; that byte's offset in it
        if (OFFS) {
            const int64_t fh = fh_next;
            fh_next = a.first_h[tile + 1];
            int64_t fe = fh_next < a.n_hay ? fh_next : a.n_hay;
            uint32_t m = fe > fh ? (uint32_t)(fe - fh) : 0u;
            constexpr uint32_t MAX_STARTS = TPOS / 8 < 255u ? TPOS / 8 : 255u;   // (the count table is bytes)
            if (m > MAX_STARTS) {                                       // the contract is broken (a haystack shorter than 8 bytes; or a tile of 2048 positions that is all 8-byte haystacks):
                m = MAX_STARTS;                                         // say so; the host scans again on the general kernel
                if (lane == 0) *a.short_hay = 1;
            }
            if ((uint32_t)lane < BW) sbits[lane] = 0;
            wave_sync();
            for (uint32_t j = lane; j < m; j += 64) {
                const uint32_t sp = (uint32_t)(a.off[fh + j] - (int64_t)e0);
                if (sp < TPOS) atomicOr(&sbits[sp >> 5], 1u << (sp & 31u));
            }
            wave_sync();
            {
                const uint32_t wv = (uint32_t)lane < BW ? sbits[lane] : 0u;
                uint32_t tot;
                const uint32_t before = wave_excl_scan((uint32_t)__popc(wv), tot);
                // two starts at one position — an empty haystack, which the contract excludes as well — would count as
                // one: the ranks of everything behind it would be off by one.  Same answer: say so, the host scans again.
                if (tot != m && lane == 0) *a.short_hay = 1;
                uint32_t lastw = wv ? (uint32_t)lane + 1u : 0u;         // 1 + the last word at or before this one that holds a start (0: none)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(lastw, d, 64); if (lane >= d && t > lastw) lastw = t; }
                if ((uint32_t)lane < BW) { sback[lane] = lastw ? (uint8_t)((uint32_t)lane + 1u - lastw) : (uint8_t)255; scnt[lane] = (uint8_t)before; }
            }
            hbase = (uint32_t)(fh - 1);                                 // (fh = 0: wraps; off[0] = 0, so every position of that tile has rank >= 1)
            base_r = fh > 0 ? (uint32_t)((int64_t)e0 - a.off[fh - 1]) : 0u;
        }
        const uint32_t any_cur = a.has_other && __any(anyo != 0) ? 1u : 0u;
        const uint32_t use_other = any_cur | any_prev;
        if (PPL == 32) obits_tile[lane] = anyo;
        else ((uint16_t*)obits_tile)[lane] = (uint16_t)anyo;
        wave_sync();

        // where position p sits: its offset in its haystack, and how many haystacks start at or before it
        auto where = [&](uint32_t p, uint32_t& r, uint32_t& rank) {
            if (!OFFS) { rank = divmod(r_tile + p, r); return; }
            const uint32_t w = p >> 5;
            const uint32_t prev = sbits[w] & (0xFFFFFFFFu >> (31u - (p & 31u)));      // starts of this word at or before p
            const uint32_t back = w ? sback[w - 1] : 255u;
            rank = scnt[w] + (uint32_t)__popc(prev);
            if (prev) r = p - (32u * w + (31 - __clz(prev)));
            else if (back != 255u) { const uint32_t wb = w - 1u - back; r = p - (32u * wb + (31 - __clz(sbits[wb]))); }
            else r = base_r + p;
        };

        // ---- filter: every lane tests its own PPL positions, windows in registers ------------------------
        // The bitmap is asked about the F newest symbols as they stand — bytes of the previous haystack and
        // bytes of no key (staged as symbol 0) included: that can only add candidates (a key of length l needs
        // its own l symbols only, and G says yes for any older ones), and the rounds below bound every match
        // by the symbols that really exist.
        W[0] = ((const uint32_t*)sym_tile_bytes)[(int)(OW * lane) - 1];
        uint32_t pw = 0;                                               // positions that pass
#ifdef ACX_PPM_DEV
        if (a.dbg & 16u) { pw = W[1] & W[2 <= OW ? 2 : 1]; } else       // timing only: no filter
#endif
        if (POW2) {
            // The stream of the lane: W[0] (32 older bits), W[1..OW].  Symbol i ends at stream bit 32 + SB (i + 1) and the code
            // of the F newest symbols is the FB = SB F bits below that.  U = the stream shifted right by 32 - FB (a
            // wave-uniform amount, once per tile): now the code of position i starts at the CONSTANT bit SB (i + 1) of U.  Its
            // word index, times 4 (the byte address in the bitmap), is a funnel shift of U by a constant and a mask; its bit
            // index is the low 5 bits of another (a shift reads no more of its amount); the pass bit enters the accumulator
            // from the top, so that after PPL of them position i sits at bit i.
            const uint32_t FB = SB * a.F;                              // 5 .. 24 bits (the launcher checks)
            const uint32_t amask = ((1u << (FB - 5u)) - 1u) << 2;
            const uint32_t ush = 32u - FB;
            uint32_t U[OW + 3];
#pragma unroll
            for (int k = 0; k < (int)OW; k++) U[k] = __builtin_amdgcn_alignbit(W[k + 1], W[k], ush);
            U[OW] = W[OW] >> ush; U[OW + 1] = 0; U[OW + 2] = 0;
            uint32_t acc = 0;
            if (GG && a.gh) {
                // The filter lives in global memory: one L2 request per position, and the L2s take about 260 G of them a second
                // over the chip — what bounds a scan of a million signatures (profiles/r3_c4_*).  gh (include/acx_blob.h) is a
                // hashed copy of it in LDS with no false negatives: a position it rejects asks G for a word that every
                // rejected lane of the instruction asks for (one request), the others ask for their own as before.
                const uint32_t cmask = FB >= 32u ? 0xFFFFFFFFu : (1u << FB) - 1u;
#pragma unroll
                for (int i0 = 0; i0 < (int)PPL; i0 += 16) {
                    uint32_t gw[16], bs[16], hb[16];
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t b = SB * (uint32_t)(i0 + i + 1), k = b >> 5, sh = b & 31u;
                        bs[i] = sh ? __builtin_amdgcn_alignbit(U[k + 1], U[k], sh) : U[k];
                        const uint32_t ix = __umulhi((bs[i] & cmask) * ACX_PPM_GH_MUL, ACX_PPM_GH_BITS);
                        hb[i] = (smem[a.lds.g_off + (ix >> 5)] >> (ix & 31u)) & 1u;
                    }
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t A = hb[i] ? (bs[i] >> 3) & amask : 0u;
                        gw[i] = *(const uint32_t*)((const uint8_t*)a.g + A);
                    }
#pragma unroll
                    for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_alignbit((gw[i] >> (bs[i] & 31u)) & hb[i], acc, 1u);
                }
            } else {
#pragma unroll
            for (int i0 = 0; i0 < (int)PPL; i0 += 16) {                // (16 probes in flight at a time: registers)
                uint32_t gw[16], bs[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t b = SB * (uint32_t)(i0 + i + 1), k = b >> 5, sh = b & 31u;    // (constants after unrolling)
                    bs[i] = sh ? __builtin_amdgcn_alignbit(U[k + 1], U[k], sh) : U[k];      // the code, and above it newer symbols
                    const uint32_t A = (bs[i] >> 3) & amask;
                    gw[i] = GG ? *(const uint32_t*)((const uint8_t*)a.g + A) : *(const uint32_t*)((const uint8_t*)smem + A);   // (the bitmap is the first thing in LDS: g_off = 0)
                }
#pragma unroll
                for (int i = 0; i < 16; i++) acc = __builtin_amdgcn_alignbit(gw[i] >> (bs[i] & 31u), acc, 1u);
            }
            }
            pw = PPL == 32 ? acc : acc >> (32 - PPL);
        } else {
            // Horner codes; optional second-level filter (include/acx_blob.h "G2"): a position that passes G is asked again
            // with F2 > F symbols in a bitmap in global memory (L2 resident) before it costs a queue entry.  Its code
            // extends the first one by the older symbols, which sit in the lane's registers too (Wm: the second word in
            // front of its own); the probe is issued only where G said yes.
            const uint32_t F2 = !GG ? a.F2 : 0u;                        // (a filter in global memory has no second level)
            uint32_t Wm = 0;
            if (F2) Wm = ((const uint32_t*)sym_tile_bytes)[(int)(OW * lane) - 2];
            auto old_sym = [&](int p, int d) -> uint32_t {            // the symbol d positions before the lane's position p (p, d: constants)
                const int b = (int)SB * (p - d);                       // its bit offset from the lane's first symbol
                const uint32_t word = b >= 0 ? W[1 + b / 32] : (b >= -32 ? W[0] : Wm);
                return __builtin_amdgcn_ubfe(word, (uint32_t)(b & 31), (uint32_t)SB);
            };
#pragma unroll
            for (int i0 = 0; i0 < (int)PPL; i0 += 16) {
                uint32_t gw[16], cf[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const uint32_t e = SB * (i0 + i + 1), k = e >> 5, sh = e & 31u;
                    const uint32_t X = sh ? __builtin_amdgcn_alignbit(W[k + 1], W[k], sh) : W[k];
                    cf[i] = code_n(X, a.F);
                    gw[i] = P.s_g[cf[i] >> 5];
                }
                uint32_t pm = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) pm |= __builtin_amdgcn_ubfe(gw[i], cf[i], 1u) << i;
                if (F2) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        uint32_t c2 = cf[i];
#pragma unroll
                        for (int d = (int)SPW; d < 2 * (int)SPW; d++)       // (G2 exists only where F fills the window: F = SPW; two words of older symbols are at hand)
                            if ((uint32_t)d < F2) c2 = (uint32_t)__umul24(c2, a.K) + old_sym(i0 + i, d);   // (codes stay below 2^25: 24-bit multiply)
                        cf[i] = c2;
                        gw[i] = 0xFFFFFFFFu;
                        if ((pm >> i) & 1u) gw[i] = a.g2[c2 >> 5];
                    }
#pragma unroll
                    for (int i = 0; i < 16; i++) pm &= ~((__builtin_amdgcn_ubfe(gw[i], cf[i], 1u) ^ 1u) << i);
                }
                pw |= pm << i0;
            }
        }
        PH(0);                                                           // stage + filter
        // haystack starts among the lane's positions (offset batches)
        uint32_t sw = 0;
        if (STARTQ) sw = PPL == 32 ? sbits[lane] : ((const uint16_t*)sbits)[lane];
        {
            const uint32_t lp = PPL * (uint32_t)lane;
            const uint32_t nv = npos > lp ? (npos - lp < PPL ? npos - lp : PPL) : 0u;
            const uint32_t vm = nv >= 32u ? 0xFFFFFFFFu : (1u << nv) - 1u;
            pw &= vm & ~anyo;                                          // (a byte of no key ends no key)
            sw &= vm;
        }
        // ---- expansion: candidates and haystack starts -> the queue, in position order; then a round -----
        // Everything at once when the queue has the room (the usual case), else a sub-step (256 positions) at a time,
        // with a round in between when the next sub-step does not fit.
        uint32_t xw = pw | sw, x_tot;
        const uint32_t x_ex = wave_excl_scan((uint32_t)__popc(xw), x_tot);

        // ---- a round: every entry of the queue (n <= 64 NE, in position order).  Lane l owns entries l, 64 + l, ..: k = ceil(n / 64)
        // of them (a store of one slot's records covers consecutive addresses); the hot cells of all its entries are requested
        // before the first is looked at; the walks below the cells — one entry in six asks for one — are collected over the
        // whole round and run on all 64 lanes at once, their entries handed over through the queue's memory, which is
        // free by then.  Straight-line code per entry slot; slots >= k are skipped (wave-uniform).
        // Slots are worked in PAIRS: a pair beyond the queue's end is skipped (tiles of sparse dictionaries hold a handful of
        // entries), and the two slots of a pair stay one basic block, so their LDS reads and gathers overlap.
#define PPM_SLOTS(e, ...) _Pragma("unroll") for (int g_ = 0; g_ < NE; g_ += 2) { if (g_ == 0 || (uint32_t)g_ < k) { _Pragma("unroll") for (int e = g_; e < g_ + 2; e++) { __VA_ARGS__ } } }
        auto do_round = [&]() {
            const uint32_t n = qcount;
            const uint32_t k = (n + 63u) >> 6;                           // slots that hold entries
            uint32_t pp[NE], XX[NE], rr[NE], LL[NE], cn[NE];
            u32x2 hc[NE];
            int32_t va[NE];
            // the second value of an entry (two keys end at one position: 0.6 % of the matching positions of config 2): one
            // per lane and round in registers, with the slot it belongs to; a lane that meets a second such entry in one
            // round sends it through the general enumeration below.  (Six registers of second values, one per slot, are
            // what kept this kernel above 112 registers: k_ppm_gather_pos then finds no room beside it on a CU.)
            int32_t vb1 = 0; uint32_t vbe = NE;
            auto set_vb = [&](uint32_t e, int32_t v) { if (vbe == (uint32_t)NE) { vb1 = v; vbe = e; } };
            // 0. bytes of no key around (rare): the symbols that exist going back from every entry.  Done apart from the
            // slots below, which must stay one basic block: a branch between two of them, even one that never diverges,
            // keeps the scheduler from overlapping their LDS reads.
#pragma unroll
            for (int e = 0; e < NE; e++) { LL[e] = longest; pp[e] = 0x18000u; XX[e] = 0; rr[e] = 1; cn[e] = 0; va[e] = 0; hc[e].x = 0; hc[e].y = 0; }
            if (use_other) {
                PPM_SLOTS(e,
                    const uint32_t qi = 64u * (uint32_t)e + (uint32_t)lane;
                    const uint32_t lo2 = other_limit(HP + (qi < n ? (uint32_t)queue[qi] & 0x7FFFu : 0u));
                    if (lo2 < LL[e]) LL[e] = lo2;
                )
            }
            // 0b. streams (acx_scan_params.dev_skip): an entry inside the context of its haystack ends no reported match
            if (a.skip) {
                PPM_SLOTS(e,
                    const uint32_t qi = 64u * (uint32_t)e + (uint32_t)lane;
                    uint32_t r; uint32_t rk;
                    where(qi < n ? (uint32_t)queue[qi] & 0x7FFFu : 0u, r, rk);
                    uint32_t hh = (OFFS ? hbase : h_tile) + rk;
                    if (OFFS && hh == 0xFFFFFFFFu) hh = 0;
                    if (r < (uint32_t)a.skip[hh]) LL[e] = 0;
                )
            }
            // 1. where the entries sit, their windows, the requests for their hot cells.  Straight-line over all NE slots
            // (a slot beyond the queue's end works on position 0 with L = 0, which matches nothing): the LDS reads and
            // the gathers of the slots overlap.
            PPM_SLOTS(e,
                const uint32_t qi = 64u * (uint32_t)e + (uint32_t)lane;
                const bool act = qi < n;
                uint32_t ent = queue[qi];                                // (within the queue's memory for any qi < 64 NE)
                ent = act ? ent : 0x18000u;                              // bit 15: no candidate, bit 16: no entry
                bool cand = ent < 0x8000u;
                const uint32_t p = ent & 0x7FFFu;
                uint32_t r, rk;
                where(p, r, rk);
                if (OFFS && hbase == 0xFFFFFFFFu && rk == 0u) cand = false;          // (a byte in front of off[0]: belongs to no haystack)
                const uint32_t L = r + 1 < LL[e] ? r + 1 : LL[e];
                pp[e] = STARTQ ? ent | (rk << 17) : ent;                 // (the rank: at most TPOS / 8 starts per tile)
                rr[e] = STARTQ ? (act ? r : 1u) : e0 + p;                // offset batches: the index in the haystack; else the global position
                LL[e] = cand ? L : 0u;
                XX[e] = P.window(HP + p);
                // the cell of the C newest symbols as they stand: what it says about depths <= L does not depend on
                // the older ones, and nothing deeper is asked when L <= C
                hc[e] = *(const u32x2*)((const uint8_t*)a.hot + (code_n(XX[e], Cn) << 3));   // (32-bit offset)
            )
            wave_sync();                                                 // (the queue's memory is free from here on)
            PH(2);
            // (the hot cells are waited for here; loads return in order, so the bytes of the next tile — requested a tile earlier — have
            //  arrived as well, and the compiler is told so: left alone it waits for them where they are used, behind the record stores of
            //  the tile's last round, i.e. for the stores' completion)
#pragma unroll
            for (int e = 0; e < NE; e++) asm volatile("" : "+v"(hc[e]));
#pragma unroll
            for (int j = 0; j < (int)DPL / 4; j++) asm volatile("" : "+v"(wnext[j]));
            // 2. top levels: how many keys end here, the value of the shallowest, whether the walk goes deeper
            uint32_t n_go = 0, gomask = 0, tvany = 0;
            uint32_t tvm[NE];
#pragma unroll
            for (int e = 0; e < NE; e++) tvm[e] = 0;
            PPM_SLOTS(e,
                const uint32_t hw = hc[e].x, hx = hc[e].y, L = LL[e];
                const uint32_t m = hw & cmask & ((1u << (L < 16u ? L : 16u)) - 1u);
                cn[e] = (uint32_t)__popc(m);
                // (0 / 1 words combined with & and |: a short-circuit && would put a divergent branch into every slot)
                uint32_t g, has_id;
                if (SB == 2) {                                           // the cell knows which children are keys and which grandchildren exist; both symbols are in X
                    has_id = (hw >> 12) != 0u ? 1u : 0u;
                    const uint32_t t = __builtin_amdgcn_ubfe(XX[e], (32u - 2u * (Cn + 2u)) & 31u, 4u), s1 = t >> 2;
                    const uint32_t kk = hw >> (12u + s1), gk = (L > Cn + 1u ? hw : 0u) >> (16u + t);
                    g = (L > Cn ? 1u : 0u) & (kk | gk) & 1u;
                } else { has_id = hw >> 31; g = has_id & (L > Cn ? 1u : 0u); }
                gomask |= g << e; n_go += g;
                va[e] = (int32_t)hx;                                     // (the shallowest key's value, unless the word holds the id)
                tvm[e] = cn[e] + has_id >= 2u ? m : 0u;                  // (a second key, or one key and the word holds the id)
                tvany |= tvm[e];
            )
            // values that the cell does not hold: its second word is the id, or a second key ends here
            if (__any(tvany != 0u)) {
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    if (tvm[e]) {
                        const uint32_t d1 = (uint32_t)__ffs(tvm[e]), m2 = tvm[e] & (tvm[e] - 1u);
                        va[e] = a.top_val[s_tb[d1] + code_n(XX[e], d1)];
                        if (m2) { const uint32_t d2 = (uint32_t)__ffs(m2); set_vb((uint32_t)e, a.top_val[s_tb[d2] + code_n(XX[e], d2)]); }
                    }
                }
            }
            PH(3);
            // 3. deeper levels: the entries that go on, 64 at a time, one per lane.  One 16-byte record per step (rows:
            // indexed by the next symbol; singles); every walker runs every iteration with selects instead of branches
            // (one that is done re-reads record 0).
            uint32_t n_deep;
            const uint32_t d_base = wave_excl_scan(n_go, n_deep);
#ifdef ACX_PPM_DEV
            if (a.dbg & 32u) n_deep = 0;                                  // timing / traffic only: no deeper walks
            if (a.dbg & 64u) { _Pragma("unroll") for (int e = 0; e < NE; e++) hc[e] = *(const u32x2*)((const uint8_t*)a.hot + ((code_n(XX[e], Cn) & 7u) << 3)); }   // (traffic only: the hot cells once more, from one line)
#endif
            uint32_t* const dq = (uint32_t*)queue;                       // [0..127]: {staged position | L << 12, id} -> {first, second value}; then 64 counts
            uint16_t* const dcnt = (uint16_t*)(dq + 130);              // (dq[128..129]: the dump slot)
            for (uint32_t d0 = 0; d0 < n_deep; d0 += 64u) {
                {
                    uint32_t rnk = d_base - d0;                          // (unsigned: ranks below d0 wrap far beyond 64)
                    PPM_SLOTS(e,
                        const uint32_t g = (gomask >> e) & 1u;
                        const uint32_t slot = (g != 0u && rnk < 64u) ? rnk : 64u;       // (slot 64: nobody reads it)
                        u32x2 v; v.x = (HP + (pp[e] & 0x7FFFu)) | (LL[e] << 12); v.y = (uint32_t)hc[e].y;
                        *(u32x2*)(dq + 2 * slot) = v;
                        rnk += g;
                    )
                }
                wave_sync();
                {
                    bool go = d0 + (uint32_t)lane < n_deep;
                    const uint32_t pk = go ? dq[2 * lane] : HP;
                    uint32_t did = go ? dq[2 * lane + 1] : 0u;
                    const uint32_t wpq = pk & 0xFFFu, wL = pk >> 12;
                    uint32_t dd = Cn, wc = 0;
                    int32_t wa = 0, wb = 0;
                    uint32_t s1 = P.sym_at(wpq - Cn);
                    for (;;) {
                        const uint32_t single = did >> 31, first = single ^ 1u;
                        uint32_t off = single ? a.single_off + (did << 4) : a.row_off + ((did + s1) << 4);   // (bit 31 shifts out; a row's id is a record index)
                        off = go ? off : a.row_off;
                        const u32x4 rec = *(const u32x4*)(a.deep_base + off);
                        const uint32_t g = go ? 1u : 0u;
                        const uint32_t len = rec.y & 0xFFu;
                        const uint32_t dn = dd + first + len;
                        const uint32_t wq = go ? wpq - dd - first : HP;      // (a walker that is done reads a harmless window)
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
                        if (!__any(go)) break;
                        s1 = go ? P.sym_at(wpq - dd) : 0u;
                    }
                    dq[2 * lane] = (uint32_t)wa; dq[2 * lane + 1] = (uint32_t)wb; dcnt[lane] = (uint16_t)wc;
                }
                wave_sync();
                {
                    uint32_t rnk = d_base - d0;
                    uint32_t found = 0;                                  // slots whose walk found keys: their values are fetched below (rare)
                    PPM_SLOTS(e,
                        const uint32_t g = (gomask >> e) & 1u;
                        const bool mine = g != 0u && rnk < 64u;
                        const uint32_t c2 = mine ? (uint32_t)dcnt[mine ? rnk : 64u] : 0u;
                        found |= (c2 ? 1u : 0u) << e;
                        cn[e] += c2 << 16;                               // (kept apart until the values are in place)
                        rnk += g;
                    )
                    if (__any(found != 0u)) {
                        uint32_t rnk2 = d_base - d0;
                        PPM_SLOTS(e,
                            const uint32_t g = (gomask >> e) & 1u;
                            if ((found >> e) & 1u) {
                                const u32x2 dv = *(const u32x2*)(dq + 2 * rnk2);
                                const uint32_t ct = cn[e] & 0xFFFFu;
                                if (ct == 0u) { va[e] = (int32_t)dv.x; if ((cn[e] >> 16) > 1u) set_vb((uint32_t)e, (int32_t)dv.y); } else if (ct == 1u) set_vb((uint32_t)e, (int32_t)dv.x);
                            }
                            rnk2 += g;
                        )
                    }
#pragma unroll
                    for (int e = 0; e < NE; e++) cn[e] = (cn[e] & 0xFFFFu) + (cn[e] >> 16);
                }
                wave_sync();
            }
            PH(4);
            // 4. place: entry e * 64 + lane; the records of a slot follow those of the slots below it
            // (two slots per prefix sum, 16 bits each: a slot has at most 64 x longest < 65536 records)
            uint32_t ex[NE], rt = 0;
            static_assert(NE % 2 == 0, "slots are placed in pairs");
#pragma unroll
            for (int e = 0; e < NE; e += 2) {
                ex[e] = rt; ex[e + 1] = rt;
                if (e == 0 || (uint32_t)e < k) {
                    uint32_t t;
                    const uint32_t x2 = wave_excl_scan(cn[e] | (cn[e + 1] << 16), t);
                    ex[e] = rt + (x2 & 0xFFFFu); rt += t & 0xFFFFu;
                    ex[e + 1] = rt + (x2 >> 16); rt += t >> 16;
                }
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
#ifdef ACX_PPM_DEV
            const bool wr = rt && !dead && !(a.dbg & 2u);                // timing only: no record writes
#else
            const bool wr = rt && !dead;
#endif
            // 5. the record offsets of the haystacks that start here (and the caller's index bases).  Stores are not
            // predicated: a lane with nothing to say writes to a spare element (divergent branches cost more than that)
            if (STARTQ) {
                uint32_t startany = 0;
#pragma unroll
                for (int e = 0; e < NE; e++) startany |= rr[e] == 0u ? 1u : 0u;       // (no entry: rr = 1)
                if (__any(startany != 0u) || a.index_base || a.skip) {
                    const uint32_t dump = (uint32_t)a.n_hay;             // hay_local[n_hay]: spare
                    PPM_SLOTS(e,
                        const uint32_t hh = (OFFS ? hbase : h_tile) + (pp[e] >> 17);      // (the rank that step 1 found)
                        a.hay_local[rr[e] == 0u ? hh : dump] = (int32_t)(run_off + ex[e]);   // the records in front of this haystack
                        if (a.index_base) rr[e] += cn[e] ? (uint32_t)a.index_base[hh] : 0u;
                        if (a.skip) rr[e] -= cn[e] ? (uint32_t)a.skip[hh] : 0u;
                    )
                }
            }
            // 6. records, longest key of a position first.  Slot rt of the round (one past its last record; the grant has
            // the room) takes the stores of the lanes that have no first / second record.
            uint32_t slow = 0, slow1 = 0;                                // slots with more than two records (slow1: or two, the second not in vb1): the general enumeration
            uint8_t* const out8 = (uint8_t*)(a.scratch + g_base + g_used);
            uint2* const out = (uint2*)out8;
            if (wr) {
                uint32_t two = 0;                                        // slots with a second record (rare with long keys)
                PPM_SLOTS(e,
                    const uint32_t c = cn[e]; const uint32_t oe = ex[e] + c - 1u;
                    *(uint2*)(out8 + ((c ? oe : rt) << 3)) = make_uint2(rr[e], (uint32_t)va[e]);
                    two |= (c > 1u ? 1u : 0u) << e;
                    slow |= (c > 2u ? 1u : 0u) << e;
                )
                if (__any(two != 0u)) {
                    const uint32_t mine = vbe < (uint32_t)NE ? 1u << vbe : 0u;   // the slot whose second value vb1 holds
                    slow1 = two & ~mine; two &= mine;
                    PPM_SLOTS(e,
                        if ((two >> e) & 1u) *(uint2*)(out8 + ((ex[e] + cn[e] - 2u) << 3)) = make_uint2(rr[e], (uint32_t)vb1);
                    )
                }
                slow |= slow1;
                while (__any(slow != 0u)) {                              // rare: one slot per lane and pass, from the 32-byte cell
                    if (slow) {
                        const uint32_t se = (uint32_t)__ffs(slow) - 1u;
                        slow &= slow - 1u;
                        const uint32_t from = (slow1 >> se) & 1u ? 1u : 2u;
                        typename Ppm<SB, POW2, false>::Ent E;
                        uint32_t oe = 0;
                        E.p = 0; E.X = 0; E.L = 0; E.idx = 0;
#pragma unroll
                        for (int e = 0; e < NE; e++) if (se == (uint32_t)e) { E.p = pp[e] & 0x7FFFu; E.L = LL[e]; E.idx = rr[e]; oe = ex[e] + cn[e] - 1; }
                        E.X = P.window(HP + E.p);                        // (read again: the windows of the round need not stay in registers for this)
                        const u32x4* cell = (const u32x4*)((const uint8_t*)a.cells + (code_n(E.X, Cn) << 5));
                        E.c0 = cell[0]; E.c1 = cell[1];
                        asm volatile("" : "+v"(E.c0), "+v"(E.c1));       // (waited for on the rare path, not where the paths join)
                        const uint32_t idx = E.idx;
                        P.matches(E, from, 0xFFFFFFFFu, [&](uint32_t kk2, int32_t v) { out[oe - kk2] = make_uint2(idx, (uint32_t)v); });
                    }
                }
            }
            if (rt && !dead) g_used += rt;
            run_off += rt;
            qcount = 0;
            wave_sync();
            PH(5);
        };
#undef PPM_SLOTS

        // (one call site of the round: its code exists once)
        uint32_t seg_lo = 0;
        for (;;) {
            if (seg_lo < 64u) {
                const uint32_t ex_lo = seg_lo ? (uint32_t)__builtin_amdgcn_readlane((int)x_ex, (int)seg_lo) : 0u;
                uint32_t seg_hi = 64u, n_seg = x_tot - ex_lo;
                if (n_seg > PPM_QCAP - qcount) { seg_hi = seg_lo + LPS; n_seg = (seg_hi < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)x_ex, (int)seg_hi) : x_tot) - ex_lo; }
                if (n_seg <= PPM_QCAP - qcount) {                        // (else: a round first — a sub-step has at most 256 entries, it fits the empty queue)
                    uint32_t w = ((uint32_t)lane >= seg_lo && (uint32_t)lane < seg_hi) ? xw : 0u;
                    uint32_t j = qcount + (x_ex - ex_lo);
                    const uint32_t lp = PPL * (uint32_t)lane;
#ifdef ACX_PPM_DEV
                    if (a.dbg & 8u) w = 0;                               // timing only: no queue
#endif
                    while (__any(w != 0u)) {
                        if (w) {
                            const uint32_t b = (uint32_t)__builtin_ctz(w);
                            w &= w - 1u;
                            queue[j++] = (uint16_t)((lp + b) | (((pw >> b) & 1u) ? 0u : 0x8000u));     // 0x8000: a start that is no candidate
                        }
                    }
                    qcount += n_seg;
                    seg_lo = seg_hi;
#ifdef ACX_PPM_DEV
                    if (a.dbg & 4u) qcount = 0;                          // timing only: no rounds
#endif
                    wave_sync();
                    if (seg_lo < 64u) continue;
                }
            }
            PH(1);                                                       // starts, scan, push
            if (qcount) do_round();                                      // (the queue is full, or the symbols of this tile are about to move)
            if (seg_lo >= 64u) break;
        }

        // ---- the tail of this tile is the halo of the next ---------------------------------------
        wave_sync();
        {
            uint32_t t = 0, tn = 0;
            if ((uint32_t)lane < HW) t = sym[2 + TW + lane];
            if ((uint32_t)lane < HP / 32) tn = obits[TPOS / 32 + lane];
            wave_sync();
            if ((uint32_t)lane < HW) sym[2 + lane] = t;
            if ((uint32_t)lane < HP / 32) obits[lane] = tn;
        }
        any_prev = any_cur;
        e0 += TPOS;
        if (!OFFS) {
            r_tile += step_r; h_tile += step_q;
            if (r_tile >= stride) { r_tile -= stride; h_tile++; }
        }
        wave_sync();
    }
    if (lane == 0) {
        desc[0] = run_off; desc[1] = ng; if (ng) desc[18 + ng - 1] = g_used;
        // fixed-stride batches: the records of the 16 waves of this block, summed where k_ppm_gather_pos finds them (no
        // launch for a prefix sum over the waves: every block of the gather adds up the 256 block sums in front of its wave)
        if (a.block_sum && run_off) __hip_atomic_fetch_add(a.block_sum + blockIdx.x, run_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef ACX_PPM_PHASES
    PH(7);
    if (lane == 0 && a.phase_out) for (int i = 0; i < 8; i++) atomicAdd(a.phase_out + i, ph[i]);
#endif
}

// bit p of `bits` = a haystack starts at byte p (k_ppm_stream4 on an offsets batch; the words were zeroed in front of this launch).  The batch's end is
// no start: off[n_hay] is left out, and so is every offset at or beyond it (empty haystacks at the end)
__global__ void __launch_bounds__(256) k_ppm_start_bits(const int64_t* off, int64_t n_hay, uint32_t* bits, uint64_t n_bits) {
    const int64_t n_threads = (int64_t)gridDim.x * 256;
    const uint64_t end = (uint64_t)off[n_hay];
    // (four offsets in flight per thread: the kernel stands in front of the scan kernel, and its time is the latency of its loads and atomics)
    for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < n_hay; h += 4 * n_threads) {
        uint64_t p[4];
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = h + j * n_threads < n_hay ? (uint64_t)off[h + j * n_threads] : ~0ull;
#pragma unroll
        for (int j = 0; j < 4; j++) if (p[j] < end && p[j] < n_bits) atomicOr(bits + (p[j] >> 5), 1u << (p[j] & 31u));
    }
}

// first_h[t] = the first haystack that starts at or after byte t * tile_pos (n_hay + 1 entries of `off`; the last
// one, the end of the batch, counts): one binary search per tile
__global__ void __launch_bounds__(256) k_ppm_first_h(const int64_t* off, int64_t n_hay, int64_t n_tiles, int64_t tile_pos, int64_t* first_h) {
    const int64_t n_threads = (int64_t)gridDim.x * 256;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t <= n_tiles; t += n_threads) {
        const int64_t x = t * tile_pos;
        int64_t lo = 0, hi = n_hay + 1;                                // smallest h in [0, n_hay + 1) with off[h] >= x, else n_hay + 1
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (off[mid] >= x) hi = mid; else lo = mid + 1; }
        first_h[t] = lo;
    }
}

// exclusive prefix sum of the waves' record counts: wave_off[w], wave_off[n_waves] = total
// ctl / host_words (nullable): as k_ppm_gather_pos does for fixed-stride scans — the total and the two flags into the host's pinned words,
// the control words back to zero for the result's next scan: no copies behind the gather, no memset in front of the next scan kernel
__global__ void __launch_bounds__(1024) k_ppm_wave_scan(const uint32_t* wave_desc, int64_t n_waves, int64_t* wave_off, unsigned long long* ctl, long long* host_words) {
    __shared__ int64_t s_part[16];
    __shared__ int64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int64_t b = 0; b < n_waves; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const int64_t v = i < n_waves ? (int64_t)wave_desc[(size_t)i * PPM_DESC_WORDS] : 0;
        int64_t inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int64_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_part[wid] = inc;
        __syncthreads();
        int64_t before = s_carry, all = 0;
        for (int w = 0; w < 16; w++) { if (w < wid) before += s_part[w]; all += s_part[w]; }
        if (i < n_waves) wave_off[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) wave_off[n_waves] = s_carry;
    if (ctl && host_words) {
        uint32_t f1 = 0, f2 = 0;
        if (threadIdx.x == 0) { f1 = (uint32_t)((const int32_t*)(ctl + 8))[0]; f2 = (uint32_t)((const int32_t*)(ctl + 9))[0]; host_words[0] = s_carry; host_words[1] = (int32_t)f1; host_words[2] = (int32_t)f2; }
        const uint32_t seen = (uint32_t)__shfl((int)(f1 | f2 | 0x100u), 0, 64);       // (a value that depends on the flags lane 0 read: the stores below wait for its loads)
        if (seen && threadIdx.x < 16) ctl[threadIdx.x] = 0ull;
    }
}

// one block per wave of k_ppm_stream: its grants, in order, to matches + wave_off[w]; then match_off[]
__global__ void __launch_bounds__(256) k_ppm_gather(const acx_ppm_gather_args c) {
    const int64_t total = c.wave_off[c.n_waves];
    if (total <= c.capacity) {
        for (int64_t w = blockIdx.x; w < c.n_waves; w += gridDim.x) {
            const uint32_t* d = c.wave_desc + (size_t)w * PPM_DESC_WORDS;
            const uint32_t ng = d[1];
            u32x2* dst = (u32x2*)(c.matches + c.wave_off[w]);
            for (uint32_t g = 0; g < ng; g++) {
                const u32x2* src = (const u32x2*)(c.scratch + d[2 + g]);
                const uint32_t n = d[18 + g];
                // two records per access where the destination allows 16-byte stores (the source only needs dword alignment)
                const uint32_t head = (n && (((uintptr_t)dst >> 3) & 1u)) ? 1u : 0u;
                if (head && threadIdx.x == 0) __builtin_nontemporal_store(__builtin_nontemporal_load(src), dst);
                const uint32_t pairs = (n - head) >> 1;
                const u32x4a* src2 = (const u32x4a*)(src + head);
                u32x4* dst2 = (u32x4*)(dst + head);
                for (uint32_t k = threadIdx.x; k < pairs; k += 256) { const u32x4a v = src2[k]; u32x4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w; __builtin_nontemporal_store(o, dst2 + k); }
                if (((n - head) & 1u) && threadIdx.x == 255) __builtin_nontemporal_store(__builtin_nontemporal_load(src + n - 1), dst + n - 1);
                dst += n;
            }
        }
    }
    const int64_t n_threads = (int64_t)gridDim.x * 256;
    const int tile_shift = 63 - __clzll((unsigned long long)c.tile_pos);       // (tiles are 1024 or 2048 positions; the batch is below 4 GiB)
    const uint32_t tpw = (uint32_t)c.tpw;
    for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h <= c.n_hay; h += n_threads) {
        if (h == c.n_hay) c.match_off[h] = total;
        else {
            const uint32_t tile = (uint32_t)(c.off[h] >> tile_shift), per_block = ACX_PPM_WAVES * tpw;      // the wave whose run holds the haystack's first byte
            const uint32_t blk = tile / per_block;
            c.match_off[h] = c.wave_off[blk * ACX_PPM_WAVES + acx_ppm_tile_slot(tile - blk * per_block, tpw, c.share_a, c.share_b)] + c.hay_local[h];
        }
    }
}

// The same for fixed-stride batches, whose records carry the GLOBAL position of their match: the index inside the
// haystack is position - h * stride (+ the caller's index base), and the record offset of haystack h is the number of
// records in front of position h * stride.  A wave of k_ppm_stream covers the positions [A, B) = its run of tiles, so its
// block here knows every haystack that starts in there: those in front of its first record, those between two records
// of different haystacks (the thread of the later record fills the gap), those behind its last record.
// OFFS: an offsets batch that k_ppm_stream4 scanned — the same records; the haystack of a position is the last one whose offset is at or below it.
template <bool OFFS>
__global__ void __launch_bounds__(PPM_GPOS_THREADS) k_ppm_gather_pos(const acx_ppm_gather_args c) {
    // One WAVE per block and no LDS: the scan kernel of the NEXT batch, beside which this kernel runs, may hold every byte of
    // a CU's LDS (k_ppm_stream4 does), and a block that asks for any — 16 bytes for a reduction — then finds no CU to start on.
    constexpr int GT = PPM_GPOS_THREADS;
    static_assert(GT == 64, "the sums below are wave sums");
    const int n_blocks = (int)(c.n_waves / ACX_PPM_WAVES);
    // sum over the block (the records of one scan are fewer than 2^32: the pool addresses them with 32 bits)
    auto block_add = [&](uint32_t x) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        return x;
    };
    uint32_t total32 = 0;
#pragma unroll 1
    for (int b = threadIdx.x; b < n_blocks; b += GT) total32 += c.block_sum[b];
    const int64_t total = (int64_t)block_add(total32);
    if (blockIdx.x == 0) {
        // what the host reads when this result completes (its pinned words), the control words and the block sums of the
        // result's NEXT scan back to zero: no copies behind this kernel, no memset in front of the next scan kernel
        uint32_t f1 = 0, f2 = 0;
        if (threadIdx.x == 0) { f1 = (uint32_t)((const int32_t*)(c.ctl + 8))[0]; f2 = (uint32_t)((const int32_t*)(c.ctl + 9))[0]; c.host_words[0] = total; c.host_words[1] = (int32_t)f1; c.host_words[2] = (int32_t)f2; }
        const uint32_t seen = (uint32_t)__shfl((int)(f1 | f2 | 0x100u), 0, 64);      // (a value that depends on the flags lane 0 read: the stores below wait for its loads)
        if (seen && threadIdx.x < 16) c.ctl[threadIdx.x] = 0ull;
        for (int b = threadIdx.x; b < ACX_PPM_MAX_BLOCKS; b += GT) c.block_sum_next[b] = 0u;
    }
    const bool fits = total <= c.capacity;
    const uint32_t stride = OFFS ? 1u : (uint32_t)c.stride;
    const uint64_t H = OFFS ? (uint64_t)c.off[c.n_hay] : (uint64_t)c.n_hay * stride;
    const int lane = threadIdx.x & 63;
    // positions of one wave lie within tpw * tile_pos of its first: the haystack of a position is a 32-bit multiply-high
    // away (exact while (offset in its haystack + distance) * stride < 2^32), else the 64-bit magic
    const uint64_t span = ((uint64_t)c.tpw + c.share_a) * (uint64_t)c.tile_pos;       // (the longest run of a wave)
    const bool small = !OFFS && (span + stride) * (uint64_t)stride < ((uint64_t)1 << 32);
    const uint32_t m32 = small ? (uint32_t)((((uint64_t)1 << 32) + stride - 1) / stride) : 0u;
    for (int64_t w = blockIdx.x; w < c.n_waves; w += gridDim.x) {
        const uint32_t* d = c.wave_desc + (size_t)w * PPM_DESC_WORDS;
        const uint32_t ng = d[1], count = d[0];
        // the run of wave w of the scan kernel (acx_ppm_slot_first_tile: the waves of a block take unequal runs)
        const uint64_t blk_first = (uint64_t)(w / ACX_PPM_WAVES) * ACX_PPM_WAVES * (uint64_t)c.tpw;
        const uint32_t slot = (uint32_t)(w % ACX_PPM_WAVES);
        uint64_t A = (blk_first + acx_ppm_slot_first_tile(slot, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        uint64_t B = (blk_first + acx_ppm_slot_first_tile(slot + 1u, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        if (A > H) A = H;
        if (B > H) B = H;
        int64_t hA, hB, h0;                                            // haystacks that start in [A, B): hA .. hB - 1; the haystack of position A
        if (OFFS && c.used_bits) {
            // the scan is over: its start bitmap back to zero, every block the words of its wave's run (16 bytes per lane and store)
            const uint64_t w0 = (blk_first + acx_ppm_slot_first_tile(slot, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)(c.tile_pos >> 5);
            uint64_t w1 = (blk_first + acx_ppm_slot_first_tile(slot + 1u, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)(c.tile_pos >> 5);
            if (w1 > c.used_words) w1 = c.used_words;
            u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
            for (uint64_t i = w0 + 4u * (uint32_t)lane; i < w1; i += 4u * GT) *(u32x4*)(c.used_bits + i) = z;
        }
        if (OFFS) {
            // (a haystack that starts at the batch's end — an empty one — belongs to the wave whose run reaches the end)
            // (the first haystack that starts at or behind a position: a binary search over the offsets, twice per wave)
            auto first_at = [&](uint64_t x) -> int64_t { int64_t lo = 0, hi = c.n_hay; while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint64_t)c.off[mid] >= x) hi = mid; else lo = mid + 1; } return lo; };
            hA = A >= H ? c.n_hay : first_at(A);
            hB = B >= H ? c.n_hay : first_at(B);
            h0 = (hA < c.n_hay && (uint64_t)c.off[hA] == A) ? hA : hA - 1;
            if (A >= H) h0 = c.n_hay - 1;
        } else {
            hA = (int64_t)((A + stride - 1) / stride); hB = (int64_t)((B + stride - 1) / stride);
            h0 = (int64_t)(A / stride);
        }
        const uint32_t rA = OFFS ? 0u : (uint32_t)(A - (uint64_t)h0 * stride), A32 = (uint32_t)A;
        // haystack (relative to h0) and index of a position of this wave
        auto locate = [&](uint32_t gpos, uint32_t& idx) -> uint32_t {
            if (OFFS) {
                int64_t lo = 0, hi = c.n_hay;                           // the smallest h with off[h] > gpos
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint64_t)c.off[mid] > (uint64_t)gpos) hi = mid; else lo = mid + 1; }
                const int64_t h = lo - 1;
                idx = gpos - (uint32_t)c.off[h];
                return (uint32_t)(h - h0);
            }
            if (small) { const uint32_t y = rA + (gpos - A32), q = __umulhi(y, m32); idx = y - q * stride; return q; }
            uint32_t rem; const uint32_t hh = div_magic(gpos, c.stride_magic, stride, rem); idx = rem; return (uint32_t)((int64_t)hh - h0);
        };
        uint32_t part = 0;                                             // records of the waves in front of w: whole blocks, then the waves of its own
        const int wb = (int)(w / ACX_PPM_WAVES);
#pragma unroll 1
        for (int b = threadIdx.x; b < wb; b += GT) part += c.block_sum[b];
        if (threadIdx.x < (int)(w % ACX_PPM_WAVES)) part += c.wave_desc[((size_t)wb * ACX_PPM_WAVES + threadIdx.x) * PPM_DESC_WORDS];
        const int64_t base = (int64_t)block_add(part);
        u32x2* dst = (u32x2*)(c.matches + base);
        uint32_t li = 0;                                               // records of this wave in front of the current grant
        uint32_t q_last = (uint32_t)(hA - 1 - h0);                     // haystack (relative) of the last record so far; none: the one in front of the first start
        const uint32_t q_max = (uint32_t)((hB > hA ? hB - 1 : hA - 1) - h0);     // the last haystack a record of this wave can belong to
        if (OFFS) {
            // One record per lane and trip.  The records of a wave are in position order, so the haystack of a record is at or behind that
            // of the record before it: the trip loads the 64 offsets from the haystack of the LAST trip's last record on (one coalesced load),
            // and a lane finds its record's haystack among them by six lane exchanges (positions and offsets fit 32 bits: the scan kernel
            // asked for that).  A record beyond those 64 haystacks (sparse matches) searches the offsets behind them.  The haystack of the record before: the lane below (lane 0: the last trip's last record).
            // (the 64 offsets are kept while the records stay within their first half — some ten trips for config 2's density —, and a trip's
            //  records are requested one trip ahead: most trips wait for no load that depends on the trip before)
            int64_t hw = -1;                                             // the haystack of W's lane 0; -1: none loaded
            uint32_t W = 0;
            for (uint32_t g = 0; g < ng && fits; g++) {
                const u32x2* src = (const u32x2*)(c.scratch + d[2 + g]);
                const uint32_t n = d[18 + g];
                u32x2 nxt; nxt.x = 0xFFFFFFFFu; nxt.y = 0u;
                if ((uint32_t)lane < n) nxt = __builtin_nontemporal_load(src + lane);
#pragma unroll 1
                for (uint32_t k0 = 0; k0 < n; k0 += GT) {
                    const uint32_t k = k0 + (uint32_t)lane;
                    const bool valid = k < n;
                    const u32x2 rec = nxt;
                    nxt.x = 0xFFFFFFFFu; nxt.y = 0u;
                    if (k + GT < n) nxt = __builtin_nontemporal_load(src + k + GT);
                    int64_t hq = h0 + (int64_t)(int32_t)q_last;           // (wave-uniform) the haystack of the record in front
                    if (hq < 0) hq = 0;
                    if (hw < 0 || hq - hw >= 32) {
                        hw = hq;
                        const int64_t wi = hw + lane;
                        W = (uint32_t)c.off[wi < c.n_hay ? wi : c.n_hay];
                    }
                    uint32_t j = 0;                                      // the largest j with W[j] <= position (W[0] is: positions only grow)
#pragma unroll
                    for (uint32_t step = 32; step >= 1; step >>= 1) { const uint32_t t = j + step; const uint32_t wv = (uint32_t)__shfl((int)W, (int)t, 64); if (wv <= rec.x) j = t; }
                    int64_t h = hw + j;
                    uint32_t oh = (uint32_t)__shfl((int)W, (int)j, 64);
                    if (__any(valid && j == 63u)) {                      // maybe beyond the 64 haystacks: the offsets behind them
                        if (valid && j == 63u) {
                            int64_t lo = h + 1, hi = c.n_hay;
                            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((uint64_t)c.off[mid] > (uint64_t)rec.x) hi = mid; else lo = mid + 1; }
                            h = lo - 1; oh = (uint32_t)c.off[h];
                        }
                    }
                    if (h > c.n_hay - 1) h = c.n_hay - 1;                 // (lanes beyond the grant's end)
                    if (h > h0 + (int64_t)(int32_t)q_max) h = h0 + (int64_t)(int32_t)q_max;   // (records of a scan whose pool ran out: see emit)
                    const uint32_t q = (uint32_t)(h - h0);
                    uint32_t qp = (uint32_t)__shfl_up((int)q, 1, 64);
                    if (lane == 0) qp = q_last;
                    if (valid) {
                        for (uint32_t qq = qp + 1; (int32_t)(qq - q) <= 0; qq++) c.match_off[h0 + (int64_t)(int32_t)qq] = c.off_base + base + li + k;   // haystacks that start between the two records
                        u32x2 o; o.y = rec.y;
                        o.x = rec.x - oh + (c.index_base ? (uint32_t)c.index_base[h] : 0u);
                        __builtin_nontemporal_store(o, dst + li + k);
                    }
                    const uint32_t nv = n - k0 < (uint32_t)GT ? n - k0 : (uint32_t)GT;
                    q_last = (uint32_t)__shfl((int)q, (int)(nv - 1u), 64);
                }
                li += n;
            }
        } else
        for (uint32_t g = 0; g < ng && fits; g++) {
            const u32x2* src = (const u32x2*)(c.scratch + d[2 + g]);
            const uint32_t n = d[18 + g];
            // two records per access where the destination allows 16-byte stores (the source only needs dword alignment)
            const uint32_t head = (n && (((uintptr_t)(dst + li) >> 3) & 1u)) ? 1u : 0u;
            const uint32_t pairs = (n - head) >> 1, tailn = (n - head) & 1u;
            const u32x4a* src2 = (const u32x4a*)(src + head);
            u32x4* dst2 = (u32x4*)(dst + li + head);
            auto emit = [&](uint32_t k, u32x2 rec, uint32_t q_prev) -> u32x2 {      // record k of the grant; q_prev: haystack of the record before it
                uint32_t idx;
                uint32_t q = locate(rec.x, idx);
                // (a wave's records belong to the haystacks up to the last one that starts in its run.  Records of a scan whose pool ran out are
                //  whatever the pool's memory held — the host scans again, but this kernel has run by then: no record may send it outside match_off)
                if ((int32_t)q > (int32_t)q_max) q = q_max;
                for (uint32_t qq = q_prev + 1; (int32_t)(qq - q) <= 0; qq++) c.match_off[h0 + (int64_t)(int32_t)qq] = c.off_base + base + li + k;   // haystacks that start between the two records
                u32x2 o; o.y = rec.y;
                o.x = idx + (c.index_base ? (uint32_t)c.index_base[h0 + (int64_t)(int32_t)q] : 0u) - (c.skip ? (uint32_t)c.skip[h0 + (int64_t)(int32_t)q] : 0u);
                return o;
            };
            auto q_of = [&](uint32_t k) -> uint32_t { uint32_t t; const uint32_t qv = locate(src[k].x, t); return (int32_t)qv > (int32_t)q_max ? q_max : qv; };
            if (head && threadIdx.x == 0) __builtin_nontemporal_store(emit(0, __builtin_nontemporal_load(src), q_last), dst + li);
            for (uint32_t k0 = 0; k0 < pairs; k0 += GT) {
                const uint32_t k = k0 + threadIdx.x;
                if (k < pairs) {
                    const u32x4a v = src2[k];
                    const uint32_t r = head + 2 * k;                   // index of the pair's first record in the grant
                    u32x2 r0; r0.x = v.x; r0.y = v.y;
                    u32x2 r1; r1.x = v.z; r1.y = v.w;
                    const uint32_t qp = r ? q_of(r - 1) : q_last;      // (the neighbour's second record: an L2 hit)
                    uint32_t t0;
                    uint32_t q0 = locate(r0.x, t0);
                    if ((int32_t)q0 > (int32_t)q_max) q0 = q_max;
                    const u32x2 o0 = emit(r, r0, qp), o1 = emit(r + 1, r1, q0);
                    u32x4 o; o.x = o0.x; o.y = o0.y; o.z = o1.x; o.w = o1.y;
                    __builtin_nontemporal_store(o, dst2 + k);
                }
            }
            if (tailn && threadIdx.x == GT - 1) __builtin_nontemporal_store(emit(n - 1, __builtin_nontemporal_load(src + n - 1), n > 1 ? q_of(n - 2) : q_last), dst + li + n - 1);
            if (n) q_last = q_of(n - 1);
            li += n;
        }
        (void)lane;
        // haystacks behind the last record (all of them, when the wave has none or nothing fits)
        const int64_t h_last = fits ? h0 + (int64_t)(int32_t)q_last : hA - 1;
        for (int64_t hh = h_last + 1 + threadIdx.x; hh < hB; hh += GT) c.match_off[hh] = c.off_base + base + (fits ? count : 0u);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) c.match_off[c.n_hay] = c.off_base + total;
}

// k_ppm_gather_pos for the usual fixed-stride scan (no index base, no stream contexts, positions of a wave a 32-bit multiply-high
// away from their haystack), written for REGISTERS: k_ppm_stream4 holds 4 x 120 of a SIMD's 512 registers per lane, and a wave
// that needs more than 32 finds no room beside it — the gather of scan k then runs in the holes the scans k + 1, k + 2 leave
// (profiles/r5e_c2_timeline_3x3.txt: it waits 200 us, and every third of a step no scan kernel runs at all).  One record per
// lane and trip (8-byte accesses, a wave's 512 bytes contiguous), everything wave-uniform in scalar registers.
__global__ void __launch_bounds__(PPM_GPOS_THREADS) k_ppm_gather_pos_lean(const acx_ppm_gather_args c) {
    constexpr int GT = PPM_GPOS_THREADS;
    const int n_blocks = (int)(c.n_waves / ACX_PPM_WAVES);
    auto block_add = [&](uint32_t x) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        return x;
    };
    uint32_t total32 = 0;
#pragma unroll 1
    for (int b = threadIdx.x; b < n_blocks; b += GT) total32 += c.block_sum[b];
    const int64_t total = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)block_add(total32));
    if (blockIdx.x == 0) {
        uint32_t f1 = 0, f2 = 0;
        if (threadIdx.x == 0) { f1 = (uint32_t)((const int32_t*)(c.ctl + 8))[0]; f2 = (uint32_t)((const int32_t*)(c.ctl + 9))[0]; c.host_words[0] = total; c.host_words[1] = (int32_t)f1; c.host_words[2] = (int32_t)f2; }
        const uint32_t seen = (uint32_t)__shfl((int)(f1 | f2 | 0x100u), 0, 64);
        if (seen && threadIdx.x < 16) c.ctl[threadIdx.x] = 0ull;
        for (int b = threadIdx.x; b < ACX_PPM_MAX_BLOCKS; b += GT) c.block_sum_next[b] = 0u;
    }
    const bool fits = total <= c.capacity;
    const uint32_t stride = (uint32_t)c.stride;
    const uint64_t H = (uint64_t)c.n_hay * stride;
    const uint32_t m32 = (uint32_t)((((uint64_t)1 << 32) + stride - 1) / stride);
    const uint32_t lane = threadIdx.x;
#pragma unroll 1
    for (int64_t w = blockIdx.x; w < c.n_waves; w += gridDim.x) {
        const uint32_t* d = c.wave_desc + (size_t)w * PPM_DESC_WORDS;
        const uint32_t ng = d[1], count = d[0];
        const uint64_t blk_first = (uint64_t)(w / ACX_PPM_WAVES) * ACX_PPM_WAVES * (uint64_t)c.tpw;
        const uint32_t slot = (uint32_t)(w % ACX_PPM_WAVES);
        uint64_t A = (blk_first + acx_ppm_slot_first_tile(slot, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        uint64_t B = (blk_first + acx_ppm_slot_first_tile(slot + 1u, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        if (A > H) A = H;
        if (B > H) B = H;
        const int64_t hA = (int64_t)((A + stride - 1) / stride), hB = (int64_t)((B + stride - 1) / stride);
        const int64_t h0 = (int64_t)(A / stride);
        const uint32_t bias = (uint32_t)(A - (uint64_t)h0 * stride) - (uint32_t)A;         // offset in its haystack of a position g of this wave, were the haystack h0: g + bias
        uint32_t part = 0;
        const int wb = (int)(w / ACX_PPM_WAVES);
#pragma unroll 1
        for (int b = threadIdx.x; b < wb; b += GT) part += c.block_sum[b];
        if (threadIdx.x < (int)(w % ACX_PPM_WAVES)) part += c.wave_desc[((size_t)wb * ACX_PPM_WAVES + threadIdx.x) * PPM_DESC_WORDS];
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)block_add(part));      // (records of one scan: fewer than 2^32)
        int64_t* const moff = c.match_off + h0;                           // match_off of the haystack of position A
        const int64_t obase = c.off_base + (int64_t)base;
        uint2* const dst = c.matches + base;
        uint32_t li = 0;
        uint32_t q_last = (uint32_t)(hA - 1 - h0);
        const uint32_t q_max = (uint32_t)((hB > hA ? hB - 1 : hA - 1) - h0);     // (no record may send the kernel outside match_off: see k_ppm_gather_pos)
        for (uint32_t g = 0; g < ng && fits; g++) {
            const uint2* src = c.scratch + d[2 + g];
            const uint32_t n = d[18 + g];
#pragma unroll 1
            for (uint32_t k0 = 0; k0 < n; k0 += GT) {
                const uint32_t k = k0 + lane;
                if (k < n) {
                    const uint2 rec = src[k];
                    const uint32_t y = rec.x + bias;
                    uint32_t q = __umulhi(y, m32);
                    if ((int32_t)q > (int32_t)q_max) q = q_max;
                    uint32_t qp = q_last;
                    if (k) { const uint32_t yp = src[k - 1].x + bias; qp = __umulhi(yp, m32); if ((int32_t)qp > (int32_t)q_max) qp = q_max; }
                    for (uint32_t qq = qp + 1; (int32_t)(qq - q) <= 0; qq++) moff[(int32_t)qq] = obase + (int64_t)(li + k);     // haystacks that start between the two records
                    dst[li + k] = make_uint2(y - q * stride, rec.y);
                }
            }
            if (n) q_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)__umulhi(src[n - 1].x + bias, m32));
            li += n;
        }
        const int64_t h_last = fits ? h0 + (int64_t)(int32_t)q_last : hA - 1;
        for (int64_t hh = h_last + 1 + threadIdx.x; hh < hB; hh += GT) c.match_off[hh] = c.off_base + (int64_t)base + (fits ? count : 0u);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) c.match_off[c.n_hay] = c.off_base + total;
}

// records of every tile -> their final place; STRIDE scans: match_off[] from the tile offsets
__global__ void __launch_bounds__(256) k_ppm_compact(const acx_ppm_compact_args c) {
    const int64_t n_items = c.n_items_dev ? *c.n_items_dev : c.n_items;
    const int64_t total = c.item_off[n_items];
    const int sub = threadIdx.x & 15;
    const int64_t n_groups = (int64_t)gridDim.x * 16;
    if (total <= c.capacity) {
        for (int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); i < n_items; i += n_groups) {
            const int32_t n = c.counts[i];
            if (n == 0) continue;
            const uint32_t so = c.scr_off[i];
            if (so == PPM_NOBASE) continue;
            const u32x2* src = (const u32x2*)(c.scratch + so);
            u32x2* dst = (u32x2*)(c.matches + c.item_off[i]);
            for (int32_t k = sub; k < n; k += 16) __builtin_nontemporal_store(__builtin_nontemporal_load(src + k), dst + k);
        }
    }
    if (c.hay_local) {
        const int64_t n_threads = (int64_t)gridDim.x * 256;
        for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h <= c.n_hay; h += n_threads) {
            if (h == c.n_hay) c.match_off[h] = total;
            else c.match_off[h] = c.item_off[(h * c.stride) / PPM_TILE] + c.hay_local[h];
        }
    }
}

// dev_skip for kernels that do not know it: the records of haystack h are sorted by end index, those of its context
// (end index below index_base[h] + skip[h]) are a prefix
__global__ void __launch_bounds__(256) k_skip_count(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* index_base,
                                                    int64_t n_hay, int32_t* kept) {
    const int64_t n_threads = (int64_t)gridDim.x * 256;
    for (int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x; h < n_hay; h += n_threads) {
        const int64_t lo0 = match_off[h], hi0 = match_off[h + 1];
        const int32_t lim = skip[h] + (index_base ? index_base[h] : 0);
        int64_t lo = lo0, hi = hi0;                                      // first record with end_index >= lim
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int32_t)matches[mid].x >= lim) hi = mid; else lo = mid + 1; }
        kept[h] = (int32_t)(hi0 - lo);
    }
}
__global__ void __launch_bounds__(256) k_skip_move(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* kept,
                                                   const int64_t* new_off, int64_t n_hay, uint2* dst) {
    const int sub = threadIdx.x & 15;
    const int64_t n_groups = (int64_t)gridDim.x * 16;
    for (int64_t h = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); h < n_hay; h += n_groups) {
        const int32_t n = kept[h], sk = skip[h];
        const uint2* src = matches + (match_off[h + 1] - n);
        uint2* d = dst + new_off[h];
        for (int32_t k = sub; k < n; k += 16) { uint2 r = src[k]; r.x -= (uint32_t)sk; d[k] = r; }
    }
}

int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            g_num_cus = prop.multiProcessorCount;
        else g_num_cus = 256;
    }
    return g_num_cus;
}

}  // namespace

int acx_num_cus() { return num_cus(); }

int64_t acx_ppm_grid_blocks(const acx_ppm_lds& lds, int64_t n_items_bound, uint32_t reserve_cus) {
    const size_t lds_bytes = (size_t)lds.total_words * 4;
    const int bpc = lds_bytes * 2 <= ACX_PPM_LDS_BYTES ? 2 : 1;       // 1024-thread blocks: at most 2 per CU
    int64_t blocks = (n_items_bound + ACX_PPM_WAVES - 1) / ACX_PPM_WAVES;
    int64_t cus = num_cus();
    if ((int64_t)reserve_cus < cus) cus -= reserve_cus;
    const int64_t cap = cus * bpc;
    if (blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

hipError_t acx_launch_ppm_scan(const acx_ppm_args& a, int64_t n_items_bound, hipStream_t s) {
    if (n_items_bound <= 0) return hipSuccess;
    const size_t lds_bytes = (size_t)a.lds.total_words * 4;
    if (lds_bytes > ACX_PPM_LDS_BYTES || a.n_pools == 0) return hipErrorInvalidValue;
    const int64_t blocks = acx_ppm_grid_blocks(a.lds, n_items_bound, a.reserve_cus);
    const bool chunk = a.ck != nullptr;
    auto launch = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(ACX_PPM_BLOCK), lds_bytes, s, a);
        return hipGetLastError();
    };
    if (a.fast) {
        if (acx_ppm_stream4_eligible(a)) return acx_launch_ppm_stream4(a, blocks, s);    // fixed-length haystacks over four letters
        const bool offs = a.off != nullptr, p2 = a.pow2 != 0, ar = a.sym_arith != 0 && a.sym_bits == 2 && p2;
        // entries per lane and round: 6 where the filter lets many positions through (2- and 4-bit symbols: short keys over
        // few letters), 4 with 8-bit symbols (a tile of text or packets holds a few dozen entries; the slot arrays of 6
        // spilled registers there: 17-24 VGPRs, 80-96 bytes of scratch per lane)
#define PPM_NE(SB) ((SB) == 8 ? 4 : ACX_PPM_NE)
#define PPM_S5(SB, N, P2, GG, AR) do { if (offs) return launch(k_ppm_stream<SB, N, P2, true, GG, AR, false, PPM_NE(SB)>); \
            if (a.m24) return launch(k_ppm_stream<SB, N, P2, false, GG, AR, true, PPM_NE(SB)>); return launch(k_ppm_stream<SB, N, P2, false, GG, AR, false, PPM_NE(SB)>); } while (0)
#define PPM_S(SB, N, GG) do { if (p2) PPM_S5(SB, N, true, GG, false); PPM_S5(SB, N, false, GG, false); } while (0)
        if (a.nsub != 8 && a.nsub != 4) return hipErrorInvalidValue;
#ifdef ACX_PPM_DEV      /* development builds: the config-2 kernel only (compile time) */
        if (a.sym_bits == 2 && a.nsub == 8 && ar && !offs && a.m24) return launch(k_ppm_stream<2, 8, true, false, false, true, true, ACX_PPM_NE>);
#ifdef ACX_PPM_DEV_C4   /* ... and config 4's (a million signatures, offsets batch, global filter + its hashed copy) */
        if (a.sym_bits == 8 && a.g_global && a.nsub == 4 && offs && p2) return launch(k_ppm_stream<8, 4, true, true, true, false, false, 4>);
#endif
#ifdef ACX_PPM_DEV_C3   /* ... and config 3's (100 k text keys, one long haystack as an offsets batch of one, the filter in LDS) */
        if (a.sym_bits == 8 && !a.g_global && a.nsub == 8 && offs && !p2) return launch(k_ppm_stream<8, 8, false, true, false, false, false, 4>);
#endif
        return hipErrorInvalidValue;
#else
        if (a.sym_bits == 8) {
            if (a.g_global) { if (a.nsub == 8) PPM_S(8, 8, true); PPM_S(8, 4, true); }
            if (a.nsub == 8) PPM_S(8, 8, false); PPM_S(8, 4, false);
        }
        if (a.sym_bits == 4) { if (a.nsub == 8) PPM_S(4, 8, false); PPM_S(4, 4, false); }
        if (ar) { if (a.nsub == 8) PPM_S5(2, 8, true, false, true); PPM_S5(2, 4, true, false, true); }
        if (a.nsub == 8) PPM_S(2, 8, false);
        PPM_S(2, 4, false);
#endif
#undef PPM_S
#undef PPM_S5
#undef PPM_NE
    }
#define PPM_CASE(SB) \
    do { \
        if (a.pow2) { if (chunk) return launch(k_ppm_scan<SB, true, true>); return launch(k_ppm_scan<SB, true, false>); } \
        if (chunk) return launch(k_ppm_scan<SB, false, true>); return launch(k_ppm_scan<SB, false, false>); \
    } while (0)
    if (a.sym_bits == 2) PPM_CASE(2);
    if (a.sym_bits == 4) PPM_CASE(4);
    if (a.sym_bits == 8) PPM_CASE(8);
#undef PPM_CASE
    return hipErrorInvalidValue;
}

hipError_t acx_launch_skip_count(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* index_base,
                                 int64_t n_hay, int32_t* kept, hipStream_t s) {
    int64_t blocks = (n_hay + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_skip_count, dim3((unsigned)blocks), dim3(256), 0, s, match_off, matches, skip, index_base, n_hay, kept);
    return hipGetLastError();
}
hipError_t acx_launch_skip_move(const int64_t* match_off, const uint2* matches, const int32_t* skip, const int32_t* kept,
                                const int64_t* new_off, int64_t n_hay, uint2* dst, hipStream_t s) {
    int64_t blocks = (n_hay + 15) / 16;
    const int64_t cap = (int64_t)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_skip_move, dim3((unsigned)blocks), dim3(256), 0, s, match_off, matches, skip, kept, new_off, n_hay, dst);
    return hipGetLastError();
}

hipError_t acx_launch_ppm_first_h(const int64_t* off, int64_t n_hay, int64_t n_tiles, int64_t tile_pos, int64_t* first_h, hipStream_t s) {
    int64_t blocks = (n_tiles + 1 + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_ppm_first_h, dim3((unsigned)blocks), dim3(256), 0, s, off, n_hay, n_tiles, tile_pos, first_h);
    return hipGetLastError();
}

hipError_t acx_launch_ppm_start_bits(const int64_t* off, int64_t n_hay, uint32_t* bits, size_t n_words, bool zero_first, hipStream_t s) {
    if (zero_first) {
        hipError_t e = hipMemsetAsync(bits, 0, n_words * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
    }
    int64_t blocks = (n_hay + 1023) / 1024;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_ppm_start_bits, dim3((unsigned)blocks), dim3(256), 0, s, off, n_hay, bits, (uint64_t)n_words * 32u);
    return hipGetLastError();
}

hipError_t acx_launch_ppm_gather(const uint32_t* wave_desc, int64_t n_waves, int64_t* wave_off, const acx_ppm_gather_args& c, hipStream_t s) {
    if (c.off && !c.pos_records) hipLaunchKernelGGL(k_ppm_wave_scan, dim3(1), dim3(1024), 0, s, wave_desc, n_waves, wave_off, c.ctl, c.host_words);
    int64_t blocks = n_waves;
    const int64_t hb = (c.n_hay + 256) / 256;
    if (blocks < hb) blocks = hb;
    const int64_t cap = (int64_t)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    if (c.off && c.pos_records) {
        const int64_t cap4 = 4 * cap;
        hipLaunchKernelGGL(k_ppm_gather_pos<true>, dim3((unsigned)(n_waves < cap4 ? n_waves : cap4)), dim3(PPM_GPOS_THREADS), 0, s, c);
    } else if (c.off) hipLaunchKernelGGL(k_ppm_gather, dim3((unsigned)blocks), dim3(256), 0, s, c);
    else {
        const int64_t cap4 = 4 * cap;
        // the lean form (at most 32 registers per lane: it runs BESIDE k_ppm_stream4) where it applies: no index base, no contexts,
        // the haystack of a position by a 32-bit multiply-high
        const uint64_t span = ((uint64_t)c.tpw + c.share_a) * (uint64_t)c.tile_pos, st = (uint64_t)c.stride;
#ifdef ACX_LEAN_GATHER                     // (A/B builds, tools/build_variant.sh.  Measured, profiles/r5_experiments.md: beside the scans it takes
                                           //  245 us where the 56-register form takes 50 us in the holes the scans leave: 538.6 against 582.5 GB/s)
        const bool lean = !c.index_base && !c.skip && st > 0 && (span + st) * st < ((uint64_t)1 << 32);
#else
        const bool lean = false; (void)span; (void)st;
#endif
        if (lean) hipLaunchKernelGGL(k_ppm_gather_pos_lean, dim3((unsigned)(n_waves < cap4 ? n_waves : cap4)), dim3(PPM_GPOS_THREADS), 0, s, c);
        else hipLaunchKernelGGL(k_ppm_gather_pos<false>, dim3((unsigned)(n_waves < cap4 ? n_waves : cap4)), dim3(PPM_GPOS_THREADS), 0, s, c);
    }
    return hipGetLastError();
}

hipError_t acx_launch_ppm_compact(const acx_ppm_compact_args& c, int64_t n_items_bound, hipStream_t s) {
    int64_t blocks = (n_items_bound + 15) / 16;
    const int64_t hb = c.hay_local ? (c.n_hay + 256) / 256 : 0;
    if (hb > blocks) blocks = hb;
    const int64_t cap = (int64_t)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_ppm_compact, dim3((unsigned)blocks), dim3(256), 0, s, c);
    return hipGetLastError();
}
