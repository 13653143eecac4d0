// acx_kernels.hip — hand-written CDNA4 (gfx950) kernels of the batch scan.
//
// The hot path of the reference is a serial, dependent pointer chase per input letter
// (automaton_search_iter_next -> ahocorasick_next -> trienode_get_next, then the
// unconditional fail-chain walk of automaton_build_output;
// src/AutomatonSearchIter.c:157-197,243-300, src/trie.c:177-194, src/trienode.c:42-57).
// Here it is re-expressed as three kernels over the flat image (include/acx_blob.h):
//
//   walk    one LANE per haystack, 64 independent DFA walks per wavefront.  Per input
//           byte: one LDS read (byte -> class*4), one v_mad_u32_u24 (state*row_bytes +
//           class*4), ONE 4-byte gather from the fail-resolved table.  The entry carries
//           the target's output count, so counting matches costs a shift+add and no
//           memory access; positions that have outputs append an 8-byte EVENT
//           {end_index, entry} to a per-haystack staging region (capacity = haystack
//           length, so it cannot overflow).  No MFMA: there is no contraction here, the
//           work is latency/transaction bound integer gathers.
//   scan    exclusive prefix sum of the per-haystack match counts -> match_off[].
//   expand  one lane per haystack replays its events through the CSR output lists and
//           writes the final (end_index, value) records at match_off[h] — reference
//           order: position ascending, then the state's fail chain (longest key first).
//
// Haystack bytes are read 16 B per lane per 16 steps straight into registers
// (global_load_dwordx4 at the lane's own address; neighbouring lanes read neighbouring
// reads, every 64-B sector is consumed completely), so the per-byte VMEM budget goes to
// the table gathers, which are what bounds the kernel (L2 transaction rate).
#include "acx_kernels.h"
#include "acx_internal.h"
#include <cstdlib>
#include "acx_blob.h"

#define ACX_WAVE 64
#define ACX_BLOCK 256

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(1))) u32x4_unaligned;
typedef __attribute__((address_space(1))) u32x4_unaligned g_u32x4_unaligned;   // explicitly global (not flat)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// NT = non-temporal (`nt` bit on the instruction): the haystack is read once and the events
// are written once, neither should displace the transition-table rows that live in L2.
template <bool NT>
__device__ __forceinline__ uint4 load16_unaligned(const uint8_t* p) {
    u32x4 v;
    if (NT) v = __builtin_nontemporal_load((const u32x4_unaligned*)p);
    else    v = *(const u32x4_unaligned*)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <bool NT>
__device__ __forceinline__ void store_event(uint2* dst, uint32_t idx, uint32_t entry) {
    u32x2 v; v.x = idx; v.y = entry;
    typedef __attribute__((address_space(1))) u32x2 g_u32x2;        // explicitly global (not flat)
    if (NT) __builtin_nontemporal_store(v, (g_u32x2*)(uintptr_t)dst);
    else    *(g_u32x2*)(uintptr_t)dst = v;
}

// 16 bytes starting at p, never touching bytes at or beyond `limit`
template <bool NT>
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* p, const uint8_t* limit) {
    if (p + 16 <= limit) return load16_unaligned<NT>(p);
    // last 15 bytes of the buffer only: keep this cold path small (no unrolling)
    uint64_t lo = 0, hi = 0;
#pragma nounroll
    for (int i = 0; i < 16 && p + i < limit; i++) {
        const uint64_t v = (uint64_t)p[i] << ((i & 7) * 8);
        if (i < 8) lo |= v; else hi |= v;
    }
    return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
}

// ---------------------------------------------------------------------------------
// walk, ACX_SCAN_ALL
//   ESCAPE : some state has >= 31 outputs, the packed count may be the escape value
//   ILP    : haystacks per lane walked as interleaved, independent dependency chains
//            (2 doubles the table gathers in flight per wave without more waves)
//   EVENTS : false = count only (diagnostic variant: isolates the cost of event stores)
// ---------------------------------------------------------------------------------
struct LaneState {
    uint32_t state;     // current table entry (low 24 bits = state id; v_mad_u32_u24 ignores the rest)
    uint32_t cnt;       // matches so far
    uint2*   ev;        // next free event slot
};

// one table lookup.  SB = 24: the raw previous entry is the state operand (v_mad_u32_u24 only
// reads its low 24 bits) and the byte offset fits 32 bits (saddr + voffset addressing);
// SB = 27: mask, 64-bit address.
template <int SB>
__device__ __forceinline__ uint32_t load_entry(const uint8_t* table_bytes, uint32_t state_raw, uint32_t row_bytes, uint32_t c4) {
    if (SB == ACX_STATE_BITS_NARROW) {
        const uint32_t o = __umul24(state_raw, row_bytes) + c4;        // v_mad_u32_u24
        return *(const uint32_t*)(table_bytes + o);
    } else {
        const uint64_t o = (uint64_t)(state_raw & ACX_ENTRY_STATE_MASK(SB)) * row_bytes + c4;
        return *(const uint32_t*)(table_bytes + o);
    }
}

template <int SB, bool ESCAPE, bool EVENTS, int NT>
__device__ __forceinline__ void step(uint32_t c4, uint32_t idx, const uint8_t* table_bytes,
                                     uint32_t row_bytes, const uint32_t* out_off, LaneState& L) {
    const uint32_t e = load_entry<SB>(table_bytes, L.state, row_bytes, c4);   // the one gather per byte
    L.state = e;
    if (EVENTS || ESCAPE) {
        // count inside the (rarely lane-wide) branch: nothing but the gather, one compare and
        // the branch stays on the common path, and no entry has to be kept live for later
        if (e >> ACX_ENTRY_CNT_SHIFT(SB)) {
            uint32_t c = e >> ACX_ENTRY_CNT_SHIFT(SB);
            if (ESCAPE) {
                if (c == ACX_ENTRY_CNT_ESCAPE(SB)) {
                    const uint32_t s = e & ACX_ENTRY_STATE_MASK(SB);
                    c = out_off[s + 1] - out_off[s];
                }
            }
            if (EVENTS) store_event<(NT & 2) != 0>(L.ev++, idx, e);
            L.cnt += c;
        }
    } else {
        L.cnt += e >> ACX_ENTRY_CNT_SHIFT(SB);
    }
}

// byte -> class*4 for 8 haystack bytes.  These LDS reads do not depend on the state, so
// they are issued back to back ahead of the 8 dependent steps (mad -> gather -> mad ...)
// instead of one LDS round trip per step.
__device__ __forceinline__ void classes8(uint32_t w0, uint32_t w1, const uint32_t* s_cls4, uint32_t (&c4)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) c4[i] = s_cls4[(w0 >> (i * 8)) & 0xffu];
#pragma unroll
    for (int i = 0; i < 4; i++) c4[4 + i] = s_cls4[(w1 >> (i * 8)) & 0xffu];
}

template <int SB, bool ESCAPE, bool EVENTS, int NT, bool GUARD>
__device__ __forceinline__ void block16(const uint4 w, uint32_t idx0, int rem, const uint32_t* s_cls4,
                                        const uint8_t* table_bytes, uint32_t row_bytes, const uint32_t* out_off,
                                        LaneState& L) {
    uint32_t c4[8];
    classes8(w.x, w.y, s_cls4, c4);
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (!GUARD || i < rem) step<SB, ESCAPE, EVENTS, NT>(c4[i], idx0 + i, table_bytes, row_bytes, out_off, L);
    classes8(w.z, w.w, s_cls4, c4);
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (!GUARD || 8 + i < rem) step<SB, ESCAPE, EVENTS, NT>(c4[i], idx0 + 8 + i, table_bytes, row_bytes, out_off, L);
}

// second __launch_bounds__ argument = waves per SIMD the register allocation must allow:
// the kernel is bound by memory transactions, not by occupancy (2 blocks per CU are as fast as 8): 6
// (<= 80 VGPRs) hold the 64 bytes of haystack per visit without spills; 8 (64 VGPRs) spilled 52-60 B and
// was 3 % slower on config 2, the same elsewhere.
#ifndef ACX_PLAIN_WPE
#define ACX_PLAIN_WPE 6
#endif
template <int SB, bool ESCAPE, int ILP, bool EVENTS, int NT>
__global__ void __launch_bounds__(ACX_BLOCK, ILP == 1 ? ACX_PLAIN_WPE : 5) k_walk_all(const acx_walk_args a) {
    __shared__ uint32_t s_cls4[256];
    s_cls4[threadIdx.x] = (uint32_t)a.cls[threadIdx.x] * 4u;   // blockDim.x == 256
    __syncthreads();

    const int lane = threadIdx.x & (ACX_WAVE - 1);
    const int64_t waves_per_block = ACX_BLOCK / ACX_WAVE;
    const int64_t per_task = (int64_t)ACX_WAVE * ILP;
    const int64_t n_tasks = (a.n_hay + per_task - 1) / per_task;
    const int64_t wave0 = (int64_t)blockIdx.x * waves_per_block + (threadIdx.x / ACX_WAVE);
    const int64_t n_waves = (int64_t)gridDim.x * waves_per_block;
    const uint8_t* table_bytes = (const uint8_t*)a.table;
    const uint8_t* limit = a.hay + a.hay_cap;

    for (int64_t task = wave0; task < n_tasks; task += n_waves) {
        int64_t h[ILP];
        bool valid[ILP];
        int len[ILP];
        const uint8_t* p[ILP];
        LaneState L[ILP];
        uint2* ev0[ILP];
        uint32_t base[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            h[q] = task * per_task + (int64_t)q * ACX_WAVE + lane;   // neighbouring lanes, neighbouring reads
            valid[q] = h[q] < a.n_hay;
            int64_t b = 0, e = 0;
            if (valid[q]) {
                if (a.off) { b = a.off[h[q]]; e = a.off[h[q] + 1]; }
                else       { b = h[q] * a.stride; e = b + a.stride; }
            }
            len[q] = (int)(e - b);
            p[q] = a.hay + b;
            L[q].state = (valid[q] && a.init_state) ? (uint32_t)a.init_state[h[q]] : 0u;
            if (L[q].state >= a.n_states) L[q].state = 0u;
            L[q].cnt = 0;
            L[q].ev = a.events + b;
            ev0[q] = L[q].ev;
            base[q] = (valid[q] && a.index_base) ? (uint32_t)a.index_base[h[q]] : 0u;
        }

        for (int j0 = 0;; j0 += 16) {
            int rem[ILP];
            bool any_left = false, all_full = true;
#pragma unroll
            for (int q = 0; q < ILP; q++) {
                rem[q] = len[q] - j0;
                any_left = any_left || rem[q] > 0;
                all_full = all_full && rem[q] >= 16;
            }
            if (!__any(any_left)) break;
            if (ILP > 1 && __all(all_full)) {
                // every chain of every lane has a full block: interleave the chains step by step
                uint4 w[ILP];
#pragma unroll
                for (int q = 0; q < ILP; q++) w[q] = load16_guarded<(NT & 1) != 0>(p[q] + j0, limit);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    uint32_t c4[ILP][8];
#pragma unroll
                    for (int q = 0; q < ILP; q++)
                        classes8(half ? w[q].z : w[q].x, half ? w[q].w : w[q].y, s_cls4, c4[q]);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
#pragma unroll
                        for (int q = 0; q < ILP; q++)
                            step<SB, ESCAPE, EVENTS, NT>(c4[q][i], base[q] + j0 + half * 8 + i, table_bytes, a.row_bytes, a.out_off, L[q]);
                    }
                }
                continue;
            }
            if (ILP == 1 && __all(rem[0] >= 64)) {
                // 64 bytes per visit: a lane streams its own cache line, and one fetched 16 bytes at a
                // time is usually evicted from L2 between visits (profiles/r02_experiments.md)
                uint4 w4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) w4[t] = load16_guarded<(NT & 1) != 0>(p[0] + j0 + 16 * t, limit);
#pragma unroll
                for (int t = 0; t < 4; t++)
                    block16<SB, ESCAPE, EVENTS, NT, false>(w4[t], base[0] + j0 + 16 * t, 16, s_cls4, table_bytes, a.row_bytes, a.out_off, L[0]);
                j0 += 48;
                continue;
            }
#pragma unroll
            for (int q = 0; q < ILP; q++) {
                if (rem[q] > 0) {   // lanes whose haystack is exhausted sit out; __all is over the active lanes
                    const uint4 w = load16_guarded<(NT & 1) != 0>(p[q] + j0, limit);
                    if (__all(rem[q] >= 16)) block16<SB, ESCAPE, EVENTS, NT, false>(w, base[q] + j0, 16, s_cls4, table_bytes, a.row_bytes, a.out_off, L[q]);
                    else                     block16<SB, ESCAPE, EVENTS, NT, true >(w, base[q] + j0, rem[q], s_cls4, table_bytes, a.row_bytes, a.out_off, L[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            if (valid[q]) {
                a.counts[h[q]] = (int32_t)L[q].cnt;
                a.nev[h[q]] = (int32_t)(L[q].ev - ev0[q]);
                if (a.final_state) a.final_state[h[q]] = (int32_t)(L[q].state & ACX_ENTRY_STATE_MASK(SB));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// chunked walk (ACX_SCAN_ALL over long / ragged haystacks)
//
// A lane owns one CHUNK (acx_chunk_desc): it walks `len` bytes from `start`, silently for
// the first `emit` bytes (the left halo of longest_word-1 bytes that rebuilds the state),
// then exactly like k_walk_all.  Chunks of one haystack are consecutive work items, so the
// per-chunk results concatenate to the haystack's result in reference order.
// ---------------------------------------------------------------------------------
template <int SB>
__device__ __forceinline__ void silent_step(uint32_t c4, const uint8_t* table_bytes, uint32_t row_bytes, LaneState& L) {
    L.state = load_entry<SB>(table_bytes, L.state, row_bytes, c4);
}

template <int SB, bool ESCAPE>
__global__ void __launch_bounds__(ACX_BLOCK, 8) k_walk_chunks(const acx_walk_args a, const acx_chunk_desc* ck,
                                                             const int64_t* n_chunks_dev) {
    __shared__ uint32_t s_cls4[256];
    s_cls4[threadIdx.x] = (uint32_t)a.cls[threadIdx.x] * 4u;
    __syncthreads();

    const int lane = threadIdx.x & (ACX_WAVE - 1);
    const int64_t n_chunks = *n_chunks_dev;
    const int64_t n_tasks = (n_chunks + ACX_WAVE - 1) / ACX_WAVE;
    const int64_t wave0 = (int64_t)blockIdx.x * (ACX_BLOCK / ACX_WAVE) + (threadIdx.x / ACX_WAVE);
    const int64_t n_waves = (int64_t)gridDim.x * (ACX_BLOCK / ACX_WAVE);
    const uint8_t* table_bytes = (const uint8_t*)a.table;
    const uint8_t* limit = a.hay + a.hay_cap;

    for (int64_t task = wave0; task < n_tasks; task += n_waves) {
        const int64_t c = task * ACX_WAVE + lane;
        const bool valid = c < n_chunks;
        acx_chunk_desc d;
        d.start = 0; d.emit = 0; d.len = 0; d.idx0 = 0; d.hay = 0; d.flags = 0; d.pad = 0;
        if (valid) d = ck[c];
        const uint8_t* p = a.hay + d.start;
        const int len = d.len, emit = d.emit;
        LaneState L;
        L.state = (valid && a.init_state && (d.flags & 1)) ? (uint32_t)a.init_state[d.hay] : 0u;
        if (L.state >= a.n_states) L.state = 0u;
        L.cnt = 0;
        L.ev = a.events + d.start + emit;
        uint2* const ev0 = L.ev;
        const uint32_t base = (uint32_t)d.idx0;

        for (int j0 = 0;; j0 += 16) {
            const int rem = len - j0;
            if (!__any(rem > 0)) break;
            if (__all(rem >= 64 && j0 >= emit)) {                  // 64 bytes per visit (see k_walk_all)
                uint4 w4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) w4[t] = load16_guarded<false>(p + j0 + 16 * t, limit);
#pragma unroll
                for (int t = 0; t < 4; t++)
                    block16<SB, ESCAPE, true, 2, false>(w4[t], base + j0 + 16 * t, 16, s_cls4, table_bytes, a.row_bytes, a.out_off, L);
                j0 += 48;
                continue;
            }
            if (rem > 0) {
                const uint4 w = load16_guarded<false>(p + j0, limit);
                if (__all(rem >= 16 && j0 >= emit)) {
                    block16<SB, ESCAPE, true, 2, false>(w, base + j0, 16, s_cls4, table_bytes, a.row_bytes, a.out_off, L);
                } else {
                    uint32_t c4[8];
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        classes8(half ? w.z : w.x, half ? w.w : w.y, s_cls4, c4);
#pragma unroll
                        for (int i = 0; i < 8; i++) {
                            const int j = j0 + half * 8 + i;
                            if (half * 8 + i < rem) {
                                if (j >= emit) step<SB, ESCAPE, true, 2>(c4[i], base + j, table_bytes, a.row_bytes, a.out_off, L);
                                else           silent_step<SB>(c4[i], table_bytes, a.row_bytes, L);
                            }
                        }
                    }
                }
            }
        }
        if (valid) {
            a.counts[c] = (int32_t)L.cnt;
            a.nev[c] = (int32_t)(L.ev - ev0);
            if (a.final_state && (d.flags & 2)) a.final_state[d.hay] = (int32_t)(L.state & ACX_ENTRY_STATE_MASK(SB));
        }
    }
}

// ---------------------------------------------------------------------------------
// walk with the IMPLICIT TOP-OF-TRIE in LDS (include/acx_blob.h "itop"), ACX_SCAN_ALL.
//
// The plain walk is bound by one cache transaction per input byte, and most of them miss L2
// (DESIGN.md §4).  Here every lane keeps the rolling history of its last D symbols and reads
// ONE nibble of the LDS-resident table ND4[hist]: how deep the longest trie node (of depth <= D)
// that is a suffix of the history is (D - 0..2: shallower levels are complete), and whether it
// has no, exactly one, or more outputs.
//   depth <  D : the lane holds (depth, hist); its step is that nibble and nothing else.  A node
//                with exactly one output is reported as pseudo state n_states + x (first_val has
//                an entry for it): no memory access; more outputs: packed entry from itop_entry[].
//   depth == D : same, after asking the node's CELL (4 or 8 bytes, indexed by the node's code:
//                a 1 MB array for DNA) whether it has a child on the new symbol.
//   depth >  D : explicit state; gathers its table entry like the plain walk and drops back to
//                (depth from the nibble, hist) when the target is not deeper than D.
// The steady-state step (itop_fast_step: every lane has seen D symbols since its last reset, no
// byte outside the key alphabet in this dword, every step reports) is written for instruction
// count; everything else takes itop_step.  ILP items per lane: their memory operations are
// issued together, before either result is used.
// One 1024-thread block per CU (ND4 is 128 KiB for DNA: b = 2, D = 9); items are haystacks
// (ck == nullptr) or chunks.  oracle/flat_walk.c:flat_iter_itop is the CPU restatement.
// ---------------------------------------------------------------------------------
#define ACX_ITOP_BLOCK 1024
#define ACX_ITOP_EXPL 255u
#define ACX_ITOP_SYM_OTHER 0x80u

struct ItopCtx {
    const uint32_t* ND;         // LDS: 4 bits per history of D symbols
    const uint32_t* Eg;         // global: existence bitmap, sentinel-indexed (warm-up only)
    const uint32_t* ient;       // global: packed entry of implicit node x
    const uint32_t* cells;      // global: child cell of the level-D node with a given code
    const uint32_t* tflags;     // global: per-state entry bits
    const uint8_t*  table_bytes;
    // 32-bit addressing for the steady-state step: table and cells from one base, tflags and the
    // implicit entries from another (the host guarantees each pair lies within 4 GiB)
    const uint8_t*  mem;        // min(table, cells)
    const uint8_t*  aux;        // min(tflags, ient)
    uint32_t toff, coff;        // table (+ 4 * has_other), cells: byte offsets from mem
    uint32_t foff, ioff;        // tflags, ient: byte offsets from aux
    const uint32_t* out_off;
    uint32_t row_bytes, b, D, bD, LD1, has_other, maskD, cs, hmin, pseudo1;
    uint32_t wlim;              // deepest shift a warm step may reach: inside the complete levels, above the shortest key
};

// Events are queued per item (two registers pairs) and written out two at a time: the walk is
// sensitive to the number of write transactions and a 16-byte store is one where
// two 8-byte stores are two; it is also one store instruction for the whole wave, issued when
// some lane has a full pair, instead of one per event.
#define ACX_ITOP_EVQ 2

struct ItopLane {
    uint32_t st;     // explicit (depth > D): raw entry (low 24 bits = state)
    uint32_t sh;     // implicit: b * depth (0 .. b*D); explicit: ACX_ITOP_EXPL
    uint32_t hist;   // last D symbols, b bits each
    uint32_t valid;  // symbols seen since the last reset, saturating at D
    uint32_t cnt;
    uint32_t pend;   // events waiting in q0, q1
    uint32_t q0i, q0e, q1i, q1e;
    uint2*   ev;
    // the event of the previous steady-state step, reported one step late: when its entry had to
    // be fetched (child with outputs, node with several outputs) that load has the whole next
    // step's gather to complete under instead of putting a second memory latency into the step
    uint32_t pf;     // packed entry (count 0: nothing to report); it ended at the byte before
};

// sentinel index of the k-gram made of the last sh/b symbols (sh = 0 -> 1, the root)
__device__ __forceinline__ uint32_t itop_x(uint32_t hist, uint32_t sh) {
    return __builtin_amdgcn_ubfe(hist, 0u, sh) | (1u << sh);
}

// Warm-up (first D steps after a reset): largest shift <= cand whose k-gram is a node, probing
// the global E bitmap level by level.  Shifts up to cs belong to complete levels and hit
// without a probe; shift 0 (the root) always hits.
__device__ __noinline__ uint32_t itop_resolve_slow(uint32_t hist, uint32_t cand, const uint32_t* Eg, uint32_t b, uint32_t cs) {
    uint32_t c = cand;
    for (;;) {
        if (c <= cs) break;
        const uint32_t x = itop_x(hist, c);
        if ((Eg[x >> 5] >> (x & 31)) & 1u) break;
        c -= b;
    }
    return c;
}

// raw cell of the level-D node `code`; decoded by itop_cell_first/bits AFTER the other loads of
// the step have been issued (decoding inside the branch would make the wave wait right there)
template <bool CELL8>
__device__ __forceinline__ uint2 itop_load_cell(const uint32_t* cells, uint32_t code) {
    if (CELL8) return ((const uint2*)cells)[code];
    return make_uint2(cells[code], 0u);
}
template <bool CELL8>
__device__ __forceinline__ uint32_t itop_cell_first(uint2 cw) { return CELL8 ? cw.x : (cw.x & 0xFFFFFFu); }
// child mask in bits 0..15, child-has-outputs in bits 16..31
template <bool CELL8>
__device__ __forceinline__ uint32_t itop_cell_bits(uint2 cw) { return CELL8 ? cw.y : (((cw.x >> 24) & 15u) | ((cw.x >> 28) << 16)); }

template <bool ESCAPE>
__device__ __forceinline__ void itop_report(uint32_t e, uint32_t c, uint32_t idx, const ItopCtx& C, ItopLane& L) {
    if (ESCAPE) {
        if (c == ACX_ENTRY_CNT_ESCAPE(ACX_STATE_BITS_NARROW)) {
            const uint32_t s = e & ACX_ENTRY_STATE_MASK(ACX_STATE_BITS_NARROW);
            c = C.out_off[s + 1] - C.out_off[s];
        }
    }
    // selects, no branch: nearly every step of a wave has SOME lane with an event (c = 0: none).
    // The queue shifts: q1 is the older event of a pair, q0 the newer (or the only) one.
    const bool has = c != 0u;
    L.q1i = has ? L.q0i : L.q1i; L.q1e = has ? L.q0e : L.q1e;
    L.q0i = has ? idx : L.q0i;   L.q0e = has ? e : L.q0e;
    L.pend += has ? 1u : 0u;
    L.cnt += c;
}

// write out the queue of every lane that has a full pair (wave-uniform call)
__device__ __forceinline__ void itop_flush(ItopLane& L) {
    if (L.pend == ACX_ITOP_EVQ) {
        u32x4 v; v.x = L.q1i; v.y = L.q1e; v.z = L.q0i; v.w = L.q0e;
        __builtin_nontemporal_store(v, (g_u32x4_unaligned*)(uintptr_t)L.ev);
        L.ev += ACX_ITOP_EVQ;
        L.pend = 0;
    }
}
// ... and what is left at the end of an item
__device__ __forceinline__ void itop_flush_rest(ItopLane& L) {
    itop_flush(L);
    if (L.pend == 1) {
        store_event<true>(L.ev, L.q0i, L.q0e);
        L.ev += 1;
        L.pend = 0;
    }
}

// report the deferred event of the last steady-state step, if any
template <bool ESCAPE>
__device__ __forceinline__ void itop_drain(uint32_t idx, const ItopCtx& C, ItopLane& L) {
    itop_report<ESCAPE>(L.pf, L.pf >> ACX_ENTRY_CNT_SHIFT(ACX_STATE_BITS_NARROW), idx, C, L);
    L.pf = 0u;
}

// One input byte right after a reset, while the lane is still inside the complete levels and
// above the shortest key: it goes one level down, nothing to look up, nothing to report.
// The caller checks (wave-uniformly) itop_warm_ok for every lane.
__device__ __forceinline__ bool itop_warm_ok(uint32_t sy, bool active, const ItopCtx& C, const ItopLane& L) {
    return active && !(sy & ACX_ITOP_SYM_OTHER) && L.sh + C.b <= C.wlim && __umul24(L.valid, C.b) == L.sh;
}
__device__ __forceinline__ void itop_warm_step(uint32_t sy, const ItopCtx& C, ItopLane& L) {
    L.hist = ((L.hist << C.b) | sy) & C.maskD;
    L.valid += 1u;
    L.sh += C.b;
}

// One input byte, any situation (warm-up, bytes outside the key alphabet, ragged ends, halo).
// sy = symbol of the byte, or ACX_ITOP_SYM_OTHER.
// Warm-up: for the first D - 1 symbols after a reset (item start, or a byte no key contains) the
// history is not full and ND4 does not apply; the lane is at most `valid` deep, and the depth is
// found by probing E downwards from its old depth + 1 (no memory access while that is inside the
// complete levels).  Nodes shallower than the shortest key (C.hmin) have no outputs.
template <bool ESCAPE, bool CELL8>
__device__ __forceinline__ void itop_step(uint32_t sy, uint32_t idx, bool active, bool emit, const ItopCtx& C, ItopLane& L) {
    constexpr int SB = ACX_STATE_BITS_NARROW;
    const bool other = (sy & ACX_ITOP_SYM_OTHER) != 0;              // a byte no key contains: root, no output
    const bool go = active && !other;
    const uint32_t sym = sy & (ACX_ITOP_SYM_OTHER - 1u);
    const uint32_t hist = ((L.hist << C.b) | sym) & C.maskD;
    const bool deep = L.sh == ACX_ITOP_EXPL, atD = L.sh == C.bD;
    uint32_t e_tab = 0;
    uint2 cw = make_uint2(0u, 0u);
    if (go && deep) e_tab = load_entry<SB>(C.table_bytes, L.st, C.row_bytes, (sym + C.has_other) << 2);
    if (go && atD) cw = itop_load_cell<CELL8>(C.cells, L.hist);
    const uint32_t q = (C.ND[hist >> 3] >> ((hist & 7u) << 2)) & 15u;
    const uint32_t cell_first = itop_cell_first<CELL8>(cw), cell_bits = itop_cell_bits<CELL8>(cw);
    const uint32_t valid = L.valid < C.D ? L.valid + 1 : L.valid;
    uint32_t e = 0, new_sh = L.sh, new_st = L.st;
    if (go) {
        bool settled = false, own_out = false;                       // own_out: e already holds the target's outputs
        if (deep) {
            e = e_tab; own_out = true;
            if ((e_tab & ACX_ENTRY_STATE_MASK(SB)) >= C.LD1) { new_st = e_tab; settled = true; }
        } else if (atD && ((cell_bits >> sym) & 1u)) {               // a child on this symbol: depth D + 1
            const uint32_t child = cell_first + (uint32_t)__popc(cell_bits & ((1u << sym) - 1u) & 0xFFFFu);
            new_st = child; new_sh = ACX_ITOP_EXPL; settled = true; own_out = true;
            if (emit && ((cell_bits >> (16 + sym)) & 1u)) e = child | C.tflags[child];
        }
        if (!settled) {                                              // the new state is not deeper than D
            const bool exact = valid >= C.D && (q & 3u) != 3u;       // full history and not a deep fall: ND4 says it all
            if (exact) new_sh = C.bD - __umul24(C.b, q & 3u);
            else new_sh = itop_resolve_slow(hist, valid >= C.D ? C.bD - 3u * C.b : L.sh + C.b, C.Eg, C.b, C.cs);
            if (!own_out && emit && new_sh >= C.hmin && (!exact || (q >> 2))) e = C.ient[itop_x(hist, new_sh)];
        }
    }
    if (active) {
        L.hist = other ? 0u : hist;
        L.valid = other ? 0u : valid;
        L.sh = other ? 0u : new_sh;
        L.st = new_st;
    }
    itop_report<ESCAPE>(emit ? e : 0u, emit ? e >> ACX_ENTRY_CNT_SHIFT(SB) : 0u, idx, C, L);
}

// One input byte for each of the lane's ILP items in the steady state: every lane active and
// reporting, D symbols seen since the last reset, no byte outside the key alphabet.
// The walk is bound by instruction issue as much as by memory latency (four waves per SIMD,
// DESIGN.md §4), so this is written as straight-line selects: the only branches are the deferred
// entry fetch, the flush of a full event pair and (NOESC = false) the probe path.
// prev = end index of the byte before (where the deferred event of the last step belongs).
template <bool ESCAPE, bool CELL8, bool NOESC, int ILP>
__device__ __forceinline__ void itop_fast_step(const uint32_t (&sym)[ILP], const uint32_t (&prev)[ILP], const ItopCtx& C, ItopLane (&L)[ILP]) {
    constexpr int SB = ACX_STATE_BITS_NARROW;
    uint32_t hist[ILP], ndw[ILP], raw0[ILP], raw1[ILP];
    bool deep[ILP];
#pragma unroll
    for (int q = 0; q < ILP; q++) {                                  // issue: ONE load (table entry | cell) and the ND4 word
        deep[q] = L[q].sh > C.bD;                                    // (= ACX_ITOP_EXPL)
        // the two kinds of lanes that need memory share one load with per-lane 32-bit offsets
        const uint32_t moff = deep[q] ? __umul24(L[q].st, C.row_bytes) + (sym[q] << 2) + C.toff
                                      : (L[q].hist << (CELL8 ? 3 : 2)) + C.coff;
        // every lane loads — a lane that needs nothing (depth < D) reads offset 0 and drops it: one
        // select instead of an exec-mask branch around the load (the walk is bound by issue, §4)
        const bool need = L[q].sh >= C.bD;                           // deep, or at depth D
        const uint32_t mo = need ? moff : 0u;
        if (CELL8) { const uint2 v = *(const uint2*)(C.mem + mo); raw0[q] = need ? v.x : 0u; raw1[q] = need ? v.y : 0u; }
        else { const uint32_t v = *(const uint32_t*)(C.mem + mo); raw0[q] = need ? v : 0u; raw1[q] = 0u; }
        hist[q] = ((L[q].hist << C.b) | sym[q]) & C.maskD;
        ndw[q] = C.ND[hist[q] >> 3];
    }
#pragma unroll
    for (int q = 0; q < ILP; q++) itop_drain<ESCAPE>(prev[q], C, L[q]);       // the previous step's event: its fetch is older than this step's load
#pragma unroll
    for (int q = 0; q < ILP; q++) {
        const uint32_t s4 = hist[q] << 2;                            // (the bit-field extract takes the offset modulo 32)
        const uint32_t dq = __builtin_amdgcn_ubfe(ndw[q], s4, 2u);  // depth field
        const uint32_t oc = __builtin_amdgcn_ubfe(ndw[q], s4 + 2u, 2u);   // output class (0 when the depth field escapes)
        uint32_t sh_nd = C.bD - __umul24(C.b, dq);
        const uint32_t ent = deep[q] ? raw0[q] : 0u;                // table entry of a deep lane
        const uint32_t cw0 = deep[q] ? 0u : raw0[q];                // cell of a lane at depth D (0: no cell, no child)
        const bool stay = (ent & ACX_ENTRY_STATE_MASK(SB)) >= C.LD1;
        bool kid, fetch_k;
        uint32_t child;
        if (CELL8) {
            const uint32_t cw1 = deep[q] ? 0u : raw1[q];
            const uint32_t t = cw1 >> sym[q];
            kid = (t & 1u) != 0u;
            fetch_k = (t & 0x10001u) == 0x10001u;
            child = cw0 + (uint32_t)__popc(__builtin_amdgcn_ubfe(cw1, 0u, sym[q]));
        } else {                                                     // first_child[0..23] | mask[24..27] | outs[28..31]; sym < 4
            const uint32_t t = cw0 >> (sym[q] + 24u);
            kid = (t & 1u) != 0u;
            fetch_k = (t & 0x11u) == 0x11u;
            child = (cw0 & 0xFFFFFFu) + (uint32_t)__popc(__builtin_amdgcn_ubfe(cw0, 24u, sym[q]));
        }
        bool esc = false;
        if (!NOESC) {
            esc = dq == 3u && !stay && !kid;                         // fell below D - 2: probe (rare by the choice of D)
            if (esc) sh_nd = itop_resolve_slow(hist[q], C.bD - 3u * C.b, C.Eg, C.b, C.cs);
        }
        const bool down = stay || kid;
        L[q].hist = hist[q];
        L[q].sh = down ? ACX_ITOP_EXPL : sh_nd;
        L[q].st = stay ? ent : child;                                // read only while sh == EXPL
        // outputs: deep lanes carry them in the entry; a shallow node with exactly one output is
        // reported as its pseudo state; the rest (child with outputs, several outputs) fetches a
        // whole entry (tflags[s] carries s), which is consumed by the NEXT step (itop_drain).
        // No output: entry 0 (count 0).
        const uint32_t x = itop_x(hist[q], sh_nd);
        const bool fetch_i = !deep[q] && !kid && (oc == 2u || (!NOESC && esc && sh_nd != 0u));
        const uint32_t fo = fetch_k ? (child << 2) + C.foff : (x << 2) + C.ioff;
        uint32_t ev = deep[q] ? ent : ((!kid && oc == 1u) ? x + C.pseudo1 : 0u);
        if (fetch_k || fetch_i) ev = *(const uint32_t*)(C.aux + fo);  // NOT used in this step
        L[q].pf = ev;
    }
}

#ifndef ACX_ITOP_WPE
#define ACX_ITOP_WPE 4
#endif
template <bool ESCAPE, bool CELL8, bool NOESC, int ILP, int HB>
__global__ void __launch_bounds__(ACX_ITOP_BLOCK, ACX_ITOP_WPE) k_walk_itop(const acx_walk_args a, const acx_chunk_desc* ck,
                                                             const int64_t* n_chunks_dev, const uint32_t* itop_lds,
                                                             uint32_t itop_words, const uint32_t* itop_entry,
                                                             const uint32_t* itop_ebits, const uint32_t* itop_cells,
                                                             const uint32_t* tflags) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    uint32_t* s_sym = s_mem + ((itop_words + 3) & ~3u);              // byte -> symbol, or ACX_ITOP_SYM_OTHER
    for (uint32_t i = threadIdx.x; i < itop_words; i += blockDim.x) s_mem[i] = itop_lds[i];
    __syncthreads();
    if (threadIdx.x < 256) {
        const uint32_t cl = a.cls[threadIdx.x], ho = s_mem[5];
        s_sym[threadIdx.x] = (ho && cl == 0) ? ACX_ITOP_SYM_OTHER : cl - ho;
    }
    __syncthreads();

    ItopCtx C;
    C.mem = (const uint8_t*)a.table < (const uint8_t*)itop_cells ? (const uint8_t*)a.table : (const uint8_t*)itop_cells;
    C.toff = (uint32_t)((const uint8_t*)a.table - C.mem) + 4u * s_mem[5];
    C.coff = (uint32_t)((const uint8_t*)itop_cells - C.mem);
    C.aux = (const uint8_t*)tflags < (const uint8_t*)itop_entry ? (const uint8_t*)tflags : (const uint8_t*)itop_entry;
    C.foff = (uint32_t)((const uint8_t*)tflags - C.aux);
    C.ioff = (uint32_t)((const uint8_t*)itop_entry - C.aux);
    C.ient = itop_entry; C.Eg = itop_ebits; C.cells = itop_cells; C.tflags = tflags; C.table_bytes = (const uint8_t*)a.table; C.out_off = a.out_off; C.row_bytes = a.row_bytes;
    C.b = s_mem[0]; C.D = s_mem[1]; C.bD = s_mem[0] * s_mem[1]; C.LD1 = s_mem[2]; C.has_other = s_mem[5]; C.maskD = s_mem[7];
    C.pseudo1 = s_mem[4] | (1u << ACX_ENTRY_CNT_SHIFT(ACX_STATE_BITS_NARROW));
    C.ND = s_mem + ACX_ITOP_HDR_WORDS;                               // (= s_mem[8]; the launcher checks)
    C.cs = s_mem[11]; C.hmin = s_mem[12];
    C.wlim = C.hmin >= C.b ? (C.cs < C.hmin - C.b ? C.cs : C.hmin - C.b) : 0u;
    if (C.wlim > C.bD - C.b) C.wlim = C.bD - C.b;                      // (a warm step never reaches depth D: valid stays below D)

    const int lane = threadIdx.x & (ACX_WAVE - 1);
    const int64_t n_items = ck ? *n_chunks_dev : a.n_hay;
    const int64_t per_task = (int64_t)ACX_WAVE * ILP;
    const int64_t n_tasks = (n_items + per_task - 1) / per_task;
    // wave numbering is block-minor: the last, partial round of tasks then spreads over ALL CUs
    // (each runs a few waves less) instead of leaving the highest-numbered CUs idle
    const int64_t wave0 = (int64_t)(threadIdx.x / ACX_WAVE) * gridDim.x + blockIdx.x;
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / ACX_WAVE);
    const uint8_t* limit = a.hay + a.hay_cap;

    for (int64_t task = wave0; task < n_tasks; task += n_waves) {
        // per item: where it starts, how long it is, where reporting starts, the index of its first
        // byte (registers are what limits two items per lane: nothing else is kept across the walk)
        const uint8_t* p[ILP];
        int32_t len[ILP], emit[ILP];
        uint32_t idx0[ILP];
        ItopLane L[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            const int64_t item = task * per_task + q * ACX_WAVE + lane;
            int64_t start = 0;
            len[q] = 0; emit[q] = 0; idx0[q] = 0;
            if (item < n_items) {
                if (ck) { const acx_chunk_desc d = ck[item]; start = d.start; len[q] = d.len; emit[q] = d.emit; idx0[q] = (uint32_t)d.idx0; }
                else {
                    const int64_t b0 = a.off ? a.off[item] : item * a.stride;
                    const int64_t e0 = a.off ? a.off[item + 1] : b0 + a.stride;
                    start = b0; len[q] = (int32_t)(e0 - b0);
                    idx0[q] = a.index_base ? (uint32_t)a.index_base[item] : 0u;
                }
            }
            p[q] = a.hay + start;
            L[q].st = 0; L[q].sh = 0; L[q].hist = 0; L[q].valid = 0; L[q].cnt = 0; L[q].pend = 0; L[q].q0i = 0; L[q].q0e = 0; L[q].q1i = 0; L[q].q1e = 0;
            L[q].pf = 0;
            L[q].ev = a.events + start + emit[q];
        }

        // The haystack is fetched HB x 16 bytes per lane at a time: lane-per-haystack means every
        // lane streams its own cache line, a line is re-read 16 bytes at a time ~20 us apart and is
        // usually evicted in between; 64 bytes per visit cut those re-fetches.
        // (Non-temporal loads were measured slower.)
        for (int j0 = 0;; j0 += 16 * HB) {
            bool more = false;
#pragma unroll
            for (int q = 0; q < ILP; q++) more = more || len[q] - j0 > 0;
            if (!__any(more)) break;
            uint4 wq[ILP][HB];
#pragma unroll
            for (int q = 0; q < ILP; q++)
#pragma unroll
                for (int t = 0; t < HB; t++) {
                    wq[q][t] = make_uint4(0, 0, 0, 0);
                    if (len[q] - (j0 + 16 * t) > 0) wq[q][t] = load16_guarded<false>(p[q] + j0 + 16 * t, limit);
                }
#pragma unroll 1
            for (int t = 0; t < HB; t++) {                           // 16-byte blocks; the buffer rotates, indices stay constant
                const int jb = j0 + 16 * t;
                bool blk_more = false;
                uint4 w[ILP];
#pragma unroll
                for (int q = 0; q < ILP; q++) {
                    blk_more = blk_more || len[q] - jb > 0;
                    w[q] = wq[q][0];
#pragma unroll
                    for (int u = 0; u + 1 < HB; u++) wq[q][u] = wq[q][u + 1];
                }
                if (!__any(blk_more)) break;
                // one or two dwords (4 or 8 steps) per iteration, NOT unrolled: the instruction cache is
                // shared.  The per-group overhead (symbols, the steady-state test, the loop) is a tenth
                // of all instructions, so two dwords are taken together whenever both are steady.
#pragma unroll 1
                for (int k = 0; k < 4;) {
                    uint32_t sy[ILP][4];
                    const int jk = jb + k * 4;
                    if (k <= 2) {                                     // (wave-uniform) a second dword in this block
                        uint32_t sz[ILP][4];
                        bool steady8 = true;
#pragma unroll
                        for (int q = 0; q < ILP; q++) {
#pragma unroll
                            for (int i = 0; i < 4; i++) { sy[q][i] = s_sym[(w[q].x >> (i * 8)) & 0xffu]; sz[q][i] = s_sym[(w[q].y >> (i * 8)) & 0xffu]; }
                            steady8 = steady8 && len[q] - jk >= 8 && jk >= emit[q] && L[q].valid >= C.D &&
                                      !((sy[q][0] | sy[q][1] | sy[q][2] | sy[q][3] | sz[q][0] | sz[q][1] | sz[q][2] | sz[q][3]) & ACX_ITOP_SYM_OTHER);
                        }
                        if (__all(steady8)) {
                            uint32_t prev[ILP];
#pragma unroll
                            for (int q = 0; q < ILP; q++) { prev[q] = idx0[q] + (uint32_t)jk - 1u; w[q].x = w[q].z; w[q].y = w[q].w; }
#pragma unroll
                            for (int i = 0; i < 8; i++) {
                                uint32_t s1[ILP], ix[ILP];
#pragma unroll
                                for (int q = 0; q < ILP; q++) { s1[q] = i < 4 ? sy[q][i & 3] : sz[q][i & 3]; ix[q] = idx0[q] + (uint32_t)(jk + i); }
                                itop_fast_step<ESCAPE, CELL8, NOESC, ILP>(s1, prev, C, L);
#pragma unroll
                                for (int q = 0; q < ILP; q++) {
                                    prev[q] = ix[q];
                                    if (__any(L[q].pend == ACX_ITOP_EVQ)) itop_flush(L[q]);
                                }
                            }
                            k += 2;
                            continue;
                        }
                    }
                    bool steady = true;
#pragma unroll
                    for (int q = 0; q < ILP; q++) {
                        const uint32_t wk = w[q].x;                   // the block rotates one dword per iteration
                        w[q].x = w[q].y; w[q].y = w[q].z; w[q].z = w[q].w;
#pragma unroll
                        for (int i = 0; i < 4; i++) sy[q][i] = s_sym[(wk >> (i * 8)) & 0xffu];
                        // four whole steps, all reported, history full, no byte outside the key alphabet
                        steady = steady && len[q] - jk >= 4 && jk >= emit[q] && L[q].valid >= C.D &&
                                 !((sy[q][0] | sy[q][1] | sy[q][2] | sy[q][3]) & ACX_ITOP_SYM_OTHER);
                    }
                    k += 1;
                    if (__all(steady)) {
                        uint32_t prev[ILP];
#pragma unroll
                        for (int q = 0; q < ILP; q++) prev[q] = idx0[q] + (uint32_t)jk - 1u;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            uint32_t s1[ILP], ix[ILP];
#pragma unroll
                            for (int q = 0; q < ILP; q++) { s1[q] = sy[q][i]; ix[q] = idx0[q] + (uint32_t)(jk + i); }
                            itop_fast_step<ESCAPE, CELL8, NOESC, ILP>(s1, prev, C, L);
#pragma unroll
                            for (int q = 0; q < ILP; q++) {
                                prev[q] = ix[q];
                                if (__any(L[q].pend == ACX_ITOP_EVQ)) itop_flush(L[q]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < ILP; q++) {
                            itop_drain<ESCAPE>(idx0[q] + (uint32_t)jk - 1u, C, L[q]);     // (a deferred event belongs to the byte before)
                            if (__any(L[q].pend == ACX_ITOP_EVQ)) itop_flush(L[q]);
#pragma unroll 1
                            for (int i = 0; i < 4; i++) {
                                const int j = jk + i;
                                if (__all(itop_warm_ok(sy[q][i], j < len[q], C, L[q]))) { itop_warm_step(sy[q][i], C, L[q]); continue; }
                                itop_step<ESCAPE, CELL8>(sy[q][i], idx0[q] + (uint32_t)j, j < len[q], j < len[q] && j >= emit[q], C, L[q]);
                                if (__any(L[q].pend == ACX_ITOP_EVQ)) itop_flush(L[q]);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            itop_drain<ESCAPE>(idx0[q] + (uint32_t)len[q] - 1u, C, L[q]);
            itop_flush_rest(L[q]);
        }
#pragma unroll
        for (int q = 0; q < ILP; q++) {
            const int64_t item = task * per_task + q * ACX_WAVE + lane;
            if (item < n_items) {
                a.counts[item] = (int32_t)L[q].cnt;
                a.nev[item] = (int32_t)(L[q].ev - (a.events + (p[q] - a.hay) + emit[q]));
                int32_t hay = (int32_t)item, flags = 3;
                if (ck) { hay = ck[item].hay; flags = ck[item].flags; }
                if (a.final_state && (flags & 2)) {
                    uint32_t fs = 0;
                    if (L[q].sh == ACX_ITOP_EXPL) fs = L[q].st & ACX_ENTRY_STATE_MASK(ACX_STATE_BITS_NARROW);
                    else if (L[q].sh > 0) fs = itop_entry[itop_x(L[q].hist, L[q].sh)] & ACX_ENTRY_STATE_MASK(ACX_STATE_BITS_NARROW);
                    a.final_state[hay] = (int32_t)fs;
                }
            }
        }
    }
}

// chunks per haystack: max(1, ceil(len / CH)) — an empty haystack still gets one (empty)
// chunk so that it owns a final_state and a match_off slot
__global__ void __launch_bounds__(ACX_BLOCK) k_chunk_count(const acx_chunk_args c) {
    const int64_t h = (int64_t)blockIdx.x * ACX_BLOCK + threadIdx.x;
    if (h >= c.n_hay) return;
    const int64_t len = c.off ? c.off[h + 1] - c.off[h] : c.stride;
    const int64_t n = (len + c.chunk_bytes - 1) / c.chunk_bytes;
    c.nck[h] = (int32_t)(n < 1 ? 1 : n);
}

// one thread per chunk: find the owning haystack by binary search in ck_first
__global__ void __launch_bounds__(ACX_BLOCK) k_chunk_fill(const acx_chunk_args c) {
    const int64_t n_chunks = c.ck_first[c.n_hay];
    const int64_t n_threads = (int64_t)gridDim.x * ACX_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * ACX_BLOCK + threadIdx.x; i < n_chunks; i += n_threads) {
        int64_t lo = 0, hi = c.n_hay - 1;          // largest h with ck_first[h] <= i
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) >> 1;
            if (c.ck_first[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const int64_t h = lo;
        const int64_t hs = c.off ? c.off[h] : h * c.stride;
        const int64_t he = c.off ? c.off[h + 1] : hs + c.stride;
        const int64_t k = i - c.ck_first[h];
        const int64_t cs = hs + k * c.chunk_bytes;
        int64_t ce = cs + c.chunk_bytes;
        if (ce > he) ce = he;
        int64_t s0 = cs - c.halo;
        if (s0 < hs) s0 = hs;
        acx_chunk_desc d;
        d.start = s0;
        d.emit = (int32_t)(cs - s0);
        d.len = (int32_t)(ce - s0);
        d.idx0 = (int32_t)(s0 - hs) + (c.index_base ? c.index_base[h] : 0);
        d.hay = (int32_t)h;
        d.flags = (s0 == hs ? 1 : 0) | (ce == he ? 2 : 0);
        d.pad = 0;
        c.ck[i] = d;
    }
}

__global__ void __launch_bounds__(ACX_BLOCK) k_hay_offsets(const int64_t* ck_first, const int64_t* ck_match_off,
                                                          int64_t n_hay, int64_t* match_off) {
    const int64_t h = (int64_t)blockIdx.x * ACX_BLOCK + threadIdx.x;
    if (h <= n_hay) match_off[h] = ck_match_off[ck_first[h]];
}

// ---------------------------------------------------------------------------------
// walk, ACX_SCAN_LONG — the state machine of automaton_search_iter_long_next
// (src/AutomatonSearchIterLong.c:89-153) on the flat table.  EDGE tells a real trie edge
// (trienode_get_next, :116) from a fail-resolved transition; EOW / FAILEOW are the two
// tests of :118-126 on the target.  A transition that is not an EDGE while a match is
// remembered ends that match (:131-132) and restarts at root one past it (:105-106);
// not an EDGE with nothing remembered is the fail walk of :134-143 followed by the edge
// step of the next loop iteration — which is exactly the fail-resolved transition.
// Every event is one final match: {end_index, state whose first output is the value}.
// ---------------------------------------------------------------------------------
// TOP: the rows of the first `n_top` states (BFS numbering: the shallowest) live in LDS.  The walk falls back to the
// root after every match, so a large share of its steps stands on a shallow state; each of those is an LDS read
// instead of an L2 request (the kernel is bound by the L2 request rate: one divergent 4-byte gather per byte).
template <int SB, bool TOP>
__global__ void __launch_bounds__(TOP ? 1024 : ACX_BLOCK) k_walk_long(const acx_walk_args a, uint32_t n_top) {
    __shared__ uint32_t s_cls4[256];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_top[];
    if (threadIdx.x < 256) s_cls4[threadIdx.x] = (uint32_t)a.cls[threadIdx.x] * 4u;
    if (TOP) for (uint32_t i = threadIdx.x; i < n_top * (a.row_bytes >> 2); i += blockDim.x) s_top[i] = a.table[i];
    __syncthreads();

    const int64_t n_threads = (int64_t)gridDim.x * blockDim.x;
    const uint8_t* table_bytes = (const uint8_t*)a.table;
    const uint8_t* limit = a.hay + a.hay_cap;

    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < a.n_hay; h += n_threads) {
        int64_t b, e;
        if (a.off) { b = a.off[h]; e = a.off[h + 1]; }
        else       { b = h * a.stride; e = b + a.stride; }
        const int len = (int)(e - b);
        const uint8_t* p = a.hay + b;
        uint2* ev = a.events + (a.ev_shift ? (b >> a.ev_shift) + h : b);
        uint2* const ev0 = ev;
        const uint32_t base = a.index_base ? (uint32_t)a.index_base[h] : 0u;

        // a carried-in state (AutomatonSearchIterLong.set(chunk, reset=False) keeps iter->state,
        // src/AutomatonSearchIterLong.c:194-211) is the trie node the unfinished walk stood on
        uint32_t state = a.init_state ? (uint32_t)a.init_state[h] : 0u;
        if (state >= a.n_states) state = 0u;
        int index = 0;
        bool have_last = false;
        int last_index = -1;
        uint32_t last_state = 0;
        // the current 16 haystack bytes live in registers: a byte load per step would cost the
        // vector-memory pipe as much as the table gather itself (one lane per clock per CU).
        // Restarts move `index` back by less than longest_word, so the block is reloaded rarely.
        uint4 blkw = make_uint4(0, 0, 0, 0);
        int blk = -1;
        for (;;) {
            bool emit = false;
            if (index < len) {
                const int nb = index >> 4;
                if (nb != blk) { blkw = load16_guarded<false>(p + (int64_t)nb * 16, limit); blk = nb; }
                const int dw = (index >> 2) & 3;
                const uint32_t wsel = dw == 0 ? blkw.x : (dw == 1 ? blkw.y : (dw == 2 ? blkw.z : blkw.w));
                const uint32_t c4 = s_cls4[(wsel >> ((index & 3) * 8)) & 0xffu];
                uint32_t en;
                if (TOP && state < n_top) en = *(const uint32_t*)((const uint8_t*)s_top + (state * a.row_bytes + c4));
                else en = load_entry<SB>(table_bytes, state, a.row_bytes, c4);
                const uint32_t next = en & ACX_ENTRY_STATE_MASK(SB);
                if (!(en & ACX_ENTRY_EDGE(SB)) && have_last) {
                    emit = true;
                } else if (next == 0) {
                    state = 0; index++;
                } else {
                    if (en & ACX_ENTRY_EOW(SB)) {
                        last_state = next; last_index = index; have_last = true;
                        state = next; index++;
                    } else if (en & ACX_ENTRY_FAILEOW(SB)) {
                        last_state = next; last_index = index; have_last = true;
                        emit = true;
                    } else {
                        state = next; index++;
                    }
                }
            } else {
                if (!have_last) break;
                emit = true;
            }
            if (emit) {
                *ev++ = make_uint2(base + (uint32_t)last_index, last_state);
                state = 0; index = last_index + 1; have_last = false;
            }
        }
        const int32_t n = (int32_t)(ev - ev0);
        a.counts[h] = n;
        a.nev[h] = n;
        if (a.final_state) a.final_state[h] = (int32_t)(state & ACX_ENTRY_STATE_MASK(SB));   // where the walk stands when the haystack ends
    }
}

// The same state machine in select form.  The branchy form above costs 138 wave-instructions a step (rocprofv3: 147 M
// VALU + 176 M scalar per 2.34 M wave-steps of config 5): every lane of a wave stands in another arm of it, each arm is
// an exec-mask region, and the SIMDs issue instructions 3/4 of the kernel's time.  Here a step is straight-line code:
//     byte    the lane's current 16 haystack bytes sit in LDS (a byte read instead of a four-way register select)
//     entry   rows of the shallowest states from LDS, read by every lane at a clamped address; the table gather only
//             for the lanes that stand deeper
//     emit  = (no edge and a match is remembered) or (the target's fail node ends a key and the target does not) or
//             (the haystack is over and a match is remembered)
//     state = emit ? root : target;  index = (emit ? last_index : index) + 1
// (a transition to the root carries neither EOW nor FAILEOW, so the root needs no test of its own).  The regions that
// stay are the reload of the 16-byte block and the store of an event.  Four steps per test of the loop condition: a
// lane that is finished idles through them.  Needs hay_cap >= 16 (a block at the end of the buffer is loaded from
// 16 bytes before its end and read at an offset).
template <int SB, bool TOP>
__global__ void __launch_bounds__(TOP ? 1024 : ACX_BLOCK) k_walk_long_sel(const acx_walk_args a, uint32_t n_top) {
    __shared__ uint32_t s_cls4[256];
    __shared__ uint4 s_blk[TOP ? 1024 : ACX_BLOCK];
    extern __shared__ __attribute__((aligned(16))) uint32_t s_top[];
    if (threadIdx.x < 256) s_cls4[threadIdx.x] = (uint32_t)a.cls[threadIdx.x] * 4u;
    if (TOP) for (uint32_t i = threadIdx.x; i < n_top * (a.row_bytes >> 2); i += blockDim.x) s_top[i] = a.table[i];
    __syncthreads();

    const int64_t n_threads = (int64_t)gridDim.x * blockDim.x;
    const uint8_t* table_bytes = (const uint8_t*)a.table;
    const uint8_t* last16 = a.hay + a.hay_cap - 16;
    const uint8_t* my_blk = (const uint8_t*)&s_blk[threadIdx.x];
    constexpr uint32_t MASK = ACX_ENTRY_STATE_MASK(SB), EDGE = ACX_ENTRY_EDGE(SB), EOW = ACX_ENTRY_EOW(SB), FEOW = ACX_ENTRY_FAILEOW(SB);
    const uint32_t top_last = TOP ? n_top - 1u : 0u;

    for (int64_t h = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; h < a.n_hay; h += n_threads) {
        int64_t b, e;
        if (a.off) { b = a.off[h]; e = a.off[h + 1]; }
        else       { b = h * a.stride; e = b + a.stride; }
        const int len = (int)(e - b);
        const uint8_t* p = a.hay + b;
        uint2* ev = a.events + (a.ev_shift ? (b >> a.ev_shift) + h : b);
        uint2* const ev0 = ev;
        const uint32_t base = a.index_base ? (uint32_t)a.index_base[h] : 0u;
        uint32_t state = a.init_state ? (uint32_t)a.init_state[h] : 0u;
        if (state >= a.n_states) state = 0u;
        int index = 0, last_index = -1;
        uint32_t last_state = 0;
        int blk = -1;
        uint32_t adj = 0;                                     // where byte 0 of the current block sits in the lane's LDS slot
        const int last = len > 0 ? len - 1 : 0;
        while (index < len || last_index >= 0) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool in = index < len;
                const int ix = index < last ? index : last;      // (a lane whose haystack is over reads a valid byte and ignores it)
                const int nb = ix >> 4;
                if (nb != blk) {
                    blk = nb;
                    const uint8_t* q = p + (int64_t)nb * 16;
                    const uint8_t* q2 = q > last16 ? last16 : q;
                    adj = (uint32_t)(q - q2);
                    s_blk[threadIdx.x] = *(const uint4*)q2;   // (unaligned 16-byte loads are fine on gfx9)
                }
                const uint32_t c4 = s_cls4[my_blk[((uint32_t)ix & 15u) + adj]];
                uint32_t en;
                if (TOP) {
                    const uint32_t st = state < top_last ? state : top_last;
                    en = *(const uint32_t*)((const uint8_t*)s_top + (st * a.row_bytes + c4));
                    if (state > top_last) en = load_entry<SB>(table_bytes, state, a.row_bytes, c4);
                } else en = load_entry<SB>(table_bytes, state, a.row_bytes, c4);
                const uint32_t next = en & MASK;
                const bool have = last_index >= 0;
                const bool emit_a = in && have && (en & EDGE) == 0u;
                const bool go = in && !emit_a;
                const bool setl = go && (en & (EOW | FEOW)) != 0u;
                const bool emit = emit_a || (go && (en & (EOW | FEOW)) == FEOW) || (!in && have);
                last_state = setl ? next : last_state;
                last_index = setl ? index : last_index;
                if (emit) *ev++ = make_uint2(base + (uint32_t)last_index, last_state);
                state = emit ? 0u : (in ? next : state);
                index = (emit ? last_index : index) + 1;
                last_index = emit ? -1 : last_index;
            }
        }
        const int32_t n = (int32_t)(ev - ev0);
        a.counts[h] = n;
        a.nev[h] = n;
        if (a.final_state) a.final_state[h] = (int32_t)(state & MASK);
    }
}

// final_state of every haystack from its last `longest` bytes (position-parallel scans keep no
// state): the state reached from the root over a window is the longest suffix of the window that
// is a trie node, and no node is longer than the longest key.
template <int SB>
__global__ void __launch_bounds__(ACX_BLOCK) k_tail_state(const acx_walk_args a, int32_t longest) {
    __shared__ uint32_t s_cls4[256];
    s_cls4[threadIdx.x] = (uint32_t)a.cls[threadIdx.x] * 4u;
    __syncthreads();
    const int64_t n_threads = (int64_t)gridDim.x * ACX_BLOCK;
    const uint8_t* table_bytes = (const uint8_t*)a.table;
    for (int64_t h = (int64_t)blockIdx.x * ACX_BLOCK + threadIdx.x; h < a.n_hay; h += n_threads) {
        int64_t b, e;
        if (a.off) { b = a.off[h]; e = a.off[h + 1]; }
        else       { b = h * a.stride; e = b + a.stride; }
        const int64_t n = e - b < longest ? e - b : longest;
        const uint8_t* p = a.hay + (e - n);
        uint32_t state = 0;
        for (int64_t i = 0; i < n; i++) state = load_entry<SB>(table_bytes, state, a.row_bytes, s_cls4[p[i]]);
        a.final_state[h] = (int32_t)(state & ACX_ENTRY_STATE_MASK(SB));
    }
}

// ---------------------------------------------------------------------------------
// scan: exclusive prefix sum int32 counts[n] -> int64 match_off[n+1]
// three small launches (per-block sums, one-block scan of the sums, per-block rescan)
// ---------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 16;                       // per thread
constexpr int SCAN_TILE = ACX_BLOCK * SCAN_ITEMS;    // 4096 per block

__device__ __forceinline__ int64_t wave_incl_scan(int64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < ACX_WAVE; d <<= 1) {
        const int64_t t = __shfl_up(v, d, ACX_WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// inclusive scan across the block of one value per thread; returns inclusive value, total via *total
__device__ __forceinline__ int64_t block_incl_scan(int64_t v, int64_t* s_wave /*[4]*/, int64_t* total) {
    const int lane = threadIdx.x & (ACX_WAVE - 1), wid = threadIdx.x / ACX_WAVE;
    const int64_t inc = wave_incl_scan(v, lane);
    if (lane == ACX_WAVE - 1) s_wave[wid] = inc;
    __syncthreads();
    int64_t basev = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ACX_BLOCK / ACX_WAVE; w++) {
        const int64_t x = s_wave[w];
        if (w < wid) basev += x;
        tot += x;
    }
    __syncthreads();
    *total = tot;
    return inc + basev;
}

__global__ void __launch_bounds__(ACX_BLOCK) k_scan_partials(const int32_t* counts, int64_t n, int64_t* partials) {
    __shared__ int64_t s_wave[ACX_BLOCK / ACX_WAVE];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int64_t i = base + (int64_t)k * ACX_BLOCK + threadIdx.x;   // coalesced
        if (i < n) sum += counts[i];
    }
    int64_t tot;
    block_incl_scan(sum, s_wave, &tot);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(ACX_BLOCK) k_scan_top(int64_t* partials, int64_t np) {
    __shared__ int64_t s_wave[ACX_BLOCK / ACX_WAVE];
    int64_t carry = 0;
    for (int64_t base = 0; base < np; base += ACX_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const int64_t v = i < np ? partials[i] : 0;
        int64_t tot;
        const int64_t inc = block_incl_scan(v, s_wave, &tot);
        if (i < np) partials[i] = carry + inc - v;      // exclusive
        carry += tot;
    }
    if (threadIdx.x == 0) partials[np] = carry;          // grand total
}

__global__ void __launch_bounds__(ACX_BLOCK) k_scan_final(const int32_t* counts, int64_t n, const int64_t* partials,
                                                          int64_t np, int64_t* match_off) {
    __shared__ int64_t s_wave[ACX_BLOCK / ACX_WAVE];
    // thread t owns SCAN_ITEMS consecutive items so its local prefix is a register loop
    const int64_t first = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int32_t c[SCAN_ITEMS];
    int64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        c[k] = (first + k < n) ? counts[first + k] : 0;
        sum += c[k];
    }
    int64_t tot;
    const int64_t inc = block_incl_scan(sum, s_wave, &tot);
    int64_t run = partials[blockIdx.x] + inc - sum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        if (first + k < n) match_off[first + k] = run;
        run += c[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) match_off[n] = partials[np];
}

// ---------------------------------------------------------------------------------
// expand: events -> final records through the CSR output lists
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(ACX_BLOCK) k_expand(const acx_expand_args a) {
    // if the staging capacity guess was too small the host grows the buffer and relaunches
    const int64_t n_items = a.n_items_dev ? *a.n_items_dev : a.n_hay;
    if (a.match_off[n_items] > a.capacity) return;
    const uint32_t state_mask = ACX_ENTRY_STATE_MASK(a.state_bits), cnt_shift = ACX_ENTRY_CNT_SHIFT(a.state_bits),
                   cnt_escape = ACX_ENTRY_CNT_ESCAPE(a.state_bits);
    const int64_t n_threads = (int64_t)gridDim.x * ACX_BLOCK;
    for (int64_t h = (int64_t)blockIdx.x * ACX_BLOCK + threadIdx.x; h < n_items; h += n_threads) {
        const int32_t n = a.nev[h];
        if (n == 0) continue;
        int64_t ev_at = a.ck ? a.ck[h].start + a.ck[h].emit : (a.off ? a.off[h] : h * a.stride);
        if (a.ev_shift) ev_at = (ev_at >> a.ev_shift) + h;
        const uint2* ev = a.events + ev_at;
        uint2* out = a.matches + a.match_off[h];
        if (a.long_mode) {
            for (int32_t k = 0; k < n; k++) {
                const uint2 v = ev[k];
                const int32_t val = a.out_val[a.out_off[v.y & state_mask]];
                *out++ = make_uint2(v.x, (uint32_t)val);
            }
        } else {
            for (int32_t k = 0; k < n; k++) {
                const uint2 v = ev[k];
                const uint32_t s = v.y & state_mask;
                uint32_t c = v.y >> cnt_shift;
                if (c == 1) { *out++ = make_uint2(v.x, (uint32_t)a.first_val[s]); continue; }   // s may be a pseudo state (itop)
                const uint32_t o = a.out_off[s];
                if (c == cnt_escape) c = a.out_off[s + 1] - o;
                for (uint32_t r = 0; r < c; r++) *out++ = make_uint2(v.x, (uint32_t)a.out_val[o + r]);
            }
        }
    }
}

// Group version (default): GROUP consecutive lanes share one haystack.  The group reads
// GROUP consecutive events with one coalesced load, turns the per-event output counts into
// record offsets with a GROUP-wide shuffle scan, and writes its records to consecutive
// slots.  Events with a single output (the overwhelmingly common case) take their value
// from first_val[s]: one gather per match instead of out_off[s] -> out_val[o].
template <int GROUP>
__global__ void __launch_bounds__(ACX_BLOCK) k_expand_grp(const acx_expand_args a) {
    const int64_t n_items = a.n_items_dev ? *a.n_items_dev : a.n_hay;
    if (a.match_off[n_items] > a.capacity) return;
    const uint32_t state_mask = ACX_ENTRY_STATE_MASK(a.state_bits), cnt_shift = ACX_ENTRY_CNT_SHIFT(a.state_bits),
                   cnt_escape = ACX_ENTRY_CNT_ESCAPE(a.state_bits);
    const int sub = threadIdx.x & (GROUP - 1);
    const int64_t groups_per_block = ACX_BLOCK / GROUP;
    const int64_t n_groups = (int64_t)gridDim.x * groups_per_block;
    for (int64_t h = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / GROUP; h < n_items; h += n_groups) {
        const int32_t n = a.nev[h];            // group-uniform
        if (n == 0) continue;
        int64_t ev_at = a.ck ? a.ck[h].start + a.ck[h].emit : (a.off ? a.off[h] : h * a.stride);
        if (a.ev_shift) ev_at = (ev_at >> a.ev_shift) + h;
        const uint2* ev = a.events + ev_at;
        int64_t out_base = a.match_off[h];
        for (int32_t k0 = 0; k0 < n; k0 += GROUP) {
            const int32_t k = k0 + sub;
            const bool valid = k < n;
            uint2 v = make_uint2(0, 0);
            // events are read once and records written once: non-temporal, so that they do not push the
            // table rows and cells of the NEXT walk out of L2 / Infinity Cache (walk 0.54 -> 0.51 ms)
            if (valid) { const u32x2 t = __builtin_nontemporal_load((const u32x2*)(ev + k)); v = make_uint2(t.x, t.y); }
            const uint32_t s = v.y & state_mask;
            uint32_t c = 0;
            if (valid) {
                if (a.long_mode) c = 1;
                else {
                    c = v.y >> cnt_shift;
                    if (c == cnt_escape) c = a.out_off[s + 1] - a.out_off[s];
                }
            }
            uint32_t incl = c;                 // inclusive scan across the group
#pragma unroll
            for (int d = 1; d < GROUP; d <<= 1) {
                const uint32_t t = __shfl_up(incl, d, GROUP);
                if (sub >= d) incl += t;
            }
            const uint32_t total = __shfl(incl, GROUP - 1, GROUP);
            uint2* out = a.matches + out_base + (incl - c);
            if (c == 1) {
                store_event<true>(out, v.x, (uint32_t)a.first_val[s]);
            } else if (c > 1) {
                const uint32_t o = a.out_off[s];
                for (uint32_t r = 0; r < c; r++) store_event<true>(out + r, v.x, (uint32_t)a.out_val[o + r]);
            }
            out_base += total;
        }
    }
}

inline int grid_for_waves(int64_t n_tasks) {
    // 256 CUs x 8 blocks of 4 waves = the chip's 32 waves/CU; grid-stride the rest
    const int64_t blocks = (n_tasks + (ACX_BLOCK / ACX_WAVE) - 1) / (ACX_BLOCK / ACX_WAVE);
    const int64_t cap = (int64_t)acx_num_cus() * 8;
    return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

}  // namespace

int64_t acx_scan_num_partials(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

// variant encoding (bench/tuning knob; 0 = default):
//   bits 0-3  ILP - 1            (0 -> one haystack per lane, 1 -> two)
//   bits 4-7  blocks per CU      (0 -> 8)
//   bit  8    count only, no events (diagnostic; the result then has no matches)
//   bits 9-11 expand kernel shape (see acx_launch_expand)
//   bit  12   non-temporal haystack loads and event stores
template <int SB, bool ESCAPE, int ILP, bool EVENTS, int NT>
static void launch_walk_all_t(const acx_walk_args& a, int blocks_per_cu, hipStream_t s) {
    const int64_t per_task = (int64_t)ACX_WAVE * ILP;
    const int64_t n_tasks = (a.n_hay + per_task - 1) / per_task;
    const int64_t blocks = (n_tasks + (ACX_BLOCK / ACX_WAVE) - 1) / (ACX_BLOCK / ACX_WAVE);
    const int64_t cap = (int64_t)acx_num_cus() * (int64_t)blocks_per_cu;
    const int grid = (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
    hipLaunchKernelGGL((k_walk_all<SB, ESCAPE, ILP, EVENTS, NT>), dim3(grid), dim3(ACX_BLOCK), 0, s, a);
}

template <bool ESCAPE>
static hipError_t dispatch_walk_all(const acx_walk_args& a, int ilp, bool events, int nt, int bpc, hipStream_t s) {
    constexpr int N = ACX_STATE_BITS_NARROW;
    if (a.state_bits == ACX_STATE_BITS_WIDE) {      // wide images: one tuned shape only
        launch_walk_all_t<ACX_STATE_BITS_WIDE, ESCAPE, 1, true, 2>(a, bpc, s);
        return hipGetLastError();
    }
    if (ilp == 2) {            // ILP 2 exists as a tuning variant only (no gain measured on MI355X)
        if (events) launch_walk_all_t<N, ESCAPE, 2, true, 0>(a, bpc, s);
        else        launch_walk_all_t<N, ESCAPE, 2, false, 0>(a, bpc, s);
    } else if (events) {
        switch (nt) {
            case 1:  launch_walk_all_t<N, ESCAPE, 1, true, 1>(a, bpc, s); break;
            case 2:  launch_walk_all_t<N, ESCAPE, 1, true, 2>(a, bpc, s); break;
            case 3:  launch_walk_all_t<N, ESCAPE, 1, true, 3>(a, bpc, s); break;
            default: launch_walk_all_t<N, ESCAPE, 1, true, 0>(a, bpc, s); break;
        }
    } else {
        if (nt & 1) launch_walk_all_t<N, ESCAPE, 1, false, 1>(a, bpc, s);
        else        launch_walk_all_t<N, ESCAPE, 1, false, 0>(a, bpc, s);
    }
    return hipGetLastError();
}

hipError_t acx_launch_walk_all(const acx_walk_args& a, bool has_escape, int variant, hipStream_t s) {
    if (a.n_hay <= 0) return hipSuccess;
    const int ilp = (variant & 0xF) + 1;
    int bpc = (variant >> 4) & 0xF;
    if (bpc == 0) bpc = 8;
    const bool events = !((variant >> 8) & 1);
    // Event stores are non-temporal by default (measured -4 % walk time on config 2: the 8-byte
    // stores, one line per lane, otherwise displace table rows from L2); haystack loads are not
    // (nt loads measured +8 %).  bit 12: nt loads too; bit 14: nt loads only; bit 15: no nt at all.
    int nt = 2;
    if ((variant >> 12) & 1) nt = 3;
    if ((variant >> 14) & 1) nt = 1;
    if ((variant >> 15) & 1) nt = 0;
    if (ilp > 2) return hipErrorInvalidValue;
    return has_escape ? dispatch_walk_all<true>(a, ilp, events, nt, bpc, s)
                      : dispatch_walk_all<false>(a, ilp, events, nt, bpc, s);
}

hipError_t acx_launch_walk_long(const acx_walk_args& a, int variant, hipStream_t s) {
    if (a.n_hay <= 0) return hipSuccess;
    // rows of the shallowest states in LDS (variant bit 21: without, A/B): 76 KiB of them, two 1024-thread blocks per
    // CU; config 5: 0.873 -> 0.774 ms.  Only for batches that fill the chip.
    static const uint32_t budget_kb = [] { const char* v = acx_tune_env("ACX_LONG_TOP_KB"); const int x = v ? atoi(v) : 0; return (x >= 4 && x <= 156) ? (uint32_t)x : 76u; }();   // tuning hook (76: two blocks per CU; 24 .. 150 KiB measured within 4 %)
    const uint32_t budget = budget_kb * 1024u;
    uint32_t n_top = a.row_bytes ? budget / a.row_bytes : 0u;
    if (n_top > a.n_states) n_top = a.n_states;
    const int64_t cus = acx_num_cus();
    if (!((variant >> 21) & 1) && n_top >= 64 && a.n_hay >= cus * 1024) {
        const size_t lds = (size_t)n_top * a.row_bytes;
        auto launch = [&](auto kernel) -> hipError_t {
            hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            const unsigned bpc = (lds + 1024) * 2 <= 160u * 1024u ? 2u : 1u;      // two blocks per CU when the rows leave the room
            hipLaunchKernelGGL(kernel, dim3((unsigned)cus * bpc), dim3(1024), lds, s, a, n_top);
            return hipGetLastError();
        };
        if (((variant >> 22) & 1) || a.hay_cap < 16) {                           // (A/B: the branchy form)
            if (a.state_bits == ACX_STATE_BITS_WIDE) return launch(k_walk_long<ACX_STATE_BITS_WIDE, true>);
            return launch(k_walk_long<ACX_STATE_BITS_NARROW, true>);
        }
        if (a.state_bits == ACX_STATE_BITS_WIDE) return launch(k_walk_long_sel<ACX_STATE_BITS_WIDE, true>);
        return launch(k_walk_long_sel<ACX_STATE_BITS_NARROW, true>);
    }
    const int grid = grid_for_waves((a.n_hay + ACX_WAVE - 1) / ACX_WAVE);
    if (!((variant >> 22) & 1) && a.hay_cap >= 16) {
        if (a.state_bits == ACX_STATE_BITS_WIDE) hipLaunchKernelGGL((k_walk_long_sel<ACX_STATE_BITS_WIDE, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, 0u);
        else                                     hipLaunchKernelGGL((k_walk_long_sel<ACX_STATE_BITS_NARROW, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, 0u);
        return hipGetLastError();
    }
    if (a.state_bits == ACX_STATE_BITS_WIDE) hipLaunchKernelGGL((k_walk_long<ACX_STATE_BITS_WIDE, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, 0u);
    else                                     hipLaunchKernelGGL((k_walk_long<ACX_STATE_BITS_NARROW, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, 0u);
    return hipGetLastError();
}

hipError_t acx_launch_tail_state(const acx_walk_args& a, int32_t longest, hipStream_t s) {
    if (a.n_hay <= 0 || !a.final_state) return hipSuccess;
    const int grid = grid_for_waves((a.n_hay + ACX_WAVE - 1) / ACX_WAVE);
    if (a.state_bits == ACX_STATE_BITS_WIDE) hipLaunchKernelGGL(k_tail_state<ACX_STATE_BITS_WIDE>, dim3(grid), dim3(ACX_BLOCK), 0, s, a, longest);
    else                                     hipLaunchKernelGGL(k_tail_state<ACX_STATE_BITS_NARROW>, dim3(grid), dim3(ACX_BLOCK), 0, s, a, longest);
    return hipGetLastError();
}

hipError_t acx_launch_scan(const int32_t* counts, int64_t n, int64_t* match_off, int64_t* partials, hipStream_t s) {
    const int64_t np = acx_scan_num_partials(n);
    if (n <= 0) {   // match_off[0] = 0
        return hipMemsetAsync(match_off, 0, sizeof(int64_t), s);
    }
    hipLaunchKernelGGL(k_scan_partials, dim3((unsigned)np), dim3(ACX_BLOCK), 0, s, counts, n, partials);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(ACX_BLOCK), 0, s, partials, np);
    hipLaunchKernelGGL(k_scan_final, dim3((unsigned)np), dim3(ACX_BLOCK), 0, s, counts, n, (const int64_t*)partials, np, match_off);
    return hipGetLastError();
}

hipError_t acx_launch_expand(const acx_expand_args& a, int variant, hipStream_t s) {
    if (a.n_hay <= 0) return hipSuccess;
    // variant bits 9-11: 0 = 16 lanes per haystack (default), 1 = one lane per haystack
    // (first version, kept for A/B), 2 = 8 lanes, 3 = 32 lanes, 4 = 4 lanes
    const int ev = (variant >> 9) & 7;
    const int group = ev == 1 ? 1 : ev == 2 ? 8 : ev == 3 ? 32 : ev == 4 ? 4 : 16;
    const int64_t per_block = ACX_BLOCK / group;
    const int64_t blocks = (a.n_hay + per_block - 1) / per_block;
    const int64_t cap = (int64_t)acx_num_cus() * 8 * 4;
    const int grid = (int)(blocks > cap ? cap : blocks);
    switch (group) {
        case 1:  hipLaunchKernelGGL(k_expand, dim3(grid), dim3(ACX_BLOCK), 0, s, a); break;
        case 4:  hipLaunchKernelGGL(k_expand_grp<4>, dim3(grid), dim3(ACX_BLOCK), 0, s, a); break;
        case 8:  hipLaunchKernelGGL(k_expand_grp<8>, dim3(grid), dim3(ACX_BLOCK), 0, s, a); break;
        case 32: hipLaunchKernelGGL(k_expand_grp<32>, dim3(grid), dim3(ACX_BLOCK), 0, s, a); break;
        default: hipLaunchKernelGGL(k_expand_grp<16>, dim3(grid), dim3(ACX_BLOCK), 0, s, a); break;
    }
    return hipGetLastError();
}

hipError_t acx_launch_chunk_count(const acx_chunk_args& c, hipStream_t s) {
    if (c.n_hay <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_chunk_count, dim3((unsigned)((c.n_hay + ACX_BLOCK - 1) / ACX_BLOCK)), dim3(ACX_BLOCK), 0, s, c);
    return hipGetLastError();
}

hipError_t acx_launch_chunk_fill(const acx_chunk_args& c, int64_t n_chunks_bound, hipStream_t s) {
    if (c.n_hay <= 0) return hipSuccess;
    int64_t blocks = (n_chunks_bound + ACX_BLOCK - 1) / ACX_BLOCK;
    if (blocks > (int64_t)acx_num_cus() * 16) blocks = (int64_t)acx_num_cus() * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_chunk_fill, dim3((unsigned)blocks), dim3(ACX_BLOCK), 0, s, c);
    return hipGetLastError();
}

hipError_t acx_launch_walk_chunks(const acx_walk_args& a, const acx_chunk_desc* ck, const int64_t* n_chunks_dev,
                                  int64_t n_chunks_bound, bool has_escape, hipStream_t s) {
    if (n_chunks_bound <= 0) return hipSuccess;
    const int grid = grid_for_waves((n_chunks_bound + ACX_WAVE - 1) / ACX_WAVE);
    constexpr int N = ACX_STATE_BITS_NARROW, W = ACX_STATE_BITS_WIDE;
    if (a.state_bits == W) {
        if (has_escape) hipLaunchKernelGGL((k_walk_chunks<W, true>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, ck, n_chunks_dev);
        else            hipLaunchKernelGGL((k_walk_chunks<W, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, ck, n_chunks_dev);
    } else {
        if (has_escape) hipLaunchKernelGGL((k_walk_chunks<N, true>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, ck, n_chunks_dev);
        else            hipLaunchKernelGGL((k_walk_chunks<N, false>), dim3(grid), dim3(ACX_BLOCK), 0, s, a, ck, n_chunks_dev);
    }
    return hipGetLastError();
}

hipError_t acx_launch_hay_offsets(const int64_t* ck_first, const int64_t* ck_match_off, int64_t n_hay,
                                  int64_t* match_off, hipStream_t s) {
    hipLaunchKernelGGL(k_hay_offsets, dim3((unsigned)((n_hay + 1 + ACX_BLOCK - 1) / ACX_BLOCK)), dim3(ACX_BLOCK), 0, s,
                       ck_first, ck_match_off, n_hay, match_off);
    return hipGetLastError();
}

// tune: bits 0-1 = items per lane - 1 (0 or 1); bit 2 unused;
// bits 4-5 = haystack bytes fetched per lane at a time (0: 64 (32 with 2 items per lane), 1: 16, 2: 32, 3: 64)
hipError_t acx_launch_walk_itop(const acx_walk_args& a, const acx_chunk_desc* ck, const int64_t* n_chunks_dev,
                                int64_t n_items_bound, bool has_escape, const uint32_t* itop_lds, uint32_t itop_words,
                                const uint32_t* itop_entry, const uint32_t* itop_ebits, const void* itop_cells,
                                const uint32_t* tflags, uint32_t cell_bytes, uint32_t itop_flags, int tune, hipStream_t s) {
    if (n_items_bound <= 0) return hipSuccess;
    const int ilp = (tune & 3) == 1 ? 2 : 1;
    const bool noesc = (itop_flags & ACX_ITOP_FLAG_NOESC) != 0;
    int hb = (tune >> 4) & 3;
    hb = hb == 0 ? (ilp == 2 ? 2 : 4) : (hb == 1 ? 1 : (hb == 2 ? 2 : 4));
    if (ilp == 2 && hb == 4) hb = 2;                              // register budget: 1024 threads -> 128 VGPRs
    int bpc_env = 0;
    int threads = ACX_ITOP_BLOCK;
    // occupancy experiments (tools/itop_sweep.sh); read once per process
    static const int env_bpc = [] { const char* v = acx_tune_env("ACX_ITOP_BPC"); const int x = v ? atoi(v) : 0; return x >= 1 && x <= 8 ? x : 0; }();
    static const int env_threads = [] { const char* v = acx_tune_env("ACX_ITOP_THREADS"); const int x = v ? atoi(v) : 0; return (x == 256 || x == 512 || x == 768) ? x : 0; }();
    bpc_env = env_bpc;
    if (env_threads) threads = env_threads;
    const size_t lds_bytes = (size_t)((itop_words + 3) & ~3u) * 4 + 1024;
    if (lds_bytes > 160 * 1024 || (cell_bytes != 4 && cell_bytes != 8)) return hipErrorInvalidValue;
    int bpc = lds_bytes * 2 <= 160 * 1024 ? 2 : 1;          // 1024-thread blocks: at most 2 per CU
    if (bpc_env) bpc = bpc_env;
    const int64_t waves_per_block = threads / ACX_WAVE;
    const int64_t n_tasks = (n_items_bound + ACX_WAVE * ilp - 1) / (ACX_WAVE * ilp);
    int64_t blocks = (n_tasks + waves_per_block - 1) / waves_per_block;
    if (blocks > (int64_t)acx_num_cus() * bpc) blocks = (int64_t)acx_num_cus() * bpc;
    if (blocks < 1) blocks = 1;
    auto launch = [&](auto kernel) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), lds_bytes, s, a, ck, n_chunks_dev,
                           itop_lds, itop_words, itop_entry, itop_ebits, (const uint32_t*)itop_cells, tflags);
        return hipGetLastError();
    };
#define ACX_ITOP_CASE(E, C8, NE) \
    do { \
        if (ilp == 2) return launch(k_walk_itop<E, C8, NE, 2, 2>); \
        if (hb == 1) return launch(k_walk_itop<E, C8, NE, 1, 1>); \
        if (hb == 2) return launch(k_walk_itop<E, C8, NE, 1, 2>); \
        return launch(k_walk_itop<E, C8, NE, 1, 4>); \
    } while (0)
#define ACX_ITOP_CASE2(E, C8) do { if (noesc) ACX_ITOP_CASE(E, C8, true); else ACX_ITOP_CASE(E, C8, false); } while (0)
    if (has_escape) { if (cell_bytes == 8) ACX_ITOP_CASE2(true, true); else ACX_ITOP_CASE2(true, false); }
    else            { if (cell_bytes == 8) ACX_ITOP_CASE2(false, true); else ACX_ITOP_CASE2(false, false); }
#undef ACX_ITOP_CASE2
#undef ACX_ITOP_CASE
    return hipErrorInvalidValue;
}
