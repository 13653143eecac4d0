// acx_long.hip — iter_long from the records of a position-parallel scan over the dictionary of acx_long.cpp.
//
// k_long_sweep: a lane per haystack.  Its records (end index, packed value: index | length << 24 | kind << 30) come from
// ACX_SCAN_ALL over D = E + FE + U, position ascending and longest first within a position — so the record in front of one
// with the same end is the next LONGER path of the trie that ends there.  The walk of the reference
// (automaton_search_iter_long_next, /root/reference/src/AutomatonSearchIterLong.c:89-153; oracle/ac_oracle.c orc_iter_long),
// restarted at r, stands at the node of record (end i, length l) iff that node is the longest path of the trie inside
// text[r..i]:  i - l + 1 >= r  and the next longer one starts in front of r.  Then
//     FE (a path whose fail node ends a key, itself none)   :122-126   report the fail node's key at i, restart at i + 1
//     E  (the node ends a key)                               :118-121   remember it and go on DOWN the trie from where this
//         path started: the later records with the same start — an FE among them is reported at once, else the deepest E —
//         :131-132, :148-150: report what is remembered, restart behind it
// tests/test_iter_long_plan_cpu.py holds the same sweep in Python, pinned against the oracle.  The reported records are
// written over the records from the front (never ahead of one still to be read); k_long_move puts them where the caller
// finds them.
#include "acx_kernels.h"
#include "acx_long.h"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef uint32_t rec2_t __attribute__((ext_vector_type(2)));          // (a record as a plain vector: pointers to it may carry an address space)
typedef __attribute__((address_space(3))) rec2_t lds_rec2_t;
// where record k of a haystack lives: in global memory, or in the wave's LDS.  There record i of the group sits at slot
// i + i / 16: the lanes of a wave stand about sixteen records apart — 128 bytes, the same bank for all of them — and
// the skew spreads them over the banks.
struct GlobalRecs { rec2_t* p; __device__ rec2_t get(uint32_t k) const { return p[k]; } __device__ void put(uint32_t k, rec2_t v) const { p[k] = v; } };
struct LdsRecs {
    lds_rec2_t* p; uint32_t first;                                      // the haystack's first record, counted from the group's first in LDS
    __device__ static uint32_t slot(uint32_t i) { return i + (i >> 4); }
    __device__ rec2_t get(uint32_t k) const { return p[slot(first + k)]; }
    __device__ void put(uint32_t k, rec2_t v) const { p[slot(first + k)] = v; }
};

// The sweep of ONE haystack over its records rec[0 .. n) (R: where they live, below), reported records written over them
// from the front; returns how many.  r: the haystack's first index.  ONE record is read per trip of the loop whatever the
// walk is doing — the 64 haystacks of a wave are swept in lock step, and inner loops (look ahead along a path, skip to the
// restart) would be run by all lanes for as long as the slowest needs:
//     looking for a stop     a record that ends in front of r is passed over; else the test of the header; FE: reported, r
//                            moves behind it; E: remembered, with the start p of its path -> following the path
//     following the path     records that start at p: an E is remembered instead, an FE is reported at once; a record that
//                            ends beyond p + longest - 1, or the end of the records: what is remembered is reported —
//                            then the walk goes on from the record behind the one reported (they are read again)
// The record in front of the current one travels in registers (for "does a longer path end here?"); after a restart the
// one in front ends before r, so it cannot end where the current one ends.
template <typename R>
__device__ __forceinline__ uint32_t sweep_one(R rec, uint32_t n, int32_t r, int32_t reach) {
    // Straight-line: every condition a 0 / 1 word, combined with & and | (a short-circuit && is an exec-mask region and a
    // branch each — the first version of this loop was 80 basic blocks), one LDS read and one predicated write per trip.
    uint32_t k = 0, w = 0;                                             // the record to read, the next slot to write
    int32_t prev_e = INT32_MIN; uint32_t prev_len = 0;
    uint32_t path = 0;                                                 // 1: following a path
    int32_t p = 0; uint32_t last_e = 0, last_i = 0, k_last = 0;
    if (n == 0u) return 0u;
    for (;;) {
        const uint32_t eor = k >= n ? 1u : 0u;                         // behind the last record
        if (eor & (path ^ 1u)) break;
        const rec2_t x = rec.get(eor ? n - 1u : k);
        const int32_t e = (int32_t)x.x;
        const uint32_t kind = x.y >> 30, len = (x.y >> 24) & 63u, idx = x.y & 0xFFFFFFu;
        const int32_t start = e - (int32_t)len + 1;
        const uint32_t is_fe = kind == 2u ? 1u : 0u, is_ev = kind != 0u ? 1u : 0u;
        // following a path: its end (the remembered record is reported, the records behind it are read again), a record of the path
        const uint32_t p_end = path & (eor | (e > p + reach ? 1u : 0u));
        const uint32_t p_hit = path & (p_end ^ 1u) & is_ev & (start == p ? 1u : 0u);
        const uint32_t p_fe = p_hit & is_fe, p_e = p_hit & (is_fe ^ 1u);
        // looking for a stop
        const uint32_t s_act = (path ^ 1u) & (eor ^ 1u);
        const uint32_t longer_in = (prev_e == e ? 1u : 0u) & (e - (int32_t)prev_len + 1 >= r ? 1u : 0u);    // a longer path ends here and starts at or behind r
        const uint32_t fires = s_act & is_ev & (start >= r ? 1u : 0u) & (longer_in ^ 1u);                     // (start >= r implies e >= r)
        const uint32_t s_fe = fires & is_fe, s_e = fires & (is_fe ^ 1u);
        const uint32_t emit = p_end | p_fe | s_fe;
        rec2_t out; out.x = p_end ? last_e : (uint32_t)e; out.y = p_end ? last_i : idx;
        if (emit) rec.put(w, out);                                      // (w <= the record being read or remembered: never ahead of one still to be read)
        w += emit;
        r = emit ? (int32_t)out.x + 1 : r;
        const uint32_t keep = p_e | s_e;                                // remember this record
        last_e = keep ? (uint32_t)e : last_e; last_i = keep ? idx : last_i;
        const uint32_t k_next = p_end ? k_last + 1u : k + 1u;
        k_last = keep ? k : k_last;
        p = s_e ? start : p;
        const uint32_t seen = s_act | p_fe;                             // this record is the one "in front" of the next
        prev_e = p_end ? INT32_MIN : (seen ? e : prev_e); prev_len = seen ? len : prev_len;
        path = (path & (p_end ^ 1u) & (p_fe ^ 1u)) | s_e;
        k = k_next;
    }
    return w;
}

// (what one lane wrote — LDS or global memory — is what the others of its wave read behind this)
__device__ __forceinline__ void long_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += t; }
    return v;
}

// A wave per 64 consecutive haystacks: their records are one contiguous range, loaded into LDS with coalesced reads (as
// many haystacks at a time as fit), swept there a lane per haystack, and the reported records of the 64 written back
// packed from the range's first slot on (k_long_move then moves one contiguous piece per wave).  A haystack with more
// records than the wave's LDS is swept in place in global memory by its lane alone.
constexpr uint32_t LONG_CAP = 1200;                                   // records per wave in LDS (10 KiB with the skew; four waves per block, four blocks per CU)
__global__ void __launch_bounds__(256) k_long_sweep(const acx_long_args a) {
    __shared__ uint2 s_rec[4][LONG_CAP + LONG_CAP / 16 + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint2* const sw = s_rec[wid];
    const int64_t n_groups = (a.n_hay + 63) / 64, n_waves = (int64_t)gridDim.x * 4;
    const int32_t reach = (int32_t)a.longest - 1;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wid; g < n_groups; g += n_waves) {
        const int64_t h = g * 64 + lane;
        const bool valid = h < a.n_hay;
        const int64_t lo = a.off[valid ? h : a.n_hay], hi = valid ? a.off[h + 1] : lo;
        const int64_t base = a.off[g * 64];
        const uint32_t nrec = (uint32_t)(hi - lo);
        const uint32_t lo_rel = (uint32_t)(lo - base);                  // (a group's records: fewer than 2^32)
        const int32_t r0 = (valid && a.index_base) ? a.index_base[h] : 0;
        const uint32_t incl = wave_incl_scan_u32(nrec, lane);
        uint32_t P = 0;                                                 // reported records of the group so far
        int done = 0;                                                   // lanes done
        while (done < 64) {
            const uint32_t start = done ? (uint32_t)__shfl((int)incl, done - 1, 64) : 0u;
            const bool fits = lane >= done && incl - start <= LONG_CAP;
            const int e = done + (int)__popcll(__ballot(fits));
            uint32_t c = 0;
            if (e == done) {
                // the haystack of lane `done` alone has more records than LDS takes: in place, in global memory
                if (lane == done) c = sweep_one(GlobalRecs{(rec2_t*)a.rec + lo}, nrec, r0, reach);
                const uint32_t cc = (uint32_t)__shfl((int)c, done, 64), src_rel = (uint32_t)__shfl((int)lo_rel, done, 64);
                long_wave_sync();
                if (src_rel != P) {                                     // down to the group's packed place: 64 at a time, read before written
                    for (uint32_t i0 = 0; i0 < cc; i0 += 64) {
                        uint2 v = make_uint2(0, 0);
                        if (i0 + lane < cc) v = a.rec[base + src_rel + i0 + lane];
                        long_wave_sync();
                        if (i0 + lane < cc) a.rec[base + P + i0 + lane] = v;
                        long_wave_sync();
                    }
                }
                if (lane == done && valid) a.counts[h] = (int32_t)c;
                P += cc; done++;
                continue;
            }
            const uint32_t sub_n = (uint32_t)__shfl((int)incl, e - 1, 64) - start;
            // (eight loads in flight per lane: a load waited for before the next is issued is a round trip to memory per 64 records)
            for (uint32_t i0 = 0; i0 < sub_n; i0 += 512u) {
                uint2 v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)lane; v[j] = i < sub_n ? a.rec[base + start + i] : make_uint2(0u, 0u); }
#pragma unroll
                for (int j = 0; j < 8; j++) { const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)lane; if (i < sub_n) sw[LdsRecs::slot(i)] = v[j]; }
            }
            long_wave_sync();
            const bool mine = lane >= done && lane < e;
            if (mine) c = sweep_one(LdsRecs{(lds_rec2_t*)sw, lo_rel - start}, nrec, r0, reach);
            const uint32_t ci = wave_incl_scan_u32(c, lane), tot = (uint32_t)__shfl((int)ci, 63, 64);
            if (mine) {
                uint2* dst = a.rec + base + P + (ci - c);
                for (uint32_t i = 0; i < c; i++) dst[i] = sw[LdsRecs::slot(lo_rel - start + i)];
                if (valid) a.counts[h] = (int32_t)c;
            }
            P += tot; done = e;
            long_wave_sync();
        }
    }
}

// the reported records of every group of 64 haystacks — one contiguous piece at the front of the group's range — to
// dst + new_off[first haystack of the group]; a wave per group
__global__ void __launch_bounds__(256) k_long_move(const uint2* rec, const int64_t* off, const int64_t* new_off, int64_t n_hay, const int32_t* __restrict__ real, uint2* dst) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t n_groups = (n_hay + 63) / 64, n_waves = (int64_t)gridDim.x * 4;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wid; g < n_groups; g += n_waves) {
        const int64_t h0 = g * 64, h1 = h0 + 64 < n_hay ? h0 + 64 : n_hay;
        const uint2* src = rec + off[h0];
        uint2* d = dst + new_off[h0];
        const int64_t n = new_off[h1] - new_off[h0];
        for (int64_t k = lane; k < n; k += 64) { uint2 v = src[k]; v.y = (uint32_t)real[v.y]; d[k] = v; }      // (entry index -> what iter_long reports for it)
    }
}

}  // namespace

hipError_t acx_launch_long_sweep(const acx_long_args& a, hipStream_t s) {
    int64_t blocks = ((a.n_hay + 63) / 64 + 3) / 4;
    const int64_t cap = (int64_t)acx_num_cus() * 12;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_sweep, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}

hipError_t acx_launch_long_move(const uint2* rec, const int64_t* off, const int64_t* new_off, int64_t n_hay, const int32_t* real, uint2* dst, hipStream_t s) {
    int64_t blocks = ((n_hay + 63) / 64 + 3) / 4;
    const int64_t cap = (int64_t)acx_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_move, dim3((unsigned)blocks), dim3(256), 0, s, rec, off, new_off, n_hay, real, dst);
    return hipGetLastError();
}
