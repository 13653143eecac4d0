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
// finds them.  Round 5: the haystacks whose records fit a wave's LDS — all but pathological ones — are swept in the COMPACT form
// (sweep_compact below); the raw form (sweep_one) remains for a haystack whose records do not fit.
#include "acx_kernels.h"
#include "acx_long.h"
#include "acx_ppm_layout.h"

#define PPM_DESC_WORDS_L ACX_PPM_DESC_WORDS       // per wave of the scan kernel: total, n_grants, 16 x base, 16 x count (acx_ppm_kernels.hip)

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef uint32_t rec2_t __attribute__((ext_vector_type(2)));          // (a record as a plain vector: pointers to it may carry an address space)
// where record k of a haystack lives for the raw in-place sweep below: in global memory (the haystack whose records outgrow a wave's LDS)
struct GlobalRecs { rec2_t* p; __device__ rec2_t get(uint32_t k) const { return p[k]; } __device__ void put(uint32_t k, rec2_t v) const { p[k] = v; } };

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
__device__ __forceinline__ uint32_t sweep_one(R rec, uint32_t n, int32_t r, int32_t reach, bool small) {
    // Straight-line: every condition a 0 / 1 word, combined with & and | (a short-circuit && is an exec-mask region and a
    // branch each — the first version of this loop was 80 basic blocks), one LDS read and one predicated write per trip.
    // Round 6: kinds 2 (FE) and 3 (an E node with no E / FE node below it) are reported the moment they fire (`now`), a
    // remembered E's path is followed for `below` letters where the values carry it (small dictionaries) — what round 5 found for
    // the compact form, without its staging pass (tests/test_iter_long_plan_cpu.py: sweep_records_lockstep_now).
    uint32_t k = 0, w = 0;                                             // the record to read, the next slot to write
    int32_t prev_e = INT32_MIN; uint32_t prev_len = 0;
    uint32_t path = 0;                                                 // 1: following a path
    int32_t p = 0, limit = 0; uint32_t last_e = 0, last_i = 0, k_last = 0;
    const uint32_t imask = small ? (1u << ACX_LONG_SMALL_BITS) - 1u : 0xFFFFFFu;
    if (n == 0u) return 0u;
    for (;;) {
        const uint32_t eor = k >= n ? 1u : 0u;                         // behind the last record
        if (eor & (path ^ 1u)) break;
        const rec2_t x = rec.get(eor ? n - 1u : k);
        const int32_t e = (int32_t)x.x;
        const uint32_t kind = x.y >> 30, len = (x.y >> 24) & 63u, idx = x.y & imask;
        const int32_t start = e - (int32_t)len + 1;
        const uint32_t now = kind >> 1, is_ev = kind != 0u ? 1u : 0u;
        // following a path: its end (the remembered record is reported, the records behind it are read again), a record of the path
        const uint32_t p_end = path & (eor | (e > limit ? 1u : 0u));
        const uint32_t p_hit = path & (p_end ^ 1u) & is_ev & (start == p ? 1u : 0u);
        const uint32_t p_now = p_hit & now, p_e = p_hit & (now ^ 1u);
        // looking for a stop
        const uint32_t s_act = (path ^ 1u) & (eor ^ 1u);
        const uint32_t longer_in = (prev_e == e ? 1u : 0u) & (e - (int32_t)prev_len + 1 >= r ? 1u : 0u);    // a longer path ends here and starts at or behind r
        const uint32_t fires = s_act & is_ev & (start >= r ? 1u : 0u) & (longer_in ^ 1u);                     // (start >= r implies e >= r)
        const uint32_t s_now = fires & now, s_e = fires & (now ^ 1u);
        const uint32_t emit = p_end | p_now | s_now;
        rec2_t out; out.x = p_end ? last_e : (uint32_t)e; out.y = p_end ? last_i : idx;
        if (emit) rec.put(w, out);                                      // (w <= the record being read or remembered: never ahead of one still to be read)
        w += emit;
        r = emit ? (int32_t)out.x + 1 : r;
        const uint32_t keep = p_e | s_e;                                // remember this record
        last_e = keep ? (uint32_t)e : last_e; last_i = keep ? idx : last_i;
        const int32_t lim_new = small ? e + (int32_t)((x.y >> ACX_LONG_SMALL_BITS) & 63u) : start + reach;   // (below <= longest - length: never beyond p + reach)
        limit = keep ? lim_new : limit;
        const uint32_t k_next = p_end ? k_last + 1u : k + 1u;
        k_last = keep ? k : k_last;
        p = s_e ? start : p;
        const uint32_t seen = s_act | p_now;                            // this record is the one "in front" of the next
        prev_e = p_end ? INT32_MIN : (seen ? e : prev_e); prev_len = seen ? len : prev_len;
        path = (path & (p_end ^ 1u) & (p_now ^ 1u)) | s_e;
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

// ---- the sweep over COMPACT records (round 5) -----------------------------------------------------------------------------
// What a trip of the loop above spends most of its instructions on does not depend on where the walk restarted: is the record
// an event, where does its path start, does a longer path of the trie end with it.  The staging pass — lane per RECORD,
// coalesced — answers that once per record and keeps only the E and FE records (a U record only ever says "a longer path ends
// here" about the record behind it):
//     {start, up_start, value}     start = end - length + 1;  up_start = the start of the record in front when it ends at the same
//                                  index of the same haystack (the next LONGER path), else INT_MIN
// and the rule of acx_long.cpp becomes  fires <=> up_start < r <= start.  kind 3 (an E node with no E / FE node below it in the
// trie, acx_long.cpp) is reported the moment it fires — nothing the walk can still meet on that path replaces it — so only E
// nodes with events below them make the lane look ahead along their path and read the records behind the report again.
// k_long_sweep (round 4) spent 72 instructions per trip and ~70 trips per group of 64 haystacks (profiles/r4_iter_long_steps.txt);
// tests/test_iter_long_plan_cpu.py holds this form in Python (sweep_records_compact), pinned against the oracle.
#ifndef ACX_LONG_EXP
#define ACX_LONG_EXP 0                    // development, timing only (results are wrong): 1 no sweep, 2 no copy-out, 4 no staging beyond the loads
#endif
#ifndef ACX_LC_CAP
#define ACX_LC_CAP 1280
#endif
#ifndef ACX_LC_LOADS
#define ACX_LC_LOADS 4
#endif
constexpr uint32_t LC_CAP = ACX_LC_CAP;                                     // compact records per wave in LDS: 12 bytes each, 15 KiB; one wave per block (the waves share nothing), ten blocks per CU
constexpr uint32_t LC_CHUNKS = LC_CAP / 64;
struct LongLds {
    uint2 a[LC_CAP + 64];                                             // {start, up_start}; a lane's reports are written over its records from the front; [LC_CAP + lane]: where a lane with nothing to report writes
    uint32_t v[LC_CAP];                                               // index | length << 24 | kind << 30
    uint32_t first_bits[LC_CAP / 32];                                 // raw record i of the batch is the first of its haystack
    uint32_t cbase[LC_CHUNKS + 1];                                    // compact records in front of raw chunk c
    unsigned long long kmask[LC_CHUNKS + 1];                          // which raw records of chunk c were kept
};

// one haystack over its compact records a[first .. first + n): reports written over them from the front; returns how many
// small: the dictionary has fewer than 2^18 entries — the index is 18 bits, the six above it `below` (acx_long.cpp): a remembered node's
// path is followed for that many letters, not for longest - 1
__device__ __forceinline__ uint32_t sweep_compact(LongLds* L, uint32_t first, uint32_t n, int32_t r, int32_t reach, bool small, uint32_t dump) {
    uint32_t k = 0, w = 0, path = 0, last_e = 0, last_i = 0, k_last = 0;
    int32_t p = 0, limit = 0;
    const uint32_t imask = small ? (1u << ACX_LONG_SMALL_BITS) - 1u : 0xFFFFFFu;
    if (n == 0u) return 0u;
    // Straight-line: one LDS read and one LDS write per trip whatever the walk is doing — a lane with nothing to report writes to its
    // dump slot —, every condition a 0 / 1 word.  (The record of the next trip requested early and re-requested after a rewind, a
    // predicated write, the two forms of `limit` as branches: five exec-mask regions per trip, 370 ns per trip and wave.)
    for (;;) {
        const uint32_t eor = k >= n ? 1u : 0u;
        if (eor & (path ^ 1u)) break;
        const uint32_t kk = first + (eor ? n - 1u : k);
        const uint2 a = L->a[kk];
        const uint32_t val = L->v[kk];
        const int32_t start = (int32_t)a.x, up = (int32_t)a.y;
        const uint32_t kind = val >> 30, idx = val & imask;
        const int32_t e = start + (int32_t)((val >> 24) & 63u) - 1;
        const uint32_t now = (kind >> 1);                              // kinds 2 (FE) and 3 (E, nothing below): reported at once
        const uint32_t p_end = path & (eor | (e > limit ? 1u : 0u));
        const uint32_t p_hit = path & (p_end ^ 1u) & (start == p ? 1u : 0u);
        const uint32_t fires = (path ^ 1u) & (eor ^ 1u) & (start >= r ? 1u : 0u) & (up < r ? 1u : 0u);
        const uint32_t hit = p_hit | fires;
        const uint32_t emit = p_end | (hit & now);
        const uint32_t keep = hit & (now ^ 1u);                        // an E with events below: remembered, its path followed
        uint2 out; out.x = p_end ? last_e : (uint32_t)e; out.y = p_end ? last_i : idx;
        L->a[emit ? first + w : dump] = out;                            // (w <= the record being read or remembered)
        w += emit;
        r = emit ? (int32_t)out.x + 1 : r;
        last_e = keep ? (uint32_t)e : last_e; last_i = keep ? idx : last_i;
        const int32_t lim_new = small ? e + (int32_t)((val >> ACX_LONG_SMALL_BITS) & 63u) : start + reach;   // (below <= longest - length: never beyond p + reach)
        limit = keep ? lim_new : limit;
        const uint32_t k_next = p_end ? k_last + 1u : k + 1u;
        k_last = keep ? k : k_last;
        p = (fires & keep) ? start : p;
        path = (path & (emit ^ 1u)) | (fires & keep);
        k = k_next;
    }
    return w;
}

// A wave per 64 consecutive haystacks: their records are one contiguous range.  As many haystacks at a time as fit (by their raw
// counts) are staged — raw records read coalesced, turned into compact ones in LDS —, swept a lane per haystack, and the
// reported records of the 64 written back packed from the range's first slot on (k_long_move then moves one contiguous piece
// per wave).  A haystack with more records than the wave's LDS is swept in place in global memory by its lane alone, in the
// raw form (sweep_one above).
__global__ void __launch_bounds__(64) k_long_sweep(const acx_long_args a) {
    __shared__ LongLds s_l;
    const int lane = threadIdx.x & 63;
    LongLds* const L = &s_l;
    const int64_t n_groups = (a.n_hay + 63) / 64, n_waves = (int64_t)gridDim.x;
    const int32_t reach = (int32_t)a.longest - 1;
    const bool small = a.n_real < ((int64_t)1 << ACX_LONG_SMALL_BITS);
    if (a.off[a.n_hay] > a.rec_capacity) return;                       // (the scan in front is incomplete: see acx_long_args)
    if (a.scan_words && (a.scan_words[1] | a.scan_words[2])) return;   // (... or its record pool ran out: its records are no records)
    for (int64_t g = (int64_t)blockIdx.x; g < n_groups; g += n_waves) {
        const int64_t h = g * 64 + lane;
        const bool valid = h < a.n_hay;
        const int64_t lo = a.off[valid ? h : a.n_hay], hi = valid ? a.off[h + 1] : lo;
        const int64_t base = a.off[g * 64];
        const uint32_t nrec = (uint32_t)(hi - lo);
        const uint32_t lo_rel = (uint32_t)(lo - base);                  // (a group's records: fewer than 2^32 — the record pool is addressed with 32 bits)
        const int32_t r0 = (valid && a.index_base) ? a.index_base[h] : 0;
        const uint32_t incl = wave_incl_scan_u32(nrec, lane);
        uint32_t P = 0;                                                 // reported records of the group so far
        int done = 0;                                                   // lanes done
        while (done < 64) {
            const uint32_t start = done ? (uint32_t)__shfl((int)incl, done - 1, 64) : 0u;
#ifdef ACX_LONG_RAW                       // (development: every haystack by the raw in-place sweep)
            const bool fits = false;
#else
            const bool fits = lane >= done && incl - start <= LC_CAP;
#endif
            const int e = done + (int)__popcll(__ballot(fits));
            uint32_t c = 0;
            if (e == done) {
                // the haystack of lane `done` alone has more records than LDS takes: in place, in global memory, raw records
                if (lane == done) c = sweep_one(GlobalRecs{(rec2_t*)a.rec + lo}, nrec, r0, reach, small);
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
            const uint32_t sub_n = (uint32_t)__shfl((int)incl, e - 1, 64) - start;      // raw records of this batch of haystacks
            const bool mine = lane >= done && lane < e;
            // which raw records are the first of their haystack
            if (lane < (int)(LC_CAP / 32)) L->first_bits[lane] = 0u;
            long_wave_sync();
            if (mine && nrec) atomicOr(&L->first_bits[(lo_rel - start) >> 5], 1u << ((lo_rel - start) & 31u));
            long_wave_sync();
            // raw -> compact, 64 records per trip; four loads in flight per lane
            uint32_t cnt = 0;                                           // compact records so far
            uint32_t carry_e = 0, carry_v = 0;                          // the raw record in front of this trip's first
            for (uint32_t i0 = 0; i0 < sub_n; i0 += 64u * ACX_LC_LOADS) {
                uint2 rv[ACX_LC_LOADS];
#pragma unroll
                for (int j = 0; j < ACX_LC_LOADS; j++) { const uint32_t i = i0 + 64u * (uint32_t)j + (uint32_t)lane; rv[j] = i < sub_n ? a.rec[base + start + i] : make_uint2(0u, 0u); }
#pragma unroll
                for (int j = 0; j < ACX_LC_LOADS; j++) {
                    const uint32_t c0 = i0 + 64u * (uint32_t)j;
                    if (c0 >= sub_n) break;                             // (wave-uniform)
                    if (ACX_LONG_EXP & 4) { if (rv[j].x == 0x12345678u) L->v[lane] = rv[j].y; cnt += 61u; if (lane == 0) { L->cbase[c0 >> 6] = cnt - 61u; L->kmask[c0 >> 6] = 0x1FFFFFFFFFFFFFFFull; } continue; }
                    const uint32_t i = c0 + (uint32_t)lane;
                    // (the shuffles by ALL lanes, then the choice: inside a `lane ? shuffle : carry` lane 0 sits the shuffle out, and lane 1 reads an inactive lane)
                    const uint32_t se = (uint32_t)__shfl_up((int)rv[j].x, 1, 64), sv = (uint32_t)__shfl_up((int)rv[j].y, 1, 64);
                    const uint32_t pe = lane ? se : carry_e;
                    const uint32_t pv = lane ? sv : carry_v;
                    carry_e = (uint32_t)__shfl((int)rv[j].x, 63, 64); carry_v = (uint32_t)__shfl((int)rv[j].y, 63, 64);
                    const bool in = i < sub_n;
                    const uint32_t fb = in ? (L->first_bits[i >> 5] >> (i & 31u)) & 1u : 0u;
                    const uint32_t val = rv[j].y;
                    const int32_t st = (int32_t)rv[j].x - (int32_t)((val >> 24) & 63u) + 1;
#ifdef ACX_LONG_NOFB
                    const bool same = in && i > 0u && pe == rv[j].x; (void)fb;
#else
                    const bool same = in && !fb && i > 0u && pe == rv[j].x;
#endif
                    const int32_t up = same ? (int32_t)pe - (int32_t)((pv >> 24) & 63u) + 1 : INT32_MIN;
                    const bool keep = in && (val >> 30) != 0u;
                    const unsigned long long km = __ballot(keep);
                    const uint32_t pos = cnt + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                    if (keep) { L->a[pos] = make_uint2((uint32_t)st, (uint32_t)up); L->v[pos] = val; }
                    if (lane == 0) { L->cbase[c0 >> 6] = cnt; L->kmask[c0 >> 6] = km; }
                    cnt += (uint32_t)__popcll(km);
                }
            }
            if (lane == 0) { L->cbase[(sub_n + 63u) >> 6] = cnt; L->kmask[(sub_n + 63u) >> 6] = 0ull; }
            long_wave_sync();
            // where the compact records of this lane's haystack begin: the kept records in front of its first raw record
            auto compact_at = [&](uint32_t raw) -> uint32_t { return L->cbase[raw >> 6] + (uint32_t)__popcll(L->kmask[raw >> 6] & ((1ull << (raw & 63u)) - 1ull)); };
            const uint32_t cf = mine ? compact_at(lo_rel - start) : 0u;
            const uint32_t cl = mine ? compact_at(lo_rel - start + nrec) : 0u;
            if (mine && !(ACX_LONG_EXP & 1)) c = sweep_compact(L, cf, cl - cf, r0, reach, small, LC_CAP + (uint32_t)lane);
            if (ACX_LONG_EXP & 1) c = (cl - cf) / 2;
            const uint32_t ci = wave_incl_scan_u32(c, lane), tot = (uint32_t)__shfl((int)ci, 63, 64);
            long_wave_sync();                                           // (every raw record of this batch has been read: the reports go over them)
            if (mine) {
                uint2* dst = a.rec + base + P + (ci - c);
                if (!(ACX_LONG_EXP & 2)) for (uint32_t i = 0; i < c; i++) dst[i] = L->a[cf + i];
                if (valid) a.counts[h] = (int32_t)c;
            }
            P += tot; done = e;
            long_wave_sync();
        }
    }
}

// ---- the sweep over RAW records in LDS (round 4's kernel, round 6: with `now` and `below` in sweep_one) ------------------------------------
// Staging is a copy — eight loads in flight per lane, record i of the batch at slot i + i / 16: the lanes of a wave stand about sixteen
// records apart (128 bytes: one bank for all of them), the skew spreads them over the banks —, four waves per block, 10 KiB each.
// Against the compact form: a trip per U record more (a few per haystack), no ballots, prefix counts and bit tables in the staging pass.
typedef __attribute__((address_space(3))) rec2_t lds_rec2_t;
struct LdsRecs {
    lds_rec2_t* p; uint32_t first;                                      // the haystack's first record, counted from the batch's first in LDS
    __device__ static uint32_t slot(uint32_t i) { return i + (i >> 4); }
    __device__ rec2_t get(uint32_t k) const { return p[slot(first + k)]; }
    __device__ void put(uint32_t k, rec2_t v) const { p[slot(first + k)] = v; }
};
#ifndef ACX_LONG_CAP
#define ACX_LONG_CAP 1200
#endif
constexpr uint32_t LONG_CAP = ACX_LONG_CAP;                                   // records per wave in LDS (10 KiB with the skew; four waves per block, four blocks per CU)
__global__ void __launch_bounds__(256) k_long_sweep_raw(const acx_long_args a) {
    __shared__ uint2 s_rec[4][LONG_CAP + LONG_CAP / 16 + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint2* const sw = s_rec[wid];
    const int64_t n_groups = (a.n_hay + 63) / 64, n_waves = (int64_t)gridDim.x * 4;
    const int32_t reach = (int32_t)a.longest - 1;
    const bool small = a.n_real < ((int64_t)1 << ACX_LONG_SMALL_BITS);
    if (a.off[a.n_hay] > a.rec_capacity) return;                       // (the scan in front is incomplete: see acx_long_args)
    if (a.scan_words && (a.scan_words[1] | a.scan_words[2])) return;   // (... or its record pool ran out: its records are no records)
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
                if (lane == done) c = sweep_one(GlobalRecs{(rec2_t*)a.rec + lo}, nrec, r0, reach, small);
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
            if (mine) c = sweep_one(LdsRecs{(lds_rec2_t*)sw, lo_rel - start}, nrec, r0, reach, small);
            const uint32_t ci = wave_incl_scan_u32(c, lane), tot = (uint32_t)__shfl((int)ci, 63, 64);
            if (mine) {
                uint2* dst = a.rec + base + P + (ci - c);
                for (uint32_t i = 0; i < c; i++) dst[i] = sw[LdsRecs::slot(lo_rel - start + i)];
                if (valid) a.counts[h] = (int32_t)c;
            }
            P += tot; done = e;
            long_wave_sync();
        }
        if (a.gtot && lane == 0) a.gtot[g] = P;
    }
}

// ---- prefix sum and move in ONE launch ---------------------------------------------------------------------------------------------
// k_long_sweep_raw leaves the reports of a group of 64 haystacks packed at the front of the group's records, a count per haystack and
// the group's total (gtot).  What followed was a prefix sum over the counts (three launches) and k_long_move: five kernels behind the
// gather, each of which has to find CUs beside the next batch's scan kernel.  k_long_place does both in one: a block takes a CHUNK of
// 16 groups and adds up the totals of all the groups in front of it ITSELF — 15 625 words for a million haystacks, in the L2 since the
// sweep wrote them; every block reads its own prefix of them, 30 KB on average —, then every wave places four groups: offsets per
// haystack from the counts, the records moved.  No look-back and no tickets: two versions with a look-back over published totals were
// built first and were slower (sweeping in the same launch: a wave waited for the slowest sweep of the 64 groups in front, 0.31 ms
// against 0.18 for sweep + prefix sum + move; publishing totals only: 8 192 waves start at once and the last walk back over 8 000
// unfinished words, a round trip to memory per 64 — 0.30 ms for the placement alone).
constexpr int LONG_PLACE_GROUPS = 16;                                  // groups per block
__global__ void __launch_bounds__(256) k_long_place(const acx_long_args a, int64_t* new_off, const int32_t* __restrict__ real, uint2* dst) {
    __shared__ uint32_t s_part[4];
    __shared__ uint32_t s_excl[LONG_PLACE_GROUPS + 1];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t n_hay = a.n_hay, n_groups = (n_hay + 63) / 64;
    if (a.off[n_hay] > a.rec_capacity) return;                         // (the scan in front is incomplete: see acx_long_args)
    if (a.scan_words && (a.scan_words[1] | a.scan_words[2])) return;   // (... or its record pool ran out: its records are no records)
    const uint32_t n_real = (uint32_t)a.n_real;
    const int64_t g0 = (int64_t)blockIdx.x * LONG_PLACE_GROUPS;
    // the reports in front of the chunk
    uint32_t sum = 0;
    {
        int64_t i = threadIdx.x;
        for (; i + 768 < g0; i += 1024) {                               // (four loads in flight per thread)
            const uint32_t v0 = a.gtot[i], v1 = a.gtot[i + 256], v2 = a.gtot[i + 512], v3 = a.gtot[i + 768];
            sum += (v0 + v1) + (v2 + v3);
        }
        for (; i < g0; i += 256) sum += a.gtot[i];
    }
    sum = wave_incl_scan_u32(sum, lane);
    if (lane == 63) s_part[wid] = sum;
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        const uint32_t t = (lane < LONG_PLACE_GROUPS && g0 + lane < n_groups) ? a.gtot[g0 + lane] : 0u;
        const uint32_t inc = wave_incl_scan_u32(t, lane);
        if (lane < LONG_PLACE_GROUPS) s_excl[lane] = before + inc - t;
        if (lane == LONG_PLACE_GROUPS - 1) s_excl[LONG_PLACE_GROUPS] = before + inc;
    }
    __syncthreads();
    for (int k = wid; k < LONG_PLACE_GROUPS; k += 4) {
        const int64_t g = g0 + k;
        if (g >= n_groups) break;
        const int64_t h = g * 64 + lane;
        const bool valid = h < n_hay;
        const uint32_t excl = s_excl[k], total = s_excl[k + 1] - excl;
        const uint32_t c = valid ? (uint32_t)a.counts[h] : 0u;
        const uint2* src = a.rec + a.off[g * 64];
        const uint32_t ci = wave_incl_scan_u32(c, lane);
        if (valid) new_off[h] = (int64_t)excl + (ci - c);
        if (g == n_groups - 1 && lane == 0) new_off[n_hay] = (int64_t)excl + total;
        uint2* const d = dst + excl;
        for (uint32_t i = (uint32_t)lane; i < total; i += 64u) { uint2 v = src[i]; v.y = v.y < n_real ? (uint32_t)real[v.y] : 0u; d[i] = v; }   // (entry index -> what iter_long reports for it)
    }
}

// the reported records of every group of 64 haystacks — one contiguous piece at the front of the group's range — to
// dst + new_off[first haystack of the group]; a wave per group
__global__ void __launch_bounds__(256) k_long_move(const acx_long_args a, const int64_t* new_off, const int32_t* __restrict__ real, uint2* dst) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t n_hay = a.n_hay, n_groups = (n_hay + 63) / 64, n_waves = (int64_t)gridDim.x * 4;
    if (a.off[n_hay] > a.rec_capacity) return;                         // (the scan in front is incomplete: see acx_long_args)
    if (a.scan_words && (a.scan_words[1] | a.scan_words[2])) return;   // (... or its record pool ran out: its records are no records)
    const uint32_t n_real = (uint32_t)a.n_real;
    for (int64_t g = (int64_t)blockIdx.x * 4 + wid; g < n_groups; g += n_waves) {
        const int64_t h0 = g * 64, h1 = h0 + 64 < n_hay ? h0 + 64 : n_hay;
        const uint2* src = a.rec + a.off[h0];
        uint2* d = dst + new_off[h0];
        const int64_t n = new_off[h1] - new_off[h0];
        for (int64_t k = lane; k < n; k += 64) { uint2 v = src[k]; v.y = v.y < n_real ? (uint32_t)real[v.y] : 0u; d[k] = v; }      // (entry index -> what iter_long reports for it)
    }
}


// ---- the sweep straight from the record pool (acx_long.h: acx_long_fuse_args) ------------------------------------------------------
// A wave per wave w of the scan kernel.  Its run of tiles covers the positions [A, B); it sweeps the haystacks that START in there,
// 64 at a time: their records are the part of w's record stream (its grants, in order) with positions in [h_lo * stride,
// h_hi * stride) — and, for the last haystack, the head of wave w + 1's stream (a haystack is no longer than a tile: it ends in the
// next wave's run at the latest).  Staging turns (global position, value) into the compact records of sweep_compact — the haystack of
// a position by a 32-bit multiply-high as k_ppm_gather_pos finds it —; the reports go, packed, to matches[base ..), base = the
// records of the waves in front (what k_ppm_gather_pos calls base: at least as many slots as reports).
struct StreamCursor {                                                // where a wave stands in a record stream: a scan wave's grants in order
    const uint32_t* d; uint32_t g, i;                                 // descriptor, grant, record within it
};
__global__ void __launch_bounds__(64) k_long_gather_sweep(const acx_ppm_gather_args c, const acx_long_fuse_args f) {
    __shared__ LongLds s_l;
    __shared__ uint32_t s_first[64];
    LongLds* const L = &s_l;
    constexpr int GT = 64;
    const int lane = threadIdx.x;
    const int n_blocks = (int)(c.n_waves / ACX_PPM_WAVES);
    auto wave_add = [&](uint32_t x) -> uint32_t {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        return x;
    };
    uint32_t total32 = 0;
#pragma unroll 1
    for (int b = lane; b < n_blocks; b += GT) total32 += c.block_sum[b];
    const int64_t total = (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)wave_add(total32));
    if (blockIdx.x == 0) {                                             // (k_ppm_gather_pos's bookkeeping)
        uint32_t f1 = 0, f2 = 0;
        if (lane == 0) { f1 = (uint32_t)((const int32_t*)(c.ctl + 8))[0]; f2 = (uint32_t)((const int32_t*)(c.ctl + 9))[0]; c.host_words[0] = total; c.host_words[1] = (int32_t)f1; c.host_words[2] = (int32_t)f2; }
        const uint32_t seen = (uint32_t)__shfl((int)(f1 | f2 | 0x100u), 0, 64);
        if (seen && lane < 16) c.ctl[lane] = 0ull;
        for (int b = lane; b < ACX_PPM_MAX_BLOCKS; b += GT) c.block_sum_next[b] = 0u;
    }
    const bool fits = total <= c.capacity;
    const uint32_t stride = (uint32_t)c.stride;
    const uint64_t H = (uint64_t)c.n_hay * stride;
    const uint32_t m32 = (uint32_t)((((uint64_t)1 << 32) + stride - 1) / stride);
    const int32_t reach = (int32_t)f.longest - 1;
    const bool small = f.n_real < ((int64_t)1 << ACX_LONG_SMALL_BITS);
#pragma unroll 1
    for (int64_t w = blockIdx.x; w < c.n_waves; w += gridDim.x) {
        const uint32_t* const d_own = c.wave_desc + (size_t)w * PPM_DESC_WORDS_L;
        const uint64_t blk_first = (uint64_t)(w / ACX_PPM_WAVES) * ACX_PPM_WAVES * (uint64_t)c.tpw;
        const uint32_t slot = (uint32_t)(w % ACX_PPM_WAVES);
        uint64_t A = (blk_first + acx_ppm_slot_first_tile(slot, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        uint64_t B = (blk_first + acx_ppm_slot_first_tile(slot + 1u, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        if (A > H) A = H;
        if (B > H) B = H;
        const int64_t hA = (int64_t)((A + stride - 1) / stride), hB = (int64_t)((B + stride - 1) / stride);   // haystacks that start in [A, B)
        const int64_t h0 = (int64_t)(A / stride);
        const uint32_t bias = (uint32_t)(A - (uint64_t)h0 * stride) - (uint32_t)A;         // position g of this wave or the next: offset g + bias from the start of haystack h0
        uint32_t part = 0;
        const int wb = (int)(w / ACX_PPM_WAVES);
#pragma unroll 1
        for (int b = lane; b < wb; b += GT) part += c.block_sum[b];
        if (lane < (int)(w % ACX_PPM_WAVES)) part += c.wave_desc[((size_t)wb * ACX_PPM_WAVES + lane) * PPM_DESC_WORDS_L];
        const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_add(part));
        // (its reports may outnumber its own records by those of the last haystack's tail in wave w + 1's stream: ACX_LONG_WAVE_SLACK slots of slack per wave)
        const uint32_t obase = base + (uint32_t)w * ACX_LONG_WAVE_SLACK, oroom = d_own[0] + ACX_LONG_WAVE_SLACK;
        if (lane == 0) f.wave_base[w] = obase;
        if (!fits) {                                                    // (the host grows the buffer and issues the scan's second half again)
            for (int64_t hh = hA + lane; hh < hB; hh += GT) f.counts[hh] = 0;
            continue;
        }
        StreamCursor cur{d_own, 0u, 0u};
        bool in_next = false;                                           // reading the head of wave w + 1's stream
        uint32_t P = 0;                                                 // reports of this wave so far
#pragma unroll 1
        for (int64_t hs = hA; hs < hB; hs += 64) {
            const int64_t he = hs + 64 < hB ? hs + 64 : hB;
            const uint32_t p_lo = (uint32_t)((uint64_t)hs * stride), p_hi = (uint32_t)((uint64_t)he * stride - 1u) ;   // positions of these haystacks: [p_lo, p_hi]  (below 2^32: the launcher checks)
            const uint32_t q_lo = (uint32_t)(hs - h0);                  // haystack numbers relative to h0
            s_first[lane] = 0xFFFFFFFFu;
            long_wave_sync();
            uint32_t cnt = 0;                                           // compact records of this batch
            uint32_t carry_pos = 0xFFFFFFFFu, carry_v = 0, last_kept_q = 0xFFFFFFFFu;
            bool overflow = false;
            bool done = false;
            while (!done && !overflow) {                                // up to 64 x ACX_LC_LOADS records of the stream per trip, all their loads in flight at once
                uint32_t ng = cur.d[1];
                while (cur.g < ng && cur.i >= cur.d[18 + cur.g]) { cur.g++; cur.i = 0; }       // (wave-uniform: d[] are scalar loads)
                if (cur.g >= ng) {
                    if (in_next || w + 1 >= c.n_waves) break;
                    in_next = true; cur.d = c.wave_desc + (size_t)(w + 1) * PPM_DESC_WORDS_L; cur.g = 0; cur.i = 0;
                    continue;
                }
                const uint32_t n_g = cur.d[18 + cur.g];
                const uint2* src = c.scratch + cur.d[2 + cur.g];
                const uint32_t i_first = cur.i;
                uint2 rvv[ACX_LC_LOADS];
#pragma unroll
                for (int j = 0; j < ACX_LC_LOADS; j++) { const uint32_t ii = i_first + 64u * (uint32_t)j + (uint32_t)lane; rvv[j] = ii < n_g ? src[ii] : make_uint2(0xFFFFFFFFu, 0u); }
#pragma unroll
                for (int j = 0; j < ACX_LC_LOADS; j++) {
                    const uint32_t i0j = i_first + 64u * (uint32_t)j;
                    if (i0j >= n_g) break;                              // (wave-uniform)
                    const uint32_t left = n_g - i0j, nv = left < 64u ? left : 64u;
                    const uint2 rv = rvv[j];
                    const bool valid = (uint32_t)lane < nv;
                    const bool inr = valid && rv.x <= p_hi;
                    const uint32_t n_in = (uint32_t)__popcll(__ballot(inr));          // (positions ascend: a prefix of the lanes)
                    const uint32_t y = rv.x + bias, q = __umulhi(y, m32);
                    const uint32_t val = rv.y;
                    const int32_t e = (int32_t)(y - q * stride) + ((inr && c.index_base) ? c.index_base[h0 + (int64_t)(int32_t)q] : 0);
                    const uint32_t spos = (uint32_t)__shfl_up((int)rv.x, 1, 64), sval = (uint32_t)__shfl_up((int)val, 1, 64);
                    const uint32_t ppos = lane ? spos : carry_pos, pval = lane ? sval : carry_v;
                    const bool same = inr && ppos == rv.x;               // the record in front ends at the same position: the next LONGER path
                    const int32_t st = e - (int32_t)((val >> 24) & 63u) + 1;
                    const int32_t up = same ? e - (int32_t)((pval >> 24) & 63u) + 1 : INT32_MIN;
                    const bool keep = inr && rv.x >= p_lo && (val >> 30) != 0u;
                    const unsigned long long km = __ballot(keep);
                    const uint32_t pos = cnt + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
                    const uint32_t nk = (uint32_t)__popcll(km);
                    if (cnt + nk > LC_CAP) { overflow = true; break; }
                    if (keep) { L->a[pos] = make_uint2((uint32_t)st, (uint32_t)up); L->v[pos] = val; }
                    // the first compact record of every haystack: the kept record in front belongs to another one
                    {
                        const unsigned long long below_me = km & ((1ull << lane) - 1ull);
                        const int prev_lane = below_me ? 63 - (int)__clzll(below_me) : -1;
                        const uint32_t sq = (uint32_t)__shfl((int)q, prev_lane < 0 ? 0 : prev_lane, 64);
                        const uint32_t pq = prev_lane < 0 ? last_kept_q : sq;
                        if (keep && pq != q) s_first[q - q_lo] = pos;
                        if (nk) last_kept_q = (uint32_t)__shfl((int)q, 63 - (int)__clzll(km), 64);
                    }
                    cnt += nk;
                    carry_pos = (uint32_t)__shfl((int)rv.x, (int)nv - 1, 64); carry_v = (uint32_t)__shfl((int)val, (int)nv - 1, 64);
                    cur.i += n_in;
                    if (n_in < nv) { done = true; break; }               // the rest of the stream belongs to later haystacks
                }
            }
            if (overflow) { if (lane == 0) atomicAdd(f.fail, 1u); for (int64_t hh = hs + lane; hh < hB; hh += GT) f.counts[hh] = 0; break; }
            long_wave_sync();
            // compact records of haystack hs + lane: [first, next haystack's first)
            const bool mine = hs + lane < he;
            const uint32_t my_first = s_first[lane];
            uint32_t nxt = my_first;                                     // the first of the next haystack that has records (suffix minimum over the lanes behind), else cnt
            {
                uint32_t v = (uint32_t)__shfl_down((int)my_first, 1, 64);
                if (lane == 63) v = 0xFFFFFFFFu;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) { const uint32_t t = (uint32_t)__shfl_down((int)v, dd, 64); if (lane + dd < 64) v = t < v ? t : v; }
                nxt = v == 0xFFFFFFFFu ? cnt : v;
            }
            const bool has = mine && my_first != 0xFFFFFFFFu;
            const int32_t r0 = (mine && c.index_base) ? c.index_base[hs + lane] : 0;
            uint32_t cc = 0;
            if (has) cc = sweep_compact(L, my_first, nxt - my_first, r0, reach, small, LC_CAP + (uint32_t)lane);
            const uint32_t ci = wave_incl_scan_u32(cc, lane), tot = (uint32_t)__shfl((int)ci, 63, 64);
            long_wave_sync();
            if (P + tot > oroom) { if (lane == 0) atomicAdd(f.fail, 1u); for (int64_t hh = hs + lane; hh < hB; hh += GT) f.counts[hh] = 0; break; }
            if (mine) {
                uint2* dst = c.matches + obase + P + (ci - cc);
                for (uint32_t i = 0; i < cc; i++) dst[i] = L->a[my_first + i];
                f.counts[hs + lane] = (int32_t)cc;
            }
            P += tot;
            long_wave_sync();
        }
    }
}

// the packed reports of the scan's wave w -> dst + new_off[hA]; a wave per wave of the scan
__global__ void __launch_bounds__(256) k_long_move_waves(const acx_ppm_gather_args c, const acx_long_fuse_args f, const int64_t* new_off, const int32_t* __restrict__ real, uint2* dst) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t stride = (uint32_t)c.stride;
    const uint64_t H = (uint64_t)c.n_hay * stride;
    const uint32_t n_real = (uint32_t)f.n_real;
    for (int64_t w = (int64_t)blockIdx.x * 4 + wid; w < c.n_waves; w += (int64_t)gridDim.x * 4) {
        const uint64_t blk_first = (uint64_t)(w / ACX_PPM_WAVES) * ACX_PPM_WAVES * (uint64_t)c.tpw;
        const uint32_t slot = (uint32_t)(w % ACX_PPM_WAVES);
        uint64_t A = (blk_first + acx_ppm_slot_first_tile(slot, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        uint64_t B = (blk_first + acx_ppm_slot_first_tile(slot + 1u, (uint32_t)c.tpw, c.share_a, c.share_b)) * (uint64_t)c.tile_pos;
        if (A > H) A = H;
        if (B > H) B = H;
        const int64_t hA = (int64_t)((A + stride - 1) / stride), hB = (int64_t)((B + stride - 1) / stride);
        if (hA >= hB) continue;
        const uint2* src = c.matches + f.wave_base[w];
        uint2* d = dst + new_off[hA];
        const int64_t n = new_off[hB] - new_off[hA];
        for (int64_t k = lane; k < n; k += 64) { uint2 v = src[k]; v.y = v.y < n_real ? (uint32_t)real[v.y] : 0u; d[k] = v; }
    }
}

}  // namespace

hipError_t acx_launch_long_sweep(const acx_long_args& a, hipStream_t s) {
    if (!a.compact) {
        int64_t blocks = ((a.n_hay + 63) / 64 + 3) / 4;
        const int64_t cap = (int64_t)acx_num_cus() * 12;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_long_sweep_raw, dim3((unsigned)blocks), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    int64_t blocks = (a.n_hay + 63) / 64;
    const int64_t cap = (int64_t)acx_num_cus() * 40;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_sweep, dim3((unsigned)blocks), dim3(64), 0, s, a);
    return hipGetLastError();
}

hipError_t acx_launch_long_place(const acx_long_args& a, int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s) {
    const int64_t n_groups = (a.n_hay + 63) / 64;
    int64_t blocks = (n_groups + LONG_PLACE_GROUPS - 1) / LONG_PLACE_GROUPS;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_place, dim3((unsigned)blocks), dim3(256), 0, s, a, new_off, real, dst);
    return hipGetLastError();
}

hipError_t acx_launch_long_move(const acx_long_args& a, const int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s) {
    int64_t blocks = ((a.n_hay + 63) / 64 + 3) / 4;
    const int64_t cap = (int64_t)acx_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_move, dim3((unsigned)blocks), dim3(256), 0, s, a, new_off, real, dst);
    return hipGetLastError();
}

hipError_t acx_launch_long_gather_sweep(const acx_ppm_gather_args& c, const acx_long_fuse_args& f, hipStream_t s) {
    int64_t blocks = c.n_waves;
    const int64_t cap = (int64_t)acx_num_cus() * 40;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_gather_sweep, dim3((unsigned)blocks), dim3(64), 0, s, c, f);
    return hipGetLastError();
}

hipError_t acx_launch_long_move_waves(const acx_ppm_gather_args& c, const acx_long_fuse_args& f, const int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s) {
    int64_t blocks = (c.n_waves + 3) / 4;
    const int64_t cap = (int64_t)acx_num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_long_move_waves, dim3((unsigned)blocks), dim3(256), 0, s, c, f, new_off, real, dst);
    return hipGetLastError();
}

