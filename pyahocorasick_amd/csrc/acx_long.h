// acx_long.h — iter_long from the records of a position-parallel scan over the dictionary of acx_long.cpp: the launchers of
// acx_long.hip.  Internal to libacx.  (Apart from acx_kernels.h: the position-parallel kernels do not depend on it.)
#ifndef ACX_LONG_H_INCLUDED
#define ACX_LONG_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "acx.h"                   // ACX_LONG_SMALL_BITS

struct acx_long_args {
    uint2* rec;                    // the scan's records, per haystack; the reported ones are written over them from the front
    const int64_t* off;            // their offsets [n_hay + 1]
    int64_t n_hay;
    const int32_t* index_base;     // nullable: the first index of haystack h (the records' end indices count from it)
    uint32_t longest;              // longest dictionary entry
    int32_t* counts;               // out: records reported per haystack
    // The sweep may be queued behind a scan that turns out incomplete (asynchronous scans: more records than its buffer holds —
    // the host issues it again at completion): off[n_hay] > rec_capacity says so, and the kernels leave at once.
    int64_t rec_capacity;
    int32_t compact;               // 1: the sweep over compact records (round 5: a staging pass drops the U records; `variant` bit 27), 0: over the raw records in LDS
    const long long* scan_words;   // nullable: the pinned words the scan's gather wrote (k_ppm_gather_pos: total, pool ran out, short haystack) — the kernels leave when
                                   // a flag is set: the records of such a scan are whatever the pool's memory held, and the host scans again
    uint32_t* gtot;                // nullable, out (the raw sweep): records reported per group of 64 haystacks
    int64_t n_real;                // entries of the dictionary (an index beyond them — stale records — reports 0); fewer than 2^18: the values carry `below` (acx.h)
};
hipError_t acx_launch_long_sweep(const acx_long_args& a, hipStream_t s);
// prefix sum over the counts and move in one launch (acx_long.hip: k_long_place; needs gtot from the raw sweep): reports to dst, their offsets per
// haystack to new_off[n_hay + 1]
hipError_t acx_launch_long_place(const acx_long_args& a, int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s);

// The sweep straight from the scan's record POOL (fixed-stride stream scans whose haystacks are no longer than a tile): what
// k_ppm_gather_pos would move to its final place first — 8 bytes per record written and read again — is read from the grants of the
// scan's waves, swept, and only the reported records are written.  One wave per wave of the scan kernel, k_ppm_gather_pos's
// bookkeeping (total and flags to the host's pinned words, control words and the next scan's block sums back to zero) included.
struct acx_ppm_gather_args;
#define ACX_LONG_WAVE_SLACK 256u     /* slots of the output region of a scan wave beyond its own record count */
struct acx_long_fuse_args {
    int32_t* counts;               // out: records reported per haystack [n_hay]
    uint32_t* wave_base;           // out: where the packed reports of the scan's wave w start in gather_args.matches [n_waves]
    uint32_t* fail;                // out: counts the batches of 64 haystacks that held more records than a wave's LDS (the host then sweeps the gathered records instead)
    uint32_t longest;              // longest dictionary entry
    const long long* scan_words;   // nullable: the pinned words the scan's gather wrote (k_ppm_gather_pos: total, pool ran out, short haystack) — the kernels leave when
                                   // a flag is set: the records of such a scan are whatever the pool's memory held, and the host scans again
    uint32_t* gtot;                // nullable, out (the raw sweep): records reported per group of 64 haystacks
    int64_t n_real;                // entries of the dictionary
};
hipError_t acx_launch_long_gather_sweep(const acx_ppm_gather_args& c, const acx_long_fuse_args& f, hipStream_t s);
// the packed reports of every wave of the scan -> dst + new_off[first haystack that starts in the wave's run]
hipError_t acx_launch_long_move_waves(const acx_ppm_gather_args& c, const acx_long_fuse_args& f, const int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s);
hipError_t acx_launch_long_move(const acx_long_args& a, const int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s);

#endif
