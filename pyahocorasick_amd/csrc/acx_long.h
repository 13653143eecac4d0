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
    int64_t n_real;                // entries of the dictionary (an index beyond them — stale records — reports 0); fewer than 2^18: the values carry `below` (acx.h)
};
hipError_t acx_launch_long_sweep(const acx_long_args& a, hipStream_t s);
hipError_t acx_launch_long_move(const acx_long_args& a, const int64_t* new_off, const int32_t* real, uint2* dst, hipStream_t s);

#endif
