// acx_long.h — iter_long from the records of a position-parallel scan over the dictionary of acx_long.cpp: the launchers of
// acx_long.hip.  Internal to libacx.  (Apart from acx_kernels.h: the position-parallel kernels do not depend on it.)
#ifndef ACX_LONG_H_INCLUDED
#define ACX_LONG_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>

struct acx_long_args {
    uint2* rec;                    // the scan's records, per haystack; the reported ones are written over them from the front
    const int64_t* off;            // their offsets [n_hay + 1]
    int64_t n_hay;
    const int32_t* index_base;     // nullable: the first index of haystack h (the records' end indices count from it)
    uint32_t longest;              // longest dictionary entry
    int32_t* counts;               // out: records reported per haystack
};
hipError_t acx_launch_long_sweep(const acx_long_args& a, hipStream_t s);
hipError_t acx_launch_long_move(const uint2* rec, const int64_t* off, const int64_t* new_off, int64_t n_hay, const int32_t* real, uint2* dst, hipStream_t s);

#endif
