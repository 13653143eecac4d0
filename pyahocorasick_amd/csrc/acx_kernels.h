// acx_kernels.h — launch interface between the C-ABI (acx_capi.hip) and the HIP kernels
// (acx_kernels.hip).  Internal to libacx.
#ifndef ACX_KERNELS_H_INCLUDED
#define ACX_KERNELS_H_INCLUDED

#include <hip/hip_runtime.h>
#include <stdint.h>

struct acx_walk_args {
    // input batch
    const uint8_t* hay;        // concatenated haystacks (device)
    int64_t        hay_cap;    // readable bytes at `hay`
    const int64_t* off;        // int64[n_hay+1] or nullptr (fixed stride)
    int64_t        stride;
    int64_t        n_hay;
    const int32_t* init_state; // nullable
    const int32_t* index_base; // nullable
    // image
    const uint8_t*  cls;       // uint8[256]
    const uint32_t* table;     // uint32[n_states*K]
    const uint32_t* out_off;   // uint32[n_states+1]
    uint32_t        row_bytes; // K*4
    // outputs
    int32_t* counts;           // matches per haystack
    int32_t* nev;              // events per haystack
    uint2*   events;           // staging: one slot per haystack byte; haystack h starts at slot off[h]
    int32_t* final_state;      // nullable
};

struct acx_expand_args {
    const int64_t* off;  int64_t stride;  int64_t n_hay;
    const int32_t* nev;
    const uint2*   events;
    const int64_t* match_off;  // int64[n_hay+1]
    const uint32_t* out_off;
    const int32_t*  out_val;
    uint2*   matches;          // acx_match_t[capacity]
    int64_t  capacity;
    int32_t  long_mode;        // 1: every event is exactly one match (iter_long)
};

// variant: 0 = default.  See acx_kernels.hip for the list.
hipError_t acx_launch_walk_all(const acx_walk_args& a, bool has_escape, int variant, hipStream_t s);
hipError_t acx_launch_walk_long(const acx_walk_args& a, int variant, hipStream_t s);
// exclusive prefix sum: counts int32[n] -> match_off int64[n+1]; partials = int64[ceil(n/4096)+1] scratch
hipError_t acx_launch_scan(const int32_t* counts, int64_t n, int64_t* match_off, int64_t* partials, hipStream_t s);
hipError_t acx_launch_expand(const acx_expand_args& a, int variant, hipStream_t s);
int64_t acx_scan_num_partials(int64_t n);

#endif
