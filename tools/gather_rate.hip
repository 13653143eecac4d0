// gather_rate.hip — what limits divergent gathers on gfx950: the CU (L1/TA) or the L2s?  Plain, non-temporal and
// system-coherent (L1-bypassing) 8-byte gathers from a 2 MiB table, with 256 / 128 / 64 / 32 CUs active.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int MODE, int NI>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t* tab, uint32_t* out, uint32_t words, int steps) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u, acc = 0;
    const uint32_t m = (words - 1) & ~1u;
    for (int it = 0; it < steps; it++) {
        uint32_t a[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) { x = x * 1664525u + 1013904223u; a[k] = (x >> 4) & m; }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            uint2 v;
            const uint2* p = (const uint2*)(tab + a[k]);
            if (MODE == 0) v = *p;
            else if (MODE == 1) { v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); }
            else if (MODE == 2) { const unsigned long long q = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.x = (uint32_t)q; v.y = (uint32_t)(q >> 32); }
            else { // same 128-byte line for groups of 4 lanes: does coalescing inside a wave help?
                const uint2* p2 = (const uint2*)(tab + ((a[k] & ~31u) | ((threadIdx.x & 3u) << 3) | (a[k] & 6u & 0u)));
                v = *p2;
            }
            acc ^= v.x ^ v.y;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    uint32_t* d_out; uint32_t* d_tab;
    const uint32_t words = 2u << 18;   // 2 MiB
    hipMalloc(&d_out, 256 * 1024 * 4); hipMalloc(&d_tab, words * 4);
    hipMemset(d_tab, 1, words * 4);
    for (int blocks : {256, 128, 64, 32}) {
        auto run = [&](auto kern, const char* nm, int ni) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int steps = 64;
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d_tab, d_out, words, steps);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 0, 0, d_tab, d_out, words, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * 1024 * steps * ni;
            printf("blocks %3d  %-28s x%d: %8.4f ms  %7.2f G lane-gathers/s  = %5.2f per ns and CU\n", blocks, nm, ni, ms, n / ms / 1e6, n / ms / 1e6 / blocks);
        };
        run(k_gather<0, 8>, "plain 8 B", 8);
        run(k_gather<1, 8>, "non-temporal", 8);
        run(k_gather<2, 8>, "agent-scope atomic load", 8);
        run(k_gather<3, 8>, "4 lanes share a line", 8);
    }
    return 0;
}
