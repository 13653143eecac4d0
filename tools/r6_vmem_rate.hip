// r6_vmem_rate.hip — what a vector-memory instruction costs a CU by its shape: how many lanes are active, how many lines they touch.
//   hipcc -O3 --offload-arch=gfx950 tools/r6_vmem_rate.hip -o tools/r6_vmem_rate.bin && tools/r6_vmem_rate.bin
// 256 blocks of 1024 threads (16 waves per CU, as k_ppm_stream4); every wave issues NI independent 8-byte accesses per trip, ITERS trips.
// Printed: ns per wave-instruction and CU, and lane-accesses per second over the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define ITERS 256
#define NI 8

// MODE: 0 gather, every lane its own 128-byte line          1 gather, 32 lanes active        2 gather, 16 lanes active      3 gather, 8 lanes active
//       4 gather, all lanes ONE address                      5 gather, lanes consecutive (512 contiguous bytes)
//       6 gather, 27 lanes their own line + 37 lanes one common address (a round's slot: dummies ask the spare cell)
//       7 store, lanes consecutive                           8 store, 27 consecutive + 37 lanes to ONE address (the dump slot)
//       9 store, 27 consecutive lanes active only            10 store, every lane its own line
//       11 gather 64 lanes own line but only 2 MiB table     12 gather 16B every lane own line
template <int MODE>
__global__ void __launch_bounds__(1024) k_vmem(const uint2* tab, uint2* out, uint32_t words8, uint32_t* sink) {
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 777u;
    const uint32_t m = words8 - 1;
    uint32_t acc = 0;
    uint2* const wout = out + (size_t)wave * (ITERS * NI * 64);       // every wave its own output region
    for (int it = 0; it < ITERS; it++) {
        uint32_t a[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) { x = x * 1664525u + 1013904223u; a[k] = ((x >> 6) & m) & ~15u; }    // a 128-byte line (16 x 8 bytes)
        const uint32_t xs = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
        if (MODE <= 6 || MODE >= 11) {
            uint2 v[NI];
#pragma unroll
            for (int k = 0; k < NI; k++) {
                uint32_t idx = a[k];
                if (MODE == 4) idx = (xs + 16u * k) & m;
                if (MODE == 5) idx = ((xs + 1024u * k) & m & ~63u) + lane;
                if (MODE == 6) idx = lane < 27 ? a[k] : 0u;
                bool act = true;
                if (MODE == 1) act = lane < 32; if (MODE == 2) act = lane < 16; if (MODE == 3) act = lane < 8;
                v[k] = make_uint2(0, 0);
                if (MODE == 12) { if (act) { const uint4 q = *(const uint4*)(tab + (idx & ~1u)); v[k] = make_uint2(q.x ^ q.z, q.y ^ q.w); } }
                else if (act) v[k] = tab[idx];
            }
#pragma unroll
            for (int k = 0; k < NI; k++) acc ^= v[k].x ^ v[k].y;
        } else {
#pragma unroll
            for (int k = 0; k < NI; k++) {
                uint2* const base = wout + (size_t)(it * NI + k) * 64;
                if (MODE == 7) base[lane] = make_uint2(x, a[k]);
                if (MODE == 8) base[lane < 27 ? lane : 27] = make_uint2(x, a[k]);
                if (MODE == 9) { if (lane < 27) base[lane] = make_uint2(x, a[k]); }
                if (MODE == 10) out[((size_t)a[k] * 64 + wave) & (((size_t)4096 * ITERS * NI * 64) - 1)] = make_uint2(x, a[k]);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); const int ncu = prop.multiProcessorCount;
    const size_t out_words = (size_t)4096 * ITERS * NI * 64;          // 8-byte words
    uint2 *d_tab, *d_out; uint32_t* d_sink;
    const uint32_t tab_words8 = (64u << 20) / 8;                      // 64 MiB
    hipMalloc(&d_tab, (size_t)tab_words8 * 8); hipMalloc(&d_out, out_words * 8); hipMalloc(&d_sink, 4);
    hipMemset(d_tab, 1, (size_t)tab_words8 * 8);
    const char* names[] = {"gather 8B: 64 lanes, own lines (2 MiB table)", "gather 8B: 32 lanes active", "gather 8B: 16 lanes active", "gather 8B: 8 lanes active",
                           "gather 8B: all lanes one address", "gather 8B: lanes consecutive (512 B)", "gather 8B: 27 own lines + 37 one address",
                           "store 8B: lanes consecutive", "store 8B: 27 consecutive + 37 one address", "store 8B: 27 consecutive lanes active", "store 8B: own lines",
                           "gather 8B: 64 lanes, own lines (64 MiB table)", "gather 16B: 64 lanes, own lines (2 MiB table)"};
    auto run = [&](auto kern, int mode, uint32_t words8, double lanes) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), 0, 0, d_tab, d_out, words8, d_sink);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), 0, 0, d_tab, d_out, words8, d_sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_inst_cu = 16.0 * ITERS * NI;
        printf("%-52s %8.4f ms  %7.2f ns per wave-instruction per CU   %7.1f G active-lane accesses/s\n", names[mode], ms, ms * 1e6 / n_inst_cu, n_inst_cu * ncu * lanes / ms / 1e6);
    };
    const uint32_t w2m = (2u << 20) / 8;
    run(k_vmem<0>, 0, w2m, 64); run(k_vmem<1>, 1, w2m, 32); run(k_vmem<2>, 2, w2m, 16); run(k_vmem<3>, 3, w2m, 8);
    run(k_vmem<4>, 4, w2m, 64); run(k_vmem<5>, 5, w2m, 64); run(k_vmem<6>, 6, w2m, 27);
    run(k_vmem<7>, 7, w2m, 64); run(k_vmem<8>, 8, w2m, 27); run(k_vmem<9>, 9, w2m, 27); run(k_vmem<10>, 10, w2m, 64);
    run(k_vmem<11>, 11, tab_words8, 64); run(k_vmem<12>, 12, w2m, 64);
    return 0;
}
