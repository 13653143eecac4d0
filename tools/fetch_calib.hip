// fetch_calib.hip — what rocprofv3's memory-side counters report for THIS path's access mix, on known byte counts.
//
// The guide (MI355X_MICROARCH.md, HBM) calibrates FETCH_SIZE for one pattern only — wide coalesced streaming reads report
// exactly half their bytes on gfx950 — and says: other widths and WRITE_SIZE are uncalibrated, calibrate on a known byte
// count in your own access pattern.  The scan kernels mix three patterns, so one launch of each, sized well past the
// 256 MiB Infinity Cache and run under  rocprofv3 --pmc  (separate passes, counters only: tools/r5_calib.sh):
//     stream16   every lane 16-byte NON-TEMPORAL loads, coalesced, each byte once        (the haystack)         N bytes
//     gather8    every lane 8-byte loads at hashed addresses of a table of T bytes       (hot cells, records)   G gathers
//                T = 2 MiB (an XCD's L2 holds it), 64 MiB, 1 GiB (every gather a miss)
//     store8     every wave 8-byte stores to consecutive records, lanes in order         (the record stream)    S bytes
// Printed per kernel: the byte / request counts the launch KNOWS; tools/calib_summary.py divides the counters by them:
// bytes per FETCH_SIZE unit for streaming reads, fabric requests and bytes per missing gather, bytes per WRITE_SIZE unit.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) calib_stream16(const u32x4* __restrict__ p, size_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int T>
__global__ void __launch_bounds__(1024) calib_gather8(const uint2* __restrict__ tab, uint32_t mask, uint32_t per_lane, uint32_t* sink) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (uint32_t k = 0; k < per_lane; k++) {
        x = x * 1664525u + 1013904223u;
        const uint2 v = tab[(x >> 7) & mask];
        acc ^= v.x + v.y;
    }
    if (acc == 0x12345678u) *sink = acc;
}

// a wave writes 64 consecutive 8-byte records per trip (512 bytes), like the record stream of the scan kernels
__global__ void __launch_bounds__(1024) calib_store8(uint2* __restrict__ out, size_t n_rec) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t per = n_rec / n_waves;
    uint2* o = out + wave * per;
    for (size_t i = lane; i < per; i += 64) o[i] = make_uint2((uint32_t)i, (uint32_t)wave);
}

int main(int argc, char** argv) {
    const size_t N = (size_t)1200 << 20;                                // streamed bytes
    const size_t S = (size_t)600 << 20;                                 // stored bytes
    const uint32_t per_lane = 64;
    const int blocks = 256 * 2;
    void *buf, *sink;
    CK(hipMalloc(&buf, N)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, N));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        calib_stream16<<<blocks, 1024>>>((const u32x4*)buf, N / 16, (uint32_t*)sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"calib_stream16\", \"bytes\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", N, ms, N / ms / 1e6);
    }
    const size_t tabs[3] = {(size_t)2 << 20, (size_t)64 << 20, (size_t)1 << 30};
    for (int t = 0; t < 3; t++) {
        const uint32_t mask = (uint32_t)(tabs[t] / 8 - 1);
        const size_t gathers = (size_t)blocks * 1024 * per_lane * (t == 0 ? 8 : 1);
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0));
            if (t == 0) calib_gather8<0><<<blocks, 1024>>>((const uint2*)buf, mask, per_lane * 8, (uint32_t*)sink);        // (a kernel name per table size: the PMC rows are told apart by it)
            else if (t == 1) calib_gather8<1><<<blocks, 1024>>>((const uint2*)buf, mask, per_lane, (uint32_t*)sink);
            else calib_gather8<2><<<blocks, 1024>>>((const uint2*)buf, mask, per_lane, (uint32_t*)sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"kernel\": \"calib_gather8<%d>\", \"table_bytes\": %zu, \"gathers\": %zu, \"ms\": %.4f, \"Ggathers_per_s\": %.1f}\n", t, tabs[t], gathers, ms, gathers / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        calib_store8<<<blocks, 1024>>>((uint2*)buf, S / 8);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const size_t n_waves = (size_t)blocks * 16, written = (S / 8 / n_waves) * n_waves * 8;
        printf("{\"kernel\": \"calib_store8\", \"bytes\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", written, ms, written / ms / 1e6);
    }
    (void)argc; (void)argv;
    return 0;
}
