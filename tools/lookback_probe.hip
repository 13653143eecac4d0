// lookback_probe.hip — what would a decoupled look-back cost k_ppm_stream?
//
// Today a wave of k_ppm_stream keeps its records in a pool and k_ppm_gather_pos moves them to their final place
// (0.042 ms and 200 MB per config-2 batch).  Writing records to their final place directly needs, per tile, the number of
// records of all tiles in front of it — a prefix sum across the 4096 waves that run at the same time.  The classic answer
// is a decoupled look-back: tiles are handed out in order (tile t of round r goes to wave t mod 4096), every tile
// publishes {count, flag} and looks back over its predecessors until it meets one whose inclusive prefix is known.
// On this chip a fence at device scope writes back / invalidates a whole L2 (that cost the kernel 60 % when tried), so
// the probe uses none: flag and value share one 64-bit word, written and read by relaxed device-scope atomics.
//
// The probe: 256 blocks x 16 waves, 18 rounds; per tile a wave spins for WORK clock ticks (the tile's scan), then
// publishes and looks back.  Reported: the kernel with and without the look-back, the average and maximum number of
// look-back steps, and a check of the final prefix.
//     hipcc --offload-arch=gfx950 -O3 tools/lookback_probe.hip -o tools/lookback_probe.bin && tools/lookback_probe.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned long long FLAG_AGG = 1ull << 62, FLAG_PFX = 2ull << 62, VAL_MASK = (1ull << 62) - 1;

template <bool LOOK>
__global__ void __launch_bounds__(1024) k_probe(unsigned long long* status, int rounds, int work_ticks, unsigned long long* out_last,
                                                unsigned int* steps_sum, unsigned int* steps_max) {
    const int lane = threadIdx.x & 63;
    const int n_waves = gridDim.x * 16;
    const int wave = blockIdx.x * 16 + (threadIdx.x >> 6);
    unsigned int my_steps = 0, my_max = 0;
    for (int r = 0; r < rounds; r++) {
        const long long t = (long long)r * n_waves + wave;
        // the tile's scan: spin for work_ticks shader clocks, jittered per wave and round as real tiles are
        const long long t0 = (long long)__builtin_amdgcn_s_memtime();
        const int jitter = (int)(((unsigned)(wave * 2654435761u + r * 40503u) >> 24) & 63) - 32;      // +- 32 x 64 ticks = +- 0.85 us
        while ((long long)__builtin_amdgcn_s_memtime() - t0 < work_ticks + 64 * jitter) __builtin_amdgcn_s_sleep(2);
        const unsigned long long cnt = 100 + (unsigned)((wave + r) & 63);                               // this tile's records
        if (!LOOK) { if (lane == 0 && t == (long long)rounds * n_waves - 1) *out_last = cnt; continue; }
        unsigned long long excl = 0;
        if (t > 0) {
            if (lane == 0) __hip_atomic_store(status + t, FLAG_AGG | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long j = t - 1;                                          // lane l looks at tile j - l
            unsigned int steps = 0;
            for (;;) {
                const long long mine = j - lane;
                unsigned long long s = mine >= 0 ? __hip_atomic_load(status + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : FLAG_PFX;
                steps++;
                const unsigned long long has_pfx = __ballot((s >> 62) == 2);
                const unsigned long long not_ready = __ballot((s >> 62) == 0);
                // the nearest predecessor whose prefix is known; everything nearer must at least have published its count
                const int first_pfx = has_pfx ? __ffsll((long long)has_pfx) - 1 : 64;
                const unsigned long long need = first_pfx < 64 ? ((1ull << first_pfx) - 1ull) : ~0ull;
                if (not_ready & need) { __builtin_amdgcn_s_sleep(4); continue; }     // look again
                unsigned long long v = lane <= first_pfx ? (s & VAL_MASK) : 0ull;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
                excl += v;
                if (first_pfx < 64) break;
                j -= 64;
            }
            my_steps += steps; my_max = steps > my_max ? steps : my_max;
        }
        if (lane == 0) {
            __hip_atomic_store(status + t, FLAG_PFX | (excl + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (long long)rounds * n_waves - 1) *out_last = excl + cnt;
        }
    }
    if (LOOK && lane == 0) { atomicAdd(steps_sum, my_steps); atomicMax(steps_max, my_max); }
}

int main() {
    const int blocks = 256, rounds = 18, n_waves = blocks * 16;
    const long long n_tiles = (long long)rounds * n_waves;
    unsigned long long *status, *last;
    unsigned int *ssum, *smax;
    CHECK(hipMalloc((void**)&status, n_tiles * 8));
    CHECK(hipMalloc((void**)&last, 8));
    CHECK(hipMalloc((void**)&ssum, 4));
    CHECK(hipMalloc((void**)&smax, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    unsigned long long want = 0;
    for (int r = 0; r < rounds; r++) for (int w = 0; w < n_waves; w++) want += 100 + (unsigned)((w + r) & 63);
    for (int work_us : {4, 8, 16}) {
        const int ticks = work_us * 2400;                           // (s_memtime counts shader clocks here: measured, 2.4 GHz)
        for (int look = 0; look < 2; look++) {
            float best = 1e9f;
            unsigned long long got = 0; unsigned int hs = 0, hm = 0;
            for (int rep = 0; rep < 5; rep++) {
                CHECK(hipMemset(status, 0, n_tiles * 8)); CHECK(hipMemset(ssum, 0, 4)); CHECK(hipMemset(smax, 0, 4));
                CHECK(hipEventRecord(e0));
                if (look) hipLaunchKernelGGL(k_probe<true>, dim3(blocks), dim3(1024), 0, 0, status, rounds, ticks, last, ssum, smax);
                else hipLaunchKernelGGL(k_probe<false>, dim3(blocks), dim3(1024), 0, 0, status, rounds, ticks, last, ssum, smax);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHECK(hipMemcpy(&got, last, 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(&hs, ssum, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&hm, smax, 4, hipMemcpyDeviceToHost));
            }
            if (look) printf("work %2d us/tile x %d rounds, look-back   : %.4f ms   steps per tile avg %.2f max %u   prefix %s\n", work_us, rounds, best,
                             (double)hs / (double)(n_tiles - 1), hm, got == want ? "ok" : "WRONG");
            else printf("work %2d us/tile x %d rounds, no look-back: %.4f ms\n", work_us, rounds, best);
        }
    }
    return 0;
}
