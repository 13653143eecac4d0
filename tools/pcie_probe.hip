// pcie_probe.hip — what the pieces of a host-to-host scan cost on this box: pageable and pinned H2D / D2H, registering a
// caller's buffer (hipHostRegister), copying it into pinned staging with 1..8 threads.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
// a kernel that reads pinned host memory itself (zero-copy): every block streams a slice with 16-byte loads
__global__ void __launch_bounds__(256) k_read_host(const uint4* src, size_t n16, unsigned long long* sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x123456789abcdefull) *sink = acc;
}
// a kernel that writes pinned host memory itself
__global__ void __launch_bounds__(256) k_write_host(uint4* dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 v; v.x = (unsigned)i; v.y = v.z = v.w = 7u; dst[i] = v; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 150u << 20;
    uint8_t* pageable = (uint8_t*)malloc(N); memset(pageable, 1, N);
    uint8_t* pinned; hipHostMalloc((void**)&pinned, N, hipHostMallocDefault); memset(pinned, 2, N);
    uint8_t* dev; hipMalloc((void**)&dev, N);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(dev, pageable, N, hipMemcpyHostToDevice); double a = now() - t;
        t = now(); hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double b = now() - t;
        t = now(); hipMemcpyAsync(pinned, dev, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double c = now() - t;
        t = now(); hipMemcpy(pageable, dev, N, hipMemcpyDeviceToHost); double d = now() - t;
        printf("150 MiB: H2D pageable %.2f ms (%.1f GB/s)  H2D pinned %.2f ms (%.1f GB/s)  D2H pinned %.2f ms (%.1f GB/s)  D2H pageable %.2f ms (%.1f GB/s)\n",
               a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9, c * 1e3, N / c / 1e9, d * 1e3, N / d / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        uint8_t* fresh = (uint8_t*)malloc(N); memset(fresh, 3, N);
        double t = now(); hipError_t e = hipHostRegister(fresh, N, hipHostRegisterDefault); double a = now() - t;
        t = now(); hipMemcpyAsync(dev, fresh, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double b = now() - t;
        t = now(); hipHostUnregister(fresh); double c = now() - t;
        printf("hipHostRegister(150 MiB) %.2f ms (rc %d), H2D from it %.2f ms (%.1f GB/s), unregister %.2f ms\n", a * 1e3, (int)e, b * 1e3, N / b / 1e9, c * 1e3);
        free(fresh);
    }
    for (int T : {1, 2, 4, 8, 16}) {
        double t = now();
        std::vector<std::thread> th;
        for (int k = 0; k < T; k++) th.emplace_back([&, k] { const size_t a = N * k / T, b = N * (k + 1) / T; memcpy(pinned + a, pageable + a, b - a); });
        for (auto& x : th) x.join();
        double a = now() - t;
        printf("memcpy pageable -> pinned, %2d threads: %.2f ms (%.1f GB/s)\n", T, a * 1e3, N / a / 1e9);
    }
    // both directions at once (full duplex?)
    uint8_t* pinned2; hipHostMalloc((void**)&pinned2, N, hipHostMallocDefault);
    uint8_t* dev2; hipMalloc((void**)&dev2, N);
    hipStream_t s2; hipStreamCreate(&s2);
    double t = now();
    hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s);
    hipMemcpyAsync(pinned2, dev2, N, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    double a = now() - t;
    printf("H2D and D2H of 150 MiB at once: %.2f ms (%.1f GB/s each way)\n", a * 1e3, N / a / 1e9);
    // zero-copy: kernels that read / write the pinned buffers themselves, alone and against a copy the other way
    unsigned long long* sink; hipMalloc((void**)&sink, 8);
    for (int blocks : {64, 256, 1024}) {
        hipDeviceSynchronize();
        t = now(); hipLaunchKernelGGL(k_read_host, dim3(blocks), dim3(256), 0, s, (const uint4*)pinned, N / 16, sink); hipStreamSynchronize(s); a = now() - t;
        t = now(); hipLaunchKernelGGL(k_write_host, dim3(blocks), dim3(256), 0, s, (uint4*)pinned2, N / 16); hipStreamSynchronize(s); double b = now() - t;
        printf("kernel reads pinned host memory, %4d blocks: %.2f ms (%.1f GB/s); kernel writes it: %.2f ms (%.1f GB/s)\n", blocks, a * 1e3, N / a / 1e9, b * 1e3, N / b / 1e9);
    }
    t = now();
    hipLaunchKernelGGL(k_read_host, dim3(256), dim3(256), 0, s, (const uint4*)pinned, N / 16, sink);
    hipMemcpyAsync(pinned2, dev2, N, hipMemcpyDeviceToHost, s2);
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    a = now() - t;
    printf("kernel reads host (150 MiB) while a copy engine writes host (150 MiB D2H): %.2f ms (%.1f GB/s each way)\n", a * 1e3, N / a / 1e9);
    t = now();
    hipLaunchKernelGGL(k_write_host, dim3(256), dim3(256), 0, s, (uint4*)pinned2, N / 16);
    hipMemcpyAsync(dev, pinned, N, hipMemcpyHostToDevice, s2);
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    a = now() - t;
    printf("kernel writes host (150 MiB) while a copy engine reads host (150 MiB H2D): %.2f ms (%.1f GB/s each way)\n", a * 1e3, N / a / 1e9);
    t = now();
    hipLaunchKernelGGL(k_read_host, dim3(256), dim3(256), 0, s, (const uint4*)pinned, N / 16, sink);
    hipLaunchKernelGGL(k_write_host, dim3(256), dim3(256), 0, s2, (uint4*)pinned2, N / 16);
    hipStreamSynchronize(s); hipStreamSynchronize(s2);
    a = now() - t;
    printf("one kernel reads host, another writes host, at once: %.2f ms (%.1f GB/s each way)\n", a * 1e3, N / a / 1e9);
    return 0;
}
