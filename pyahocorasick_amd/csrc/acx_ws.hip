// acx_ws.hip — ACX_SCAN_SKIP_WS: white space never touches the automaton.
//
// AutomatonSearchIter with ignore_white_space=True steps over every white-space letter without changing its state and
// reports end indices of the ORIGINAL string (src/AutomatonSearchIter.c:269-274: `while (index < end and
// iswspace(word[index])) index += 1` after every step; over the letters a bytes build can produce iswspace() is
// 0x09..0x0D and 0x20).  On the device that is: take the white space out of the whole batch (a stream compaction of the
// haystack buffer, the map from compacted to original positions kept), scan the compacted batch with the kernels that
// scan any other batch, and send the end indices of the records back through the map.  Three passes over the bytes
// and one over the records; the scan kernels stay what they are.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "acx_kernels.h"

namespace {

constexpr int WS_THREADS = 256;
constexpr int WS_PER = 16;                                  // bytes per thread
constexpr int WS_TILE = WS_THREADS * WS_PER;                // 4096 bytes per block

__device__ __forceinline__ bool is_ws(uint32_t b) { return b == 0x20u || (b - 9u) <= 4u; }

// the thread's 16 bytes (zero behind the end): one 16-byte load when the buffer allows it
__device__ __forceinline__ void load16(const uint8_t* hay, int64_t pos, int64_t total, bool aligned, uint32_t w[4]) {
    if (aligned && pos + 16 <= total) {
        const uint4 v = *(const uint4*)(hay + pos);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (int i = 0; i < 16; i++) if (pos + i < total) w[i >> 2] |= (uint32_t)hay[pos + i] << ((i & 3) * 8);
    }
}

// bit i set: byte i of the thread's 16 is kept (inside the buffer and no white space)
__device__ __forceinline__ uint32_t keep_mask(const uint32_t w[4], int64_t pos, int64_t total) {
    uint32_t m = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t b = (w[i >> 2] >> ((i & 3) * 8)) & 0xffu;
        m |= (uint32_t)(!is_ws(b) && pos + i < total) << i;
    }
    return m;
}

__global__ void __launch_bounds__(WS_THREADS) k_ws_count(const uint8_t* hay, int64_t total, int32_t* tile_count) {
    __shared__ int32_t s_sum[WS_THREADS / 64];
    const int64_t pos = (int64_t)blockIdx.x * WS_TILE + (int64_t)threadIdx.x * WS_PER;
    uint32_t w[4];
    int32_t c = 0;
    if (pos < total) { load16(hay, pos, total, ((uintptr_t)hay & 15u) == 0, w); c = __popc(keep_mask(w, pos, total)); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
}

// compacted bytes and, for each, the position it came from
__global__ void __launch_bounds__(WS_THREADS) k_ws_move(const uint8_t* hay, int64_t total, const int64_t* tile_off,
                                                        uint8_t* out_hay, uint32_t* out_map) {
    __shared__ int32_t s_sum[WS_THREADS / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t pos = (int64_t)blockIdx.x * WS_TILE + (int64_t)threadIdx.x * WS_PER;
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t m = 0;
    if (pos < total) { load16(hay, pos, total, ((uintptr_t)hay & 15u) == 0, w); m = keep_mask(w, pos, total); }
    const int32_t c = __popc(m);
    int32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) s_sum[wid] = inc;
    __syncthreads();
    int32_t before = inc - c;
    for (int k = 0; k < wid; k++) before += s_sum[k];
    int64_t at = tile_off[blockIdx.x] + before;
    while (m) {
        const int i = __ffs(m) - 1;
        m &= m - 1;
        out_hay[at] = (uint8_t)(w[i >> 2] >> ((i & 3) * 8));
        out_map[at] = (uint32_t)(pos + i);
        at++;
    }
}

// how many kept bytes lie in front of original position x: the map is sorted
__device__ __forceinline__ int64_t kept_before(const uint32_t* map, int64_t n, int64_t x) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)map[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// offsets (and context lengths) of the compacted batch
__global__ void __launch_bounds__(256) k_ws_offsets(const int64_t* off, int64_t stride, int64_t n_hay, const int32_t* skip,
                                                    const uint32_t* map, const int64_t* n_kept, int64_t* c_off, int32_t* c_skip) {
    const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (h > n_hay) return;
    const int64_t nk = *n_kept;
    const int64_t o = off ? off[h] : h * stride;
    const int64_t c = kept_before(map, nk, o);
    c_off[h] = c;
    if (c_skip && h < n_hay) c_skip[h] = (int32_t)(kept_before(map, nk, o + skip[h]) - c);
}

// end indices of the records: compacted -> original (context taken off, the caller's base added)
__global__ void __launch_bounds__(256) k_ws_remap(const acx_ws_remap_args a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total) return;
    int64_t lo = 0, hi = a.n_hay;                            // the haystack of record i: the last h with match_off[h] <= i
    while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (a.match_off[mid] <= i) lo = mid; else hi = mid - 1; }
    const int64_t h = lo;
    uint2 r = a.matches[i];
    const int64_t c = a.c_off[h] + (a.c_skip ? a.c_skip[h] : 0) + (int64_t)(int32_t)r.x;
    const int64_t o = a.off ? a.off[h] : h * a.stride;
    r.x = (uint32_t)((int64_t)a.map[c] - o - (a.skip ? a.skip[h] : 0) + (a.index_base ? a.index_base[h] : 0));
    a.matches[i] = r;
}

}  // namespace

int64_t acx_ws_num_tiles(int64_t total) { return (total + WS_TILE - 1) / WS_TILE; }

hipError_t acx_launch_ws_count(const uint8_t* hay, int64_t total, int32_t* tile_count, hipStream_t s) {
    const int64_t nt = acx_ws_num_tiles(total);
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ws_count, dim3((unsigned)nt), dim3(WS_THREADS), 0, s, hay, total, tile_count);
    return hipGetLastError();
}

hipError_t acx_launch_ws_move(const uint8_t* hay, int64_t total, const int64_t* tile_off, uint8_t* out_hay, uint32_t* out_map, hipStream_t s) {
    const int64_t nt = acx_ws_num_tiles(total);
    if (nt <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ws_move, dim3((unsigned)nt), dim3(WS_THREADS), 0, s, hay, total, tile_off, out_hay, out_map);
    return hipGetLastError();
}

hipError_t acx_launch_ws_offsets(const int64_t* off, int64_t stride, int64_t n_hay, const int32_t* skip, const uint32_t* map,
                                 const int64_t* n_kept, int64_t* c_off, int32_t* c_skip, hipStream_t s) {
    hipLaunchKernelGGL(k_ws_offsets, dim3((unsigned)((n_hay + 1 + 255) / 256)), dim3(256), 0, s, off, stride, n_hay, skip, map, n_kept, c_off, c_skip);
    return hipGetLastError();
}

hipError_t acx_launch_ws_remap(const acx_ws_remap_args& a, hipStream_t s) {
    if (a.total <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_ws_remap, dim3((unsigned)((a.total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}
