// acx_build.hip — build the dense fail-resolved transition table IN HBM from the sparse
// form of the image (edge CSR, fail vector, per-target bits, level boundaries).
//
// the table is n_states x K x 4 B (34 GB for the 1 M-signature automaton
// of config 4); built on the host it costs tens of seconds of flatten time, an equally large
// H2D copy, and K x the bytes on every xGMI link of the one RCCL broadcast.  Built here it is
// two memory-bound kernels per BFS level: row(s) = row(fail(s)) with the EDGE bit cleared
// (fail(s) is shallower, hence finished), then s's own edges.  Same arithmetic as step 4 of
// acx_flatten (acx_trie.cpp); tests compare the two tables bit for bit.
#include "acx_kernels.h"
#include "acx_blob.h"

namespace {

__global__ void __launch_bounds__(256) k_build_rows_copy(uint32_t* table, const int32_t* fail, uint32_t K,
                                                        uint32_t s0, uint64_t n_entries, uint32_t keep_mask) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_entries; i += stride) {
        const uint32_t r = (uint32_t)(i / K), c = (uint32_t)(i - (uint64_t)r * K);
        const uint32_t s = s0 + r;
        table[(uint64_t)s * K + c] = table[(uint64_t)(uint32_t)fail[s] * K + c] & keep_mask;
    }
}

__global__ void __launch_bounds__(256) k_build_rows_patch(uint32_t* table, const uint32_t* edge_off, const uint8_t* edge_cls,
                                                         const uint32_t* edge_dst, const uint32_t* tflags, uint32_t K,
                                                         uint32_t s0, uint32_t s1, uint32_t edge_bit) {
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t s = s0 + blockIdx.x * 256 + threadIdx.x; s < s1; s += stride) {
        for (uint32_t e = edge_off[s]; e < edge_off[s + 1]; e++) {
            const uint32_t t = edge_dst[e];
            table[(uint64_t)s * K + edge_cls[e]] = t | tflags[t] | edge_bit;
        }
    }
}

}  // namespace

// lvl_first_host: uint32[n_levels + 1] (host copy of the level boundaries)
hipError_t acx_launch_build_table(uint32_t* table, const int32_t* fail, const uint32_t* edge_off, const uint8_t* edge_cls,
                                  const uint32_t* edge_dst, const uint32_t* tflags, const uint32_t* lvl_first_host,
                                  uint32_t n_levels, uint32_t K, uint32_t state_bits, hipStream_t s) {
    const uint32_t edge_bit = ACX_ENTRY_EDGE(state_bits);
    hipError_t e = hipMemsetAsync(table, 0, (size_t)K * 4, s);            // row(root): every class loops to the root
    if (e != hipSuccess) return e;
    for (uint32_t d = 0; d < n_levels; d++) {
        const uint32_t s0 = lvl_first_host[d], s1 = lvl_first_host[d + 1];
        if (s1 <= s0) continue;
        if (d > 0) {
            const uint64_t n_entries = (uint64_t)(s1 - s0) * K;
            uint64_t blocks = (n_entries + 255) / 256;
            if (blocks > 256 * 32) blocks = 256 * 32;
            hipLaunchKernelGGL(k_build_rows_copy, dim3((unsigned)blocks), dim3(256), 0, s, table, fail, K, s0, n_entries, ~edge_bit);
        }
        uint32_t pb = (s1 - s0 + 255) / 256;
        if (pb > 256 * 32) pb = 256 * 32;
        hipLaunchKernelGGL(k_build_rows_patch, dim3(pb), dim3(256), 0, s, table, edge_off, edge_cls, edge_dst, tflags, K, s0, s1, edge_bit);
    }
    return hipGetLastError();
}
