// valu_rate.hip — issue cost of the integer instructions the scan kernels are made of, on gfx950.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o tools/valu_rate.bin && tools/valu_rate.bin
// Every wave runs 8 independent chains of one instruction; a block is 1024 threads (4 waves per SIMD) or 256 (1 per SIMD),
// one block per CU.  Printed: shader cycles (s_memtime) per wave-instruction and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 2048

#define CHAIN8(OPSTR)                                                                                   \
    asm volatile(OPSTR(0) OPSTR(1) OPSTR(2) OPSTR(3) OPSTR(4) OPSTR(5) OPSTR(6) OPSTR(7)                \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)     \
                 : "v"(c), "v"(d), "s"(sc) : "vcc", "s10", "s11")

#define OP_ADD(i)      "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 3, %8\n"
#define OP_ALIGN(i)    "v_alignbit_b32 %" #i ", %" #i ", %8, 7\n"
#define OP_ALIGNV(i)   "v_alignbit_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_BFE(i)      "v_bfe_u32 %" #i ", %" #i ", 3, 17\n"
#define OP_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_MUL24(i)    "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define OP_MAD24(i)    "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP_MULLO(i)    "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define OP_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define OP_BCNT(i)     "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define OP_FFBL(i)     "v_ffbl_b32 %" #i ", %" #i "\n"
#define OP_DPP(i)      "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_CNDMASK(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP_LSHR(i)     "v_lshrrev_b32 %" #i ", 3, %" #i "\n"
#define OP_AND(i)      "v_and_b32 %" #i ", %8, %" #i "\n"
#define OP_ADDS(i)     "v_add_u32 %" #i ", %" #i ", %10\n"
#define OP_LSHLADD(i)  "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n"
#define OP_BFI(i)      "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define OP_OR3(i)      "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define OP_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define OP_CNDS(i)     "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP_CMPCND(i)   "v_cmp_lt_u32_e32 vcc, %8, %" #i "\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define OP_CMPCNDS(i)  "v_cmp_lt_u32_e64 s[10:11], %8, %" #i "\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[10:11]\n"
#define OP_CMP(i)      "v_cmp_lt_u32_e32 vcc, %8, %" #i "\n"
#define OP_MINU(i)     "v_min_u32 %" #i ", %" #i ", %8\n"
#define OP_BITOP3(i)   "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0x96\n"
#define OP_LSHRV(i)    "v_lshrrev_b32 %" #i ", %9, %" #i "\n"
#define OP_SUBREV(i)   "v_sub_u32 %" #i ", %8, %" #i "\n"

template <int OP>
__global__ void __launch_bounds__(1024) k_valu(uint32_t* out, unsigned long long* cyc, uint32_t c, uint32_t d) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const uint32_t sc = c;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        if (OP == 0) CHAIN8(OP_ADD);
        if (OP == 1) CHAIN8(OP_LSHLOR);
        if (OP == 2) CHAIN8(OP_ALIGN);
        if (OP == 3) CHAIN8(OP_ALIGNV);
        if (OP == 4) CHAIN8(OP_BFE);
        if (OP == 5) CHAIN8(OP_PERM);
        if (OP == 6) CHAIN8(OP_MUL24);
        if (OP == 7) CHAIN8(OP_MAD24);
        if (OP == 8) CHAIN8(OP_MULLO);
        if (OP == 9) CHAIN8(OP_ANDOR);
        if (OP == 10) CHAIN8(OP_ADD3);
        if (OP == 11) CHAIN8(OP_BCNT);
        if (OP == 12) CHAIN8(OP_FFBL);
        if (OP == 13) CHAIN8(OP_DPP);
        if (OP == 14) CHAIN8(OP_CNDMASK);
        if (OP == 15) CHAIN8(OP_LSHR);
        if (OP == 16) CHAIN8(OP_AND);
        if (OP == 17) CHAIN8(OP_ADDS);
        if (OP == 18) CHAIN8(OP_LSHLADD);
        if (OP == 19) CHAIN8(OP_BFI);
        if (OP == 20) CHAIN8(OP_OR3);
        if (OP == 21) CHAIN8(OP_ADDCO);
        if (OP == 22) { asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11"); CHAIN8(OP_CNDS); }
        if (OP == 23) CHAIN8(OP_CMPCND);
        if (OP == 24) CHAIN8(OP_CMPCNDS);
        if (OP == 25) CHAIN8(OP_CMP);
        if (OP == 26) CHAIN8(OP_MINU);
        if (OP == 27) CHAIN8(OP_BITOP3);
        if (OP == 28) CHAIN8(OP_LSHRV);
        if (OP == 29) CHAIN8(OP_SUBREV);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// scalar ALU: 8 independent s_add chains
__global__ void __launch_bounds__(1024) k_salu(uint32_t* out, unsigned long long* cyc, uint32_t c) {
    uint32_t s0 = c, s1 = c + 1, s2 = c + 2, s3 = c + 3, s4 = c + 4, s5 = c + 5, s6 = c + 6, s7 = c + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        asm volatile("s_add_u32 %0, %0, %8\ns_add_u32 %1, %1, %8\ns_add_u32 %2, %2, %8\ns_add_u32 %3, %3, %8\n"
                     "s_add_u32 %4, %4, %8\ns_add_u32 %5, %5, %8\ns_add_u32 %6, %6, %8\ns_add_u32 %7, %7, %8\n"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(c) : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// mixed: 4 VALU + 4 SALU per group (do they issue side by side?)
__global__ void __launch_bounds__(1024) k_mix(uint32_t* out, unsigned long long* cyc, uint32_t c) {
    uint32_t s0 = c, s1 = c + 1, s2 = c + 2, s3 = c + 3;
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
        asm volatile("v_add_u32 %0, %0, %8\ns_add_u32 %4, %4, %9\nv_add_u32 %1, %1, %8\ns_add_u32 %5, %5, %9\n"
                     "v_add_u32 %2, %2, %8\ns_add_u32 %6, %6, %9\nv_add_u32 %3, %3, %8\ns_add_u32 %7, %7, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(c), "s"(c) : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 ^ s1 ^ s2 ^ s3 ^ x0 ^ x1 ^ x2 ^ x3;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS: 8 independent ds_read_b32 per group at random word addresses in a table of `words` (a power of two)
template <int WIDTH>
__global__ void __launch_bounds__(1024) k_lds(uint32_t* out, unsigned long long* cyc, uint32_t words, uint32_t mode) {
    extern __shared__ uint32_t sm[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) sm[i] = i * 2654435761u;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + 12345u, acc = 0;
    const uint32_t m = words - 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS / 4; it++) {
        uint32_t a[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t r = mode == 0 ? (x >> 8) : (mode == 1 ? (threadIdx.x & 63) + 64u * (uint32_t)k : 32u * (x >> 8));   // random / conflict-free / same bank
            a[k] = r & m;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (WIDTH == 4) v[k] = sm[a[k]];
            else if (WIDTH == 1) v[k] = ((const uint8_t*)sm)[a[k] * 4 + (x & 3)];
            else v[k] = ((const uint16_t*)sm)[a[k] * 2 + (x & 1)];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// gather latency: a dependent chain of 4- or 8-byte loads from a table of `bytes` bytes (pointer chasing), one wave per SIMD or four
__global__ void __launch_bounds__(1024) k_chase(const uint32_t* tab, uint32_t* out, unsigned long long* cyc, uint32_t words, int steps) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u;
    const uint32_t m = words - 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < steps; it++) x = tab[x & m] + (uint32_t)it;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}
// gather throughput: `NI` independent 4-, 8- or 16-byte gathers per lane and step from a table of `words` words
template <int BYTES, int NI>
__global__ void __launch_bounds__(1024) k_gather(const uint32_t* tab, uint32_t* out, unsigned long long* cyc, uint32_t words, int steps) {
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u, acc = 0;
    const uint32_t m = (words - 1) & ~(uint32_t)(BYTES / 4 - 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < steps; it++) {
        uint32_t a[NI];
#pragma unroll
        for (int k = 0; k < NI; k++) { x = x * 1664525u + 1013904223u; a[k] = (x >> 4) & m; }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            if (BYTES == 4) acc ^= tab[a[k]];
            else if (BYTES == 8) { const uint2 v = *(const uint2*)(tab + a[k]); acc ^= v.x ^ v.y; }
            else { const uint4 v = *(const uint4*)(tab + a[k]); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

static double avg_cycles(unsigned long long* d_cyc, int n_waves) {
    std::vector<unsigned long long> h(n_waves);
    hipMemcpy(h.data(), d_cyc, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    return s / n_waves;
}

int main() {
    int ncu = 256;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0); ncu = prop.multiProcessorCount;
    uint32_t* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_out, (size_t)ncu * 1024 * 4 * 4); hipMalloc(&d_cyc, (size_t)ncu * 16 * 8 * 4);
    const char* names[] = {"v_add_u32", "v_lshl_or_b32", "v_alignbit_b32 imm", "v_alignbit_b32 vgpr", "v_bfe_u32", "v_perm_b32", "v_mul_u32_u24",
                           "v_mad_u32_u24", "v_mul_lo_u32", "v_and_or_b32", "v_add3_u32", "v_bcnt_u32_b32", "v_ffbl_b32", "v_add_u32 dpp row_shr",
                           "v_cndmask_b32", "v_lshrrev_b32", "v_and_b32", "v_add_u32 sgpr", "v_lshl_add_u32", "v_bfi_b32", "v_or3_b32", "v_add_co_u32",
                           "v_cndmask_b32_e64 sgpr mask", "v_cmp + v_cndmask vcc (2 instr)", "v_cmp + v_cndmask sgpr (2 instr)", "v_cmp_lt_u32 vcc", "v_min_u32", "v_bitop3_b32",
                           "v_lshrrev_b32 vgpr amount", "v_sub_u32"};
    printf("# s_memtime ticks are a constant 100 MHz clock on gfx9: cycles below = ticks x (shader clock / 100 MHz) is NOT applied; wall time is\n");
    for (int bs : {1024, 256}) {
        const int wps = bs / 256;
        printf("== block %d threads (%d wave(s) per SIMD), %d blocks\n", bs, wps, ncu);
        auto run = [&](auto kern, const char* name, double insts_per_wave) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(kern, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u, 5u);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u, 5u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ticks = avg_cycles(d_cyc, ncu * bs / 64);
            // per SIMD: wps waves x insts_per_wave instructions in `ms`
            const double ns_per_inst_simd = ms * 1e6 / (wps * insts_per_wave);
            printf("%-24s  %8.4f ms  %7.3f ns per wave-instruction per SIMD  = %5.2f cycles @2.4GHz   (memtime ticks per wave %.0f)\n", name, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4, ticks);
        };
#define RUNV(OP) run(k_valu<OP>, names[OP], (double)ITERS * 8)
        RUNV(0); RUNV(1); RUNV(2); RUNV(3); RUNV(4); RUNV(5); RUNV(6); RUNV(7); RUNV(8); RUNV(9); RUNV(10); RUNV(11); RUNV(12); RUNV(13); RUNV(14);
        RUNV(15); RUNV(16); RUNV(17); RUNV(18); RUNV(19); RUNV(20); RUNV(21); RUNV(22); RUNV(23); RUNV(24); RUNV(25); RUNV(26); RUNV(27); RUNV(28); RUNV(29);
        {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_salu, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_salu, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s  %8.4f ms  %7.3f ns per wave-instruction per CU = %5.2f cycles @2.4GHz (all %d waves of a CU)\n", "s_add_u32", ms, ms * 1e6 / (bs / 64 * (double)ITERS * 8), ms * 1e6 / (bs / 64 * (double)ITERS * 8) * 2.4, bs / 64);
            hipLaunchKernelGGL(k_mix, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mix, dim3(ncu), dim3(bs), 0, 0, d_out, d_cyc, 3u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-24s  %8.4f ms  (4 VALU + 4 SALU per group; VALU alone would be %.4f)\n", "v_add + s_add mixed", ms, 0.0);
        }
        for (uint32_t words : {32768u, 8192u}) for (uint32_t mode : {0u, 1u, 2u}) {
            auto runl = [&](auto kern, const char* nm) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
                hipLaunchKernelGGL(kern, dim3(ncu), dim3(bs), words * 4, 0, d_out, d_cyc, words, mode);
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(ncu), dim3(bs), words * 4, 0, d_out, d_cyc, words, mode);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double n = (double)(ITERS / 4) * 8 * (bs / 64);
                printf("%-10s words=%6u mode=%u (0 random,1 linear,2 one bank)  %8.4f ms  %7.3f ns per wave-read per CU = %5.2f cycles @2.4GHz\n", nm, words, mode, ms, ms * 1e6 / n, ms * 1e6 / n * 2.4);
            };
            runl(k_lds<4>, "ds_read_b32"); if (mode == 0) { runl(k_lds<1>, "ds_read_u8"); runl(k_lds<2>, "ds_read_u16"); }
        }
    }
    // gathers
    const size_t max_words = (size_t)64 << 20;   // 256 MiB
    uint32_t* d_tab; hipMalloc(&d_tab, max_words * 4);
    {
        std::vector<uint32_t> h(max_words);
        uint32_t x = 1; for (size_t i = 0; i < max_words; i++) { x = x * 1664525u + 1013904223u; h[i] = x >> 3; }
        hipMemcpy(d_tab, h.data(), max_words * 4, hipMemcpyHostToDevice);
    }
    for (int bs : {1024, 256}) for (uint32_t kb : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u, 262144u}) {
        const uint32_t words = kb * 256u;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int steps = 256;
        hipLaunchKernelGGL(k_chase, dim3(ncu), dim3(bs), 0, 0, d_tab, d_out, d_cyc, words, steps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chase, dim3(ncu), dim3(bs), 0, 0, d_tab, d_out, d_cyc, words, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("chase  block %4d table %7u KiB: %8.4f ms, %7.1f ns per dependent 4-byte gather step (every lane its own line)\n", bs, kb, ms, ms * 1e6 / steps);
    }
    for (uint32_t kb : {1024u, 2048u, 4096u, 8192u, 16384u, 65536u}) {
        const uint32_t words = kb * 256u;
        auto rung = [&](auto kern, const char* nm, int ni) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int steps = 64;
            hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), 0, 0, d_tab, d_out, d_cyc, words, steps);
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(ncu), dim3(1024), 0, 0, d_tab, d_out, d_cyc, words, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)ncu * 1024 * steps * ni;
            printf("gather %-6s x%d table %6u KiB: %8.4f ms  %7.2f G lane-gathers/s\n", nm, ni, kb, ms, n / ms / 1e6);
        };
        rung(k_gather<4, 8>, "4B", 8); rung(k_gather<8, 8>, "8B", 8); rung(k_gather<16, 4>, "16B", 4); rung(k_gather<16, 8>, "16B", 8);
    }
    return 0;
}
