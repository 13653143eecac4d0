// acx_ppm_device.h — device helpers shared by the position-parallel scan kernels (acx_ppm_kernels.hip: every alphabet;
// acx_ppm_stream4.hip: four-letter alphabets).  Internal to libacx; every translation unit gets its own copy.
#ifndef ACX_PPM_DEVICE_H_INCLUDED
#define ACX_PPM_DEVICE_H_INCLUDED

#include "acx_kernels.h"
#include "acx_ppm_layout.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));   // (global loads need dword alignment only)
typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));

// LDS traffic of one wave is ordered in hardware; this only stops the compiler from moving LDS
// accesses across the hand-over points between lanes of the same wave.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ uint32_t div_magic(uint32_t e, uint64_t M, uint32_t d, uint32_t& r) {
    if (M == 0) { r = 0; return e; }                                  // d == 1
    const uint64_t t = (uint64_t)e * (uint32_t)M;
    const uint64_t u = (uint64_t)e * (uint32_t)(M >> 32) + (t >> 32);
    const uint32_t q = (uint32_t)(u >> 32);
    r = e - q * d;
    return q;
}

// the dword at byte b of a buffer of cap bytes that ends inside it (the last tile of a batch): out of line, it is cold
__device__ __attribute__((noinline)) uint32_t load_dw_tail(const uint8_t* hay, int64_t cap, uint32_t b) {
    uint32_t w = 0;
    for (int k = 0; k < 4; k++) if ((int64_t)b + k < cap) w |= (uint32_t)hay[b + k] << (8 * k);
    return w;
}

// wave64 exclusive prefix sum on the DPP network (row shifts, then the two row broadcasts gfx9 has):
// no LDS round trips.  total = sum over the wave.
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t& total) {
    uint32_t x = v;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);    // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);    // row_bcast:31 -> rows 2, 3
    total = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
    return x - v;
}

struct Geo {                    // wave-uniform description of a tile
    const uint8_t* abase;       // dword-aligned address of the first staged byte
    uint32_t q0;                // staged position of the first end position
    uint32_t ndw;               // dwords to stage
    int32_t  npos;              // end positions in the tile (<= 256)
    uint32_t idx_first;         // CHUNK: end_index of the first end position
    uint32_t halo;              // CHUNK: bytes of the same haystack in front of it (<= longest - 1)
    uint32_t e0;                // STRIDE: global byte index of the first end position
};

template <int SB, bool POW2, bool CHUNK>
struct Ppm {
    const acx_ppm_args& a;
    const uint32_t* s_g;        // LDS: filter bitmap
    const uint8_t*  s_map;      // LDS: byte -> symbol, 0xFF = other
    uint32_t* s_sym;            // LDS, this wave: packed symbols (pad word in front)
    uint8_t*  s_oth;            //                 per staged dword: which bytes are "other"
    uint16_t* s_last;           //                 per staged dword: last other position at or before its end, +1
    uint8_t*  s_queue;          //                 positions that passed the filter, ascending
    uint32_t* s_qoff;           //                 exclusive record offset of every queue entry (u16 or u32)
    Geo T;
    uint32_t has_other;         // some staged byte of this tile occurs in no key

    __device__ __forceinline__ Ppm(const acx_ppm_args& a_) : a(a_) {}

    __device__ __forceinline__ uint32_t qoff_get(uint32_t i) const { return a.lds.cnt32 ? s_qoff[i] : ((const uint16_t*)s_qoff)[i]; }
    __device__ __forceinline__ void qoff_set(uint32_t i, uint32_t v) { if (a.lds.cnt32) s_qoff[i] = v; else ((uint16_t*)s_qoff)[i] = (uint16_t)v; }

    // the 32 bits of packed symbols that end with staged position q: newest symbol on top
    __device__ __forceinline__ uint32_t window(uint32_t q) const {
        const uint32_t endbit = SB * (q + 1) + 32;
        const uint32_t w = endbit >> 5;
        return __builtin_amdgcn_alignbit(s_sym[w], s_sym[w - 1], endbit & 31u);
    }
    // A deep record's label against the symbols that end at position q (include/acx_blob.h: the first 32 / SB symbols in x, the next
    // 16 / SB in the top half of y): 0 iff the len newest agree.  len >= 1 (a caller lets a record without symbols pass by itself);
    // the second window is read at q when the label has no second part, so that no walker reads in front of the staged symbols.
    __device__ __forceinline__ uint32_t label_diff(uint32_t q, uint32_t len, uint32_t x, uint32_t y) const {
        constexpr uint32_t MS = 32u / SB;
        const uint32_t l1 = len < MS ? len : MS, l2 = len - l1;
        const uint32_t d1 = (window(q) ^ x) >> ((0u - SB * l1) & 31u);
        const uint32_t d2 = ((window(l2 ? q - MS : q) ^ y) >> 16) >> ((16u - SB * l2) & 15u);
        return d1 | (l2 ? d2 : 0u);
    }
    __device__ __forceinline__ uint32_t sym_at(uint32_t q) const {
        const uint32_t bit = SB * q + 32;
        return (s_sym[bit >> 5] >> (bit & 31u)) & ((1u << SB) - 1u);
    }
    // symbols available going back from end position p (0 = none: no match can end here)
    __device__ __forceinline__ uint32_t limit(uint32_t p, uint32_t r /* STRIDE: offset in its haystack */) const {
        uint32_t L = CHUNK ? T.halo + p + 1 : r + 1;
        if (L > a.longest) L = a.longest;
        if (has_other) {
            const uint32_t q = T.q0 + p, dw = q >> 2;
            const uint32_t nib = s_oth[dw] & ((2u << (q & 3u)) - 1u);
            uint32_t last;
            if (nib) last = 4 * dw + (31 - __clz(nib)) + 1;
            else last = dw ? s_last[dw - 1] : 0u;
            const uint32_t lo = q + 1 - last;
            if (lo < L) L = lo;
        }
        return (int32_t)p < T.npos ? L : 0u;
    }
    // code of the d newest symbols of window X when only L of them exist (the rest read as 0)
    __device__ __forceinline__ uint32_t code_of(uint32_t X, uint32_t L, uint32_t d) const {
        if (POW2) {
            const uint32_t Xm = SB * L >= 32 ? X : (X & ~(0xFFFFFFFFu >> (SB * L)));
            return Xm >> (32 - SB * d);
        }
        uint32_t c = 0;
        for (uint32_t i = 1; i <= d; i++) {
            const uint32_t s = i <= L ? __builtin_amdgcn_ubfe(X, 32 - SB * i, SB) : 0u;
            c = c * a.K + s;
        }
        return c;
    }
    __device__ __forceinline__ void codes_CF(uint32_t X, uint32_t L, uint32_t& cC, uint32_t& cF) const {
        if (POW2) {
            const uint32_t Xm = SB * L >= 32 ? X : (X & ~(0xFFFFFFFFu >> (SB * L)));
            cC = Xm >> (32 - SB * a.C);
            cF = Xm >> (32 - SB * a.F);
            return;
        }
        uint32_t c = 0; cC = 0;
        for (uint32_t i = 1; i <= a.F; i++) {
            const uint32_t s = i <= L ? __builtin_amdgcn_ubfe(X, 32 - SB * i, SB) : 0u;
            c = c * a.K + s;
            if (i == a.C) cC = c;
        }
        cF = c;
    }

    struct Ent { uint32_t p, X, L, idx; u32x4 c0, c1; };

    __device__ __forceinline__ Ent load_ent(uint32_t p) const {
        Ent E;
        E.p = p;
        uint32_t r = 0;
        if (!CHUNK) {
            const uint32_t h = div_magic(T.e0 + p, a.stride_magic, (uint32_t)a.stride, r);
            E.idx = r + (a.index_base ? (uint32_t)a.index_base[h] : 0u);
        } else E.idx = T.idx_first + p;
        E.L = limit(p, r);
        E.X = window(T.q0 + p);
        uint32_t cC, cF;
        codes_CF(E.X, E.L, cC, cF);
        const u32x4* cell = (const u32x4*)(a.cells + (size_t)cC * 8);
        E.c0 = cell[0]; E.c1 = cell[1];
        return E;
    }

    // Every match ending at E's position, shortest first: f(k, value) for the k-th.  Matches k in
    // [from, upto) get their values; returns the number of matches.
    template <typename F>
    __device__ __forceinline__ uint32_t matches(const Ent& E, uint32_t from, uint32_t upto, F&& f) const {
        uint32_t n = 0;
        uint32_t mask = E.c0.x;
        if (E.L < 32) mask &= (1u << E.L) - 1u;
        while (mask) {                                                // top levels, ascending depth
            const uint32_t d = (uint32_t)__ffs(mask);                // depth = bit + 1
            mask &= mask - 1;
            if (n >= from && n < upto) {
                int32_t v;                                            // the cell lists the values of its first five levels
                if (n == 0) v = (int32_t)E.c0.w;
                else if (n == 1) v = (int32_t)E.c1.x;
                else if (n == 2) v = (int32_t)E.c1.y;
                else if (n == 3) v = (int32_t)E.c1.z;
                else if (n == 4) v = (int32_t)E.c1.w;
                else v = a.top_val[a.top_base[d] + code_of(E.X, E.L, d)];
                f(n, v);
            }
            n++;
        }
        uint32_t id = E.c0.y, d = a.C;
        if (id && E.L > d) {
            const uint32_t q = T.q0 + E.p;
            bool go = true;
            if (SB == 2 && E.c0.z) {                                  // K <= 4: the cell knows children and grandchildren
                const uint32_t s1 = sym_at(q - d);
                if (!((E.c0.z >> s1) & 1u)) go = false;
                else if (!((E.c0.z >> (4 + s1)) & 1u)) {
                    if (E.L > d + 1) { const uint32_t s2 = sym_at(q - d - 1); go = ((E.c0.z >> (8 + s1 * 4 + s2)) & 1u) != 0; }
                    else go = false;
                }
            }
            while (go && id) {                                        // one 16-byte record per step
                const bool single = (id >> 31) != 0;
                if (!single && E.L <= d) break;
                const uint32_t first = single ? 0u : 1u;
                const size_t ri = single ? (size_t)(id & 0x7FFFFFFFu) : (size_t)id + sym_at(q - d);      // (a row's id: the index of its first record)
                const u32x4 rec = *(const u32x4*)((single ? a.chains : a.kids) + ri * 4);
                if (!(rec.y & 0x200u)) break;
                const uint32_t len = rec.y & 0xFFu;
                if (E.L < d + first + len) break;
                if (len && label_diff(q - d - first, len, rec.x, rec.y)) break;
                d += first + len;
                if (rec.y & 0x100u) { if (n >= from && n < upto) f(n, (int32_t)rec.z); n++; }
                id = rec.w;
            }
        }
        return n;
    }
};

}  // namespace

#endif
