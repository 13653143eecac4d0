// acx_ppm_layout.h — LDS budget of the position-parallel scan (include/acx_blob.h "ppm"), shared by
// the builder (acx_ppm.cpp: how large may the filter bitmap be?) and the launcher
// (acx_ppm_kernels.hip: where does everything live?).  Internal to libacx.
#ifndef ACX_PPM_LAYOUT_H_INCLUDED
#define ACX_PPM_LAYOUT_H_INCLUDED

#include <stddef.h>
#include <stdint.h>
#include "acx_blob.h"

#define ACX_PPM_BLOCK      1024u                    /* threads per block: 16 waves share one filter bitmap */
#define ACX_PPM_WAVES      (ACX_PPM_BLOCK / 64u)
#define ACX_PPM_LDS_BYTES  (160u * 1024u)
#define ACX_PPM_MAX_LONGEST 1024u                   /* longer keys: the serial walk kernels */

// Per-wave staging, in 32-bit words.
//   sym   : packed symbols of the tile and its left halo (longest - 1 bytes), one pad word in front
//           (a window of 32 bits ends at every position) and one behind (the funnel shift reads it)
//   oth   : one byte per staged dword: which of its 4 bytes occur in no key; then one uint16 per
//           staged dword: the last such position at or before the dword's end (+1; 0 = none)
//   queue : positions that passed the filter (one byte each)
//   cnt   : matches per position, then their exclusive prefix sum (uint16, or uint32 for long keys)
struct acx_ppm_lds {
    uint32_t dwords;        // staged haystack dwords per tile
    uint32_t sym_words, oth_words, queue_words, cnt_words, wave_words;
    uint32_t g_off, map_off, wave_off, total_words;     // word offsets inside the block's LDS
    uint32_t cnt32;         // 1: cnt/off entries are 32 bits wide
};

static inline acx_ppm_lds acx_ppm_lds_layout(uint32_t g_words, uint32_t sym_bits, uint32_t longest) {
    acx_ppm_lds L;
    const uint32_t halo = longest > 0 ? longest - 1 : 0;
    L.dwords = (ACX_PPM_TILE + halo + 3) / 4 + 1 + 8;                   // unaligned start: one more; halo rounded up to whole words: some more
    L.sym_words = ((L.dwords * 4 * sym_bits + 31) / 32 + 2 + 3u) & ~3u;
    L.oth_words = ((L.dwords + 3) / 4 + (L.dwords * 2 + 3) / 4 + 3u) & ~3u;
    L.queue_words = ACX_PPM_TILE / 4;
    L.cnt32 = (uint64_t)longest * ACX_PPM_TILE >= 65536u ? 1u : 0u;
    L.cnt_words = L.cnt32 ? ACX_PPM_TILE : ACX_PPM_TILE / 2;
    L.wave_words = (L.sym_words + L.oth_words + L.queue_words + L.cnt_words + 3u) & ~3u;
    L.g_off = 0;
    L.map_off = (g_words + 3u) & ~3u;
    L.wave_off = L.map_off + 64;                                       // byte -> symbol map: 256 bytes
    L.total_words = L.wave_off + ACX_PPM_WAVES * L.wave_words;
    return L;
}


// The tiles of a block, 16 tpw of them, over its 16 waves.  Equal runs end unequally: the four waves of a SIMD do not run
// equally fast (the issue arbiter prefers the older wave — profiles/r4_wave_end_times.txt: with 18 tiles each the oldest
// wave of every SIMD ends after 205 us, the next after 217, 232, 247), and a SIMD whose first waves are gone hides less
// latency for the rest: the kernel's last 40 us run on half-empty SIMDs.  So wave slot i — the (i / 4)-th oldest wave of
// SIMD i % 4 — takes tpw + d tiles, d = +a, +b, -b, -a for i / 4 = 0 .. 3 (a = b = 0: equal runs).  First tile of slot
// `slot` (0 .. 16) within its block:
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t acx_ppm_slot_first_tile(uint32_t slot, uint32_t tpw, uint32_t a, uint32_t b) {
    const uint32_t g = slot >> 2, r = slot & 3u;
    const uint32_t t0 = tpw + a, t1 = tpw + b, t2 = tpw - b, t3 = tpw - a;
    const uint32_t before = g == 0 ? 0u : g == 1 ? 4u * t0 : g == 2 ? 4u * (t0 + t1) : g == 3 ? 4u * (t0 + t1 + t2) : 16u * tpw;
    const uint32_t own = g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3;
    return before + r * own;                                            // (slot 16: g = 4, r = 0 -> 16 tpw)
}

// the inverse: which slot of its block holds tile t (0 .. 16 tpw - 1, counted from the block's first tile)
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t acx_ppm_tile_slot(uint32_t t, uint32_t tpw, uint32_t a, uint32_t b) {
    const uint32_t t0 = tpw + a, t1 = tpw + b, t2 = tpw - b, t3 = tpw - a;
    if (t < 4u * t0) return t / t0;
    t -= 4u * t0;
    if (t < 4u * t1) return 4u + t / t1;
    t -= 4u * t1;
    if (t < 4u * t2) return 8u + t / t2;
    t -= 4u * t2;
    return 12u + t / t3;
}

// k_ppm_stream: tiles of nsub x 256 positions, halo of halo_pos positions (a multiple of 32) carried in LDS, a ring
// queue.  oth: one bit per staged position (a byte of no key); cnt: the start tables of offsets batches.
static inline acx_ppm_lds acx_ppm_stream_layout(uint32_t g_words, uint32_t sym_bits, uint32_t halo_pos, uint32_t nsub, int offs) {
    acx_ppm_lds L;
    const uint32_t tpos = nsub * 256u, spw = 32u / sym_bits;
    L.dwords = (halo_pos + tpos) / 4;
    L.sym_words = (2 + halo_pos / spw + tpos / spw + 1 + 3u) & ~3u;
    L.oth_words = ((halo_pos + tpos) / 32 + 1 + 3u) & ~3u;
    L.queue_words = 384 / 2 + 2;                                       // PPM_QCAP uint16 entries + a spare slot
    L.cnt32 = 0; L.cnt_words = offs ? (tpos / 32) + (tpos / 32) / 2 + 2 : 0;   // offsets batches: start bitmap (a word per 32 positions), two byte tables (a byte per word each)
    L.wave_words = (L.sym_words + L.oth_words + L.queue_words + L.cnt_words + 3u) & ~3u;
    L.g_off = 0;
    L.map_off = (g_words + 3u) & ~3u;
    L.wave_off = L.map_off + 64 + 24;                                  // byte -> symbol map (256 bytes), top_base[] (22 words)
    L.total_words = L.wave_off + ACX_PPM_WAVES * L.wave_words;
    return L;
}

#endif
