/*
 * ppm_walk.c — CPU walk of the position-parallel scan image (include/acx_blob.h "ppm").
 * TEST INFRASTRUCTURE ONLY (see oracle/orc.py): it separates builder bugs (acx_ppm.cpp) from
 * kernel bugs (acx_ppm_kernels.hip) — the same split oracle/flat_walk.c makes for the dense table.
 *
 * What it must reproduce: at every position e, the keys that are suffixes of hay[..e], longest
 * first — what automaton_search_iter_next + automaton_build_output emit
 * (/root/reference/src/AutomatonSearchIter.c:157-197, 243-300).  Every position is handled on its
 * own: filter bit of the F newest symbols, 32-byte cell of the C newest, then the dense child rows
 * of the reversed trie; matches come out shortest first and are reversed.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "acx_blob.h"

#define PPM_MAX_MATCH 4096

/* fill != 0: symbols older than what the haystack holds at a position (before its start, behind a byte of no key)
 * read as pseudo-random symbols instead of 0 wherever a code is formed — the kernels read whatever the previous
 * haystack left there (acx_ppm_kernels.hip: the filters and the cell index are asked "as the symbols stand"), and
 * the image must give the same matches for ANY such filling. */
int64_t ppm_iter_fill(const uint8_t* blob, const uint8_t* hay, int64_t len, int32_t index_base,
                      int32_t* out_end, int32_t* out_val, int64_t cap, uint32_t fill) {
    acx_blob_header bh;
    memcpy(&bh, blob, sizeof bh);
    if (!bh.off_ppm) return -2;
    const uint8_t* sec = blob + bh.off_ppm;
    acx_ppm_header h;
    memcpy(&h, sec, sizeof h);
    if (h.magic != ACX_PPM_MAGIC) return -3;
    const uint8_t* symtab = sec + h.off_symtab;                        /* byte -> symbol, 0xFF = a byte of no key */
    const uint32_t* G = (const uint32_t*)(sec + h.off_g);
    const uint32_t* cells = (const uint32_t*)(sec + h.off_cells);
    const int32_t* top_val = (const int32_t*)(sec + h.off_top_val);
    const uint32_t* rows = (const uint32_t*)(sec + h.off_kids);        /* K 16-byte records per row */
    const uint32_t* singles = (const uint32_t*)(sec + h.off_chains);   /* one 16-byte record per single */
    const uint32_t K = h.K, C = h.C, F = h.F, F2 = h.F2, SB = h.sym_bits;
    const uint32_t* G2 = F2 ? (const uint32_t*)(sec + h.off_g2) : 0;      /* second-level filter */
    int64_t n = 0;
    int64_t last_other = -1;            /* last position holding a byte no key contains */
    static int32_t mv[PPM_MAX_MATCH];
    for (int64_t e = 0; e < len; e++) {
        if (h.has_other && symtab[hay[e]] == 0xFFu) { last_other = e; continue; }   /* (all 256 bytes in keys: symbol 255 is a symbol, as in the kernels) */
        int64_t L = e - last_other;     /* symbols available going back from e */
        if (L > (int64_t)h.longest) L = h.longest;
        /* codes of the newest d symbols, zero beyond L */
        uint32_t codeC = 0, codeF = 0, codeF2 = 0, code = 0;
        for (uint32_t d = 1; d <= (F2 > F ? F2 : F); d++) {
            uint32_t s = 0;
            if ((int64_t)d <= L) s = (uint32_t)symtab[hay[e - (d - 1)]];
            else if (fill) { uint32_t x = (uint32_t)e * 2654435761u ^ d * 40503u ^ fill * 2246822519u; x ^= x >> 15; x *= 2654435761u; x ^= x >> 13; s = x % K; }
            code = code * K + s;
            if (d == C) codeC = code;
            if (d == F) codeF = code;
            if (d == F2) codeF2 = code;
        }
        if (!((G[codeF >> 5] >> (codeF & 31)) & 1u)) continue;
        if (F2 && !((G2[codeF2 >> 5] >> (codeF2 & 31)) & 1u)) continue;
        const uint32_t* cell = cells + (size_t)codeC * 8;
        int nm = 0;
        uint32_t mask = cell[0];
        if (L < 32) mask &= (1u << L) - 1u;
        /* top levels, ascending depth */
        {
            uint32_t pc = 0;
            for (uint32_t d = 1; d <= C; d++) {
                const uint32_t s = (int64_t)d <= L ? (uint32_t)symtab[hay[e - (d - 1)]] : 0u;
                pc = pc * K + s;
                if ((mask >> (d - 1)) & 1u) {           /* the cell lists the values of its first five levels */
                    const int32_t v = nm < 5 ? (int32_t)cell[3 + nm] : top_val[h.top_base[d] + pc];
                    if (nm < PPM_MAX_MATCH) mv[nm++] = v;
                }
            }
        }
        /* deeper: one 16-byte record per step; K <= 4: the cell already knows children and grandchildren */
        uint32_t id = cell[1];
        int64_t d = C;                  /* depth of `id` */
        if (id && L > d) {
            int go = 1;
            if (K <= 4 && cell[2]) {
                const uint32_t s1 = (uint32_t)symtab[hay[e - d]];
                if (!((cell[2] >> s1) & 1u)) go = 0;
                else if (!((cell[2] >> (4 + s1)) & 1u)) {
                    /* child exists, is no key: worth a row gather only if a grandchild continues */
                    if (L > d + 1) {
                        const uint32_t s2 = (uint32_t)symtab[hay[e - d - 1]];
                        if (!((cell[2] >> (8 + s1 * 4 + s2)) & 1u)) go = 0;
                    } else go = 0;
                }
            }
            while (go && id) {
                const uint32_t* rec;
                uint32_t first;
                if (id >> 31) { rec = singles + (size_t)(id & 0x7FFFFFFFu) * 4; first = 0; }
                else {
                    if (L <= d) break;
                    rec = rows + ((size_t)id + (uint32_t)symtab[hay[e - d]]) * 4; first = 1;   /* a row's id is the index of its first record */
                }
                if (!(rec[1] & 0x200u)) break;          /* no child on this symbol */
                const uint32_t len = rec[1] & 0xFFu;
                if (L < d + (int64_t)first + len) break;
                /* the label: 32 / SB symbols in the first word, the next 16 / SB in the top half of the second (include/acx_blob.h) */
                uint32_t label = 0, more = 0;
                const uint32_t ms = 32u / SB;
                for (uint32_t i = 0; i < len; i++) {
                    const uint32_t sy = (uint32_t)symtab[hay[e - d - first - i]];
                    if (i < ms) label |= sy << (32 - SB * (i + 1));
                    else more |= sy << (32 - SB * (i - ms + 1));
                }
                if (label != rec[0] || more != (rec[1] & 0xFFFF0000u)) break;
                d += first + len;
                if (rec[1] & 0x100u) { if (nm < PPM_MAX_MATCH) mv[nm++] = (int32_t)rec[2]; }
                id = rec[3];
            }
        }
        for (int k = nm - 1; k >= 0; k--) {          /* longest first */
            if (n < cap) { out_end[n] = (int32_t)e + index_base; out_val[n] = mv[k]; }
            n++;
        }
    }
    return n;
}

int64_t ppm_iter(const uint8_t* blob, const uint8_t* hay, int64_t len, int32_t index_base,
                 int32_t* out_end, int32_t* out_val, int64_t cap) {
    return ppm_iter_fill(blob, hay, len, index_base, out_end, out_val, cap, 0u);
}

/* The 8-byte hot cells (what the stream kernel reads first) against the 32-byte cells they summarise:
 * returns the number of cells that disagree (0 = consistent), < 0 on a malformed image. */
int64_t ppm_check_hot(const uint8_t* blob) {
    acx_blob_header bh;
    memcpy(&bh, blob, sizeof bh);
    if (!bh.off_ppm) return -2;
    const uint8_t* sec = blob + bh.off_ppm;
    acx_ppm_header h;
    memcpy(&h, sec, sizeof h);
    if (h.magic != ACX_PPM_MAGIC || !h.off_hot) return -3;
    const uint32_t* cells = (const uint32_t*)(sec + h.off_cells);
    const uint32_t* hot = (const uint32_t*)(sec + h.off_hot);
    uint64_t nC = 1;
    for (uint32_t i = 0; i < h.C; i++) nC *= h.K;
    int64_t bad = 0;
    for (uint64_t c = 0; c < nC; c++) {
        const uint32_t* cell = cells + c * 8;
        const uint32_t hw = hot[2 * c], hx = hot[2 * c + 1];
        const uint32_t cmask = h.C >= 32 ? 0xFFFFFFFFu : (1u << h.C) - 1u;
        int ok = (hw & cmask) == cell[0];
        if (h.sym_bits == 2) {
            ok = ok && h.C <= 12 && ((hw >> 12) & 0xFu) == ((cell[2] >> 4) & 0xFu) && (hw >> 16) == ((cell[2] >> 8) & 0xFFFFu);
            ok = ok && (((hw >> 12) != 0) == (cell[1] != 0));          /* children <=> some child is a key or has a child */
        } else ok = ok && ((hw >> 31) == (cell[1] != 0 ? 1u : 0u)) && (hw & 0x7FFFFFFFu & ~cmask) == 0;
        if (cell[1]) ok = ok && hx == cell[1];
        else ok = ok && hx == (cell[0] ? cell[3] : 0u);
        if (!ok) bad++;
    }
    /* hot4 / cid (k_ppm_stream4, include/acx_blob.h): the value where a key ends, else the id; "go deeper" for two symbols */
    if (h.sym_bits == 2) {
        if (!h.off_hot4) return -4;
        const uint32_t* hot4 = (const uint32_t*)(sec + h.off_hot4);
        const uint32_t* cid = h.off_cid ? (const uint32_t*)(sec + h.off_cid) : 0;      /* 0: 12-byte cells, the id in their third word (ACX_FLATTEN_HOT12) */
        const uint32_t w = cid ? 2u : 3u;
        for (uint64_t c = 0; c <= nC; c++) {
            if (c == nC) { if (hot4[w * c] | hot4[w * c + 1] | (cid ? cid[c] : hot4[w * c + 2])) bad++; continue; }      /* the spare cell: all zero */
            const uint32_t* cell = cells + c * 8;
            uint32_t go = 0;
            for (uint32_t s1 = 0; s1 < 4; s1++)
                for (uint32_t s2 = 0; s2 < 4; s2++)
                    if (cell[1] && (((cell[2] >> (4 + s1)) & 1u) || ((cell[2] >> (8 + 4 * s1 + s2)) & 1u))) go |= 1u << (4 * s1 + s2);
            uint32_t xw = cell[0] | go << 16;
            if ((cell[1] >> 31) && h.C <= 9) {                          /* one child: the path of its single record (bit 13; its symbols; the shift that keeps as many as it has, up to 7) instead of the 16 bits */
                const uint32_t* rec = (const uint32_t*)(sec + h.off_chains) + (size_t)(cell[1] & 0x7FFFFFFFu) * 4;
                const uint32_t len = rec[1] & 0xFFu, np = len < 7u ? len : 7u;
                xw = cell[0] | (14u - 2u * np) << 9 | 1u << 13 | (rec[0] >> 18) << 16;
                if (np == 0 || !(rec[1] & 0x200u)) bad++;
            }
            int ok = hot4[w * c] == xw && (cell[0] >> 16) == 0 && (cid ? cid[c] : hot4[w * c + 2]) == cell[1];
            if (cid) ok = ok && hot4[2 * c + 1] == (cell[0] ? cell[3] : cell[1]);
            else ok = ok && hot4[3 * c + 1] == (cell[0] ? cell[3] : 0u);
            ok = ok && ((go != 0) == (cell[1] != 0));
            if (!ok) bad++;
        }
    } else if (h.off_hot4 || h.off_cid) bad++;
    /* gh: the hashed copy of a global filter (include/acx_blob.h ACX_PPM_GH_*) is exactly the image of G under the hash */
    if (h.off_gh) {
        if (!h.g_global) return -5;
        const uint32_t* G = (const uint32_t*)(sec + h.off_g);
        const uint32_t* gh = (const uint32_t*)(sec + h.off_gh);
        uint32_t* want = (uint32_t*)calloc(ACX_PPM_GH_WORDS, 4);
        if (!want) return -6;
        for (uint32_t w = 0; w < h.g_words; w++)
            for (uint32_t b = 0; b < 32; b++)
                if ((G[w] >> b) & 1u) { const uint32_t ix = ACX_PPM_GH_INDEX(w * 32u + b); want[ix >> 5] |= 1u << (ix & 31); }
        for (uint32_t w = 0; w < ACX_PPM_GH_WORDS; w++) if (want[w] != gh[w]) bad++;
        free(want);
    }
    /* the arithmetic symbol map, where the image has one */
    const uint8_t* symtab = sec + h.off_symtab;
    if (h.sym_arith) {
        const uint32_t sh = h.sym_arith - 1;
        for (int b = 0; b < 256; b++) {
            const uint32_t s = ((uint32_t)b >> sh) & 3u;
            const int is_key_byte = ((h.sym_lut >> (8 * s)) & 0xFFu) == (uint32_t)b;
            if (is_key_byte ? symtab[b] != s : symtab[b] != 0xFFu) bad++;
        }
    }
    return bad;
}
