/*
 * flat_walk.c — CPU walker over the FLAT image (include/acx_blob.h).
 * TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE (same rules as ac_oracle.c).
 *
 * Purpose: separate flattener bugs from kernel bugs.  ac_oracle.c walks the pointer
 * trie the way the reference does; this file walks the flattened arrays the way the HIP
 * kernels do (one table entry per input byte), on the CPU.  tests/ compare
 *      reference/_ref  ==  ac_oracle.c  ==  flat_walk.c(acx_flatten(...))  ==  HIP kernels.
 *
 * Semantics restated from the reference:
 *   ALL  : src/AutomatonSearchIter.c:157-197, 243-300 (+ src/trie.c:177-194)
 *   LONG : src/AutomatonSearchIterLong.c:89-153
 */
#include <stdint.h>
#include <stddef.h>
#include "acx_blob.h"

typedef struct {
    const acx_blob_header* h;
    const uint8_t*  cls;
    const uint32_t* table;
    const uint32_t* out_off;
    const int32_t*  out_val;
    const int32_t*  first_val;
} flat_t;

static int flat_open(const void* blob, flat_t* f) {
    const acx_blob_header* h = (const acx_blob_header*)blob;
    if (h->magic != ACX_BLOB_MAGIC || h->version != ACX_BLOB_VERSION) return -1;
    if (!h->table_in_blob) return -1;      /* these walkers read the host-built table (ACX_FLATTEN_TABLE=host) */
    const uint8_t* b = (const uint8_t*)blob;
    f->h = h;
    f->cls = b + h->off_cls;
    f->table = (const uint32_t*)(b + h->off_table);
    f->out_off = (const uint32_t*)(b + h->off_out_off);
    f->out_val = (const int32_t*)(b + h->off_out_val);
    f->first_val = (const int32_t*)(b + h->off_first_val);
    return 0;
}

/* returns number of matches (first `cap` stored); -1 on bad blob */
int64_t flat_iter(const void* blob, const uint8_t* hay, int64_t len,
                  int32_t* state_io, int64_t index_base,
                  int32_t* out_end, int32_t* out_val, int64_t cap) {
    flat_t f;
    if (flat_open(blob, &f) < 0) return -1;
    const uint32_t K = f.h->n_classes;
    const uint32_t SB = f.h->state_bits;
    uint32_t state = state_io ? (uint32_t)*state_io : 0;
    int64_t n = 0;
    for (int64_t i = 0; i < len; i++) {
        uint32_t e = f.table[(size_t)state * K + f.cls[hay[i]]];
        state = e & ACX_ENTRY_STATE_MASK(SB);
        uint32_t cnt = e >> ACX_ENTRY_CNT_SHIFT(SB);
        if (cnt) {
            uint32_t o0 = f.out_off[state], o1 = f.out_off[state + 1];
            /* the packed count must agree with the CSR unless it is the escape value */
            if (cnt != ACX_ENTRY_CNT_ESCAPE(SB) && cnt != o1 - o0) return -3;
            if (cnt == ACX_ENTRY_CNT_ESCAPE(SB) && o1 - o0 < ACX_ENTRY_CNT_ESCAPE(SB)) return -3;
            if (f.first_val[state] != f.out_val[o0]) return -4;
            for (uint32_t r = o0; r < o1; r++) {
                if (n < cap) { out_end[n] = (int32_t)(i + index_base); out_val[n] = f.out_val[r]; }
                n++;
            }
        }
    }
    if (state_io) *state_io = (int32_t)state;
    return n;
}

/*
 * The same walk, but shallow states are held IMPLICITLY as (depth, k-gram code) and their
 * transitions are resolved from the "itop" ND4 table (include/acx_blob.h) instead of table rows —
 * the CPU restatement of k_walk_itop.  Checked against flat_iter()/the oracle in tests; it also
 * cross-checks every implicit step against the explicit table (returns -(1000 + source line) on
 * disagreement).
 * Returns -6 if the image has no itop.
 */
int64_t flat_iter_itop(const void* blob, const uint8_t* hay, int64_t len, int64_t index_base,
                       int32_t* final_state, int32_t* out_end, int32_t* out_val, int64_t cap) {
#define BAD return -(int64_t)(1000 + __LINE__)
    flat_t f;
    if (flat_open(blob, &f) < 0) return -1;
    if (f.h->itop_depth == 0) return -6;
    const uint8_t* bb = (const uint8_t*)blob;
    const uint32_t* lds = (const uint32_t*)(bb + f.h->off_itop_lds);
    const uint32_t* ient = (const uint32_t*)(bb + f.h->off_itop_entry);
    const uint32_t* E = (const uint32_t*)(bb + f.h->off_itop_ebits);
    const uint32_t* cells = (const uint32_t*)(bb + f.h->off_itop_cells);
    const uint32_t* tflags = (const uint32_t*)(bb + f.h->off_tflags);
    const uint32_t K = f.h->n_classes, SB = f.h->state_bits;
    const uint32_t b = lds[0], D = lds[1], LD1 = lds[2], cell_bytes = lds[3], NS = lds[4], has_other = lds[5], maskD = lds[7];
    const uint32_t* ND = lds + lds[8];
    const uint32_t cs = lds[11];
    const uint32_t bD = b * D;
    if (cell_bytes != f.h->itop_cell_bytes || (cell_bytes != 4 && cell_bytes != 8) || NS != f.h->n_states) BAD;
#define XIDX(sh) ((hist & ((1u << (sh)) - 1u)) | (1u << (sh)))       /* sentinel index of the last sh/b symbols */
#define BIT(M, x) (((M)[(x) >> 5] >> ((x) & 31)) & 1u)
    int64_t n = 0;
    uint32_t hist = 0, sh = 0, s = 0, valid = 0, shadow = 0;   /* sh = b * implicit depth (0..bD); shadow: cross-check only */
    int expl = 0;                                              /* 1: depth > D, state s */
    for (int64_t i = 0; i < len; i++) {
        const uint32_t cls = f.cls[hay[i]];
        const uint32_t se = f.table[(size_t)shadow * K + cls];
        shadow = se & ACX_ENTRY_STATE_MASK(SB);
        uint32_t ev = 0;                                  /* entry to report, 0 = none */
        int pseudo = 0;                                   /* the kernel reports NS + x instead of the real entry */
        if (has_other && cls == 0) { sh = 0; expl = 0; hist = 0; valid = 0; if (shadow != 0) BAD; continue; }
        const uint32_t sym = cls - has_other;
        const uint32_t code_before = hist;                /* code of the level-D node when sh == bD */
        hist = ((hist << b) | sym) & maskD;
        if (valid < D) valid++;
        const uint32_t q = (ND[hist >> 3] >> ((hist & 7u) * 4)) & 15u;
        const uint32_t delta = q & 3u, oc = q >> 2;       /* oc: 0 no output, 1 exactly one, 2 more */
        int settled = 0, own_out = 0;
        if (oc == 3u || (delta == 3u && oc)) BAD;          /* delta 3 = shallower than D - 2: probe */
        if (expl) {                                       /* depth > D: the table row */
            const uint32_t e = f.table[(size_t)s * K + cls];
            if (valid < D) BAD;
            if (e >> ACX_ENTRY_CNT_SHIFT(SB)) ev = e;
            own_out = 1;
            if ((e & ACX_ENTRY_STATE_MASK(SB)) >= LD1) { s = e & ACX_ENTRY_STATE_MASK(SB); settled = 1; }
            else expl = 0;
        } else if (sh == bD) {                            /* depth == D: the cell of this node */
            uint32_t first, mask, outs;
            if (cell_bytes == 4) { const uint32_t cw = cells[code_before]; first = cw & 0xFFFFFFu; mask = (cw >> 24) & 15u; outs = cw >> 28; }
            else { first = cells[2 * (size_t)code_before]; mask = cells[2 * (size_t)code_before + 1] & 0xFFFFu; outs = cells[2 * (size_t)code_before + 1] >> 16; }
            if ((mask >> sym) & 1u) {
                const uint32_t child = first + (uint32_t)__builtin_popcount(mask & ((1u << sym) - 1u));
                if (child < LD1) BAD;
                s = child; expl = 1; settled = 1; own_out = 1;
                if ((outs >> sym) & 1u) ev = child | tflags[child];
            }
        }
        if (!settled) {                                   /* the new state is not deeper than D */
            if (valid >= D && delta != 3u) {              /* steady state: ND4 says how deep */
                sh = bD - b * delta;
                if (!BIT(E, XIDX(sh))) BAD;
                for (uint32_t c = sh + b; c <= bD; c += b) if (BIT(E, XIDX(c))) BAD;   /* it is the longest */
                if (!own_out && oc) {
                    const uint32_t real = ient[XIDX(sh)];
                    if (oc == 1) {                        /* reported as pseudo state NS + x, count 1 */
                        if ((real >> ACX_ENTRY_CNT_SHIFT(SB)) != 1u) BAD;
                        if (f.first_val[NS + XIDX(sh)] != f.first_val[real & ACX_ENTRY_STATE_MASK(SB)]) BAD;
                        pseudo = 1;
                    } else if ((real >> ACX_ENTRY_CNT_SHIFT(SB)) < 2u) BAD;
                    ev = real;
                }
            } else {                                      /* warm-up after a reset, or a deep fall */
                uint32_t c;
                if (valid < D) c = sh + b;
                else { if (bD < 3u * b) BAD; c = bD - 3u * b; for (uint32_t c2 = c + b; c2 <= bD; c2 += b) if (BIT(E, XIDX(c2))) BAD; }
                if (c > bD) BAD;
                for (;;) {
                    if (c <= cs) { if (!BIT(E, XIDX(c))) BAD; break; }
                    if (BIT(E, XIDX(c))) break;
                    c -= b;
                }
                sh = c;
                if (!own_out) {
                    ev = c ? ient[XIDX(c)] : 0;
                    if (!(ev >> ACX_ENTRY_CNT_SHIFT(SB))) ev = 0;
                }
            }
        }
        /* cross-check the position and the output decision against the explicit walk */
        {
            uint32_t cur = expl ? s : (sh == 0 ? 0 : (ient[XIDX(sh)] & ACX_ENTRY_STATE_MASK(SB)));
            if (cur != shadow) BAD;
            if ((ev != 0) != ((se >> ACX_ENTRY_CNT_SHIFT(SB)) != 0)) BAD;
            if (ev && ((ev ^ se) & ~ACX_ENTRY_EDGE(SB))) BAD;      /* same target, same flags */
        }
        if (ev && pseudo) {                               /* one record, value through the pseudo state */
            if (n < cap) { out_end[n] = (int32_t)(i + index_base); out_val[n] = f.first_val[NS + XIDX(sh)]; }
            n++;
        } else if (ev) {
            const uint32_t st = ev & ACX_ENTRY_STATE_MASK(SB);
            for (uint32_t r = f.out_off[st]; r < f.out_off[st + 1]; r++) {
                if (n < cap) { out_end[n] = (int32_t)(i + index_base); out_val[n] = f.out_val[r]; }
                n++;
            }
        }
    }
    if (final_state) *final_state = (int32_t)(expl ? s : (sh == 0 ? 0 : (ient[XIDX(sh)] & ACX_ENTRY_STATE_MASK(SB))));
#undef XIDX
#undef BIT
#undef BAD
    return n;
}

int64_t flat_iter_long(const void* blob, const uint8_t* hay, int64_t len, int64_t index_base,
                       int32_t* out_end, int32_t* out_val, int64_t cap) {
    flat_t f;
    if (flat_open(blob, &f) < 0) return -1;
    const uint32_t K = f.h->n_classes;
    const uint32_t SB = f.h->state_bits;
    int64_t n = 0;
    uint32_t state = 0;
    int64_t index = 0;
    int64_t last_index = -1;
    uint32_t last_state = 0;          /* the state whose FIRST output is the value to report */
    int have_last = 0;

    while (1) {
        int emit = 0;
        while (index < len) {
            uint32_t e = f.table[(size_t)state * K + f.cls[hay[index]]];
            uint32_t next = e & ACX_ENTRY_STATE_MASK(SB);
            if (!(e & ACX_ENTRY_EDGE(SB)) && have_last) { emit = 1; break; }   /* …IterLong.c:131-132 */
            if (next == 0) { state = 0; index++; continue; }               /* fail walk ended in NULL (:137-140) */
            /* a real edge from `state`, or from the first fail ancestor that has one (:134-143
             * followed by the next loop iteration at :116): the checks of :118-126 on `next` */
            if (e & ACX_ENTRY_EOW(SB)) {
                last_state = next; last_index = index; have_last = 1;
            } else if (e & ACX_ENTRY_FAILEOW(SB)) {
                last_state = next; last_index = index; have_last = 1;      /* reports fail(next): first output of next */
                emit = 1; break;
            }
            state = next; index++;
        }
        if (!emit && !have_last) break;                                    /* StopIteration */
        /* emit (…IterLong.c:99-111) */
        if (n < cap) {
            out_end[n] = (int32_t)(index_base + last_index);
            out_val[n] = f.first_val[last_state];
            if (f.first_val[last_state] != f.out_val[f.out_off[last_state]]) return -4;
        }
        n++;
        state = 0; index = last_index + 1; have_last = 0; last_index = -1;
    }
    return n;
}
