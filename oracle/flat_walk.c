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
 * transitions are resolved from the "itop" bitmaps (include/acx_blob.h) instead of table rows —
 * the CPU restatement of k_walk_itop.  Checked against flat_iter()/the oracle in tests; it also
 * cross-checks every implicit step against the explicit table (returns -5 on disagreement).
 * Returns -6 if the image has no itop.
 */
int64_t flat_iter_itop(const void* blob, const uint8_t* hay, int64_t len, int64_t index_base,
                       int32_t* final_state, int32_t* out_end, int32_t* out_val, int64_t cap) {
    flat_t f;
    if (flat_open(blob, &f) < 0) return -1;
    if (f.h->itop_depth == 0) return -6;
    const uint8_t* bb = (const uint8_t*)blob;
    const uint32_t* lds = (const uint32_t*)(bb + f.h->off_itop_lds);
    const uint32_t* ient = (const uint32_t*)(bb + f.h->off_itop_entry);
    const uint32_t* E = (const uint32_t*)(bb + f.h->off_itop_ebits);
    const uint32_t K = f.h->n_classes, SB = f.h->state_bits;
    const uint32_t b = lds[0], D = lds[1], LD = lds[2], has_other = lds[5], maskD = lds[7];
    const uint32_t* ND = lds + lds[8];
    const uint32_t* H = lds + lds[9];
    const uint32_t cs = lds[11], hmin = lds[12], h_first = lds[13];
    const uint16_t* rank16 = (const uint16_t*)(lds + lds[3]);
    const uint32_t* rank32 = lds + lds[4];
    const uint32_t bD = b * D;
#define XIDX(sh) ((hist & ((1u << (sh)) - 1u)) | (1u << (sh)))       /* sentinel index of the last sh/b symbols */
#define BIT(M, x) (((M)[(x) >> 5] >> ((x) & 31)) & 1u)
    int64_t n = 0;
    uint32_t hist = 0, sh = 0, s = 0, valid = 0, shadow = 0;   /* sh = b * implicit depth; shadow: cross-check only */
    int expl = 0;
    for (int64_t i = 0; i < len; i++) {
        const uint32_t cls = f.cls[hay[i]];
        const uint32_t se = f.table[(size_t)shadow * K + cls];
        shadow = se & ACX_ENTRY_STATE_MASK(SB);
        uint32_t ev = 0;                                  /* entry to report, 0 = none */
        if (has_other && cls == 0) { sh = 0; expl = 0; hist = 0; valid = 0; if (shadow != 0) return -5; continue; }
        hist = ((hist << b) | (cls - has_other)) & maskD;
        if (valid < D) valid++;
        uint32_t cand = 0;
        int resolve = 0;                                  /* 1: dropped out of the explicit zone, 2: implicit source */
        if (expl) {
            const uint32_t e = f.table[(size_t)s * K + cls];
            const uint32_t t = e & ACX_ENTRY_STATE_MASK(SB);
            if (e >> ACX_ENTRY_CNT_SHIFT(SB)) ev = e;
            if (t >= LD) s = t;
            else { expl = 0; cand = bD - b; resolve = 1; }
        } else { cand = sh + b; resolve = 2; }
        if (resolve) {
            uint32_t c, x;
            const uint32_t ndw = ND[hist >> 4];
            const uint32_t fld = (ndw >> ((hist & 15) * 2)) & 3u;
            if (valid >= D && fld != 3u) {                /* steady state: the table says how deep */
                c = bD - b * fld;
                if (c > cand) return -5;                  /* cannot be deeper than source + 1 (or D-1 after a drop) */
                x = XIDX(c);
                if (!BIT(E, x)) return -5;
            } else {                                      /* warm-up after a reset, or a fall of more than two levels */
                c = (valid >= D) ? (bD >= 3 * b ? bD - 3 * b : 0) : cand;
                if (c > cand) c = cand;
                for (;;) {
                    x = XIDX(c);
                    if (c <= cs) { if (!BIT(E, x)) return -5; break; }
                    if (BIT(E, x)) break;
                    c -= b;
                }
            }
            sh = c;
            if (resolve == 2 && c >= hmin && ((H[(x >> 5) - h_first] >> (x & 31)) & 1u)) ev = ient[x];
            if (c == bD) {                                /* hand over to the explicit rows: id = first + #ND zeros before */
                if (fld != 0) return -5;
                const uint32_t wi = hist >> 4;
                const uint32_t z = ~(ndw | (ndw >> 1)) & 0x55555555u;
                s = LD + rank32[wi >> 6] + rank16[wi] + (uint32_t)__builtin_popcount(z & ((1u << ((hist & 15) * 2)) - 1u));
                expl = 1;
                if ((ient[x] & ACX_ENTRY_STATE_MASK(SB)) != s) return -5;
            }
        }
        /* cross-check the position and the output decision against the explicit walk */
        {
            uint32_t cur = expl ? s : (sh == 0 ? 0 : (ient[XIDX(sh)] & ACX_ENTRY_STATE_MASK(SB)));
            if (cur != shadow) return -5;
            if ((ev != 0) != ((se >> ACX_ENTRY_CNT_SHIFT(SB)) != 0)) return -5;
        }
        if (ev) {
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
