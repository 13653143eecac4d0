// acx_trie.cpp — host side of libacx: arena trie, BFS failure links, flattener.
//
// CPU only (no HIP in this file).  Mirrors what stays on the CPU in the reference:
//   trie_add_word / trie_find / trie_remove_word / trie_longest   src/trie.c:14-175
//   automaton_add_word value + version rules                      src/Automaton.c:201-300
//   automaton_make_automaton (BFS fail links)                     src/Automaton.c:560-649
// and adds the step the reference does not have: flattening the finalised trie into the
// contiguous image of include/acx_blob.h that the HIP kernels read.
//
// Design differences from the reference (deliberate, this is not a port):
//   * nodes live in one std::vector (index links, no per-node malloc): 24 B/node
//     instead of 32 B + a separately realloc'ed Pair array;
//   * children are a sibling list in insertion order + a 256-way direct table at the
//     root (the only node that is routinely wide);
//   * BFS order is recorded by make_automaton and reused as the state numbering.
#include "acx_trie_impl.h"
#include "acx_internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <ctime>
#include <new>
#include <atomic>
#include <vector>

extern "C" {

int acx_trie_new(acx_trie_t** out) {
    if (!out) return acx_fail(ACX_E_INVAL, "acx_trie_new: out is NULL");
    acx_trie* t = new (std::nothrow) acx_trie();
    if (!t) return acx_fail(ACX_E_NOMEM, "acx_trie_new: out of memory");
    *out = t;
    return ACX_OK;
}

void acx_trie_free(acx_trie_t* t) { delete t; }

void acx_trie_clear(acx_trie_t* t) {
    if (!t) return;
    // automaton_clear, src/Automaton.c:405-416
    t->nodes.clear(); t->nodes.shrink_to_fit();
    t->wide_index.clear();
    t->bfs.clear(); t->bfs.shrink_to_fit();
    t->level_first.clear();
    for (auto& c : t->root_child) c = -1;
    t->kind = ACX_KIND_EMPTY;
    t->count = 0; t->longest_word = 0; t->live_nodes = 0;
    t->version += 1;
}

int acx_trie_add_word(acx_trie_t* t, const uint8_t* key, size_t len, int64_t value, int* is_new) {
    if (!t || (!key && len)) return acx_fail(ACX_E_INVAL, "acx_trie_add_word: NULL argument");
    if (is_new) *is_new = 0;
    if (len == 0) return ACX_OK;                       // src/Automaton.c:257: empty key ignored
    try {
        if (t->kind == ACX_KIND_EMPTY) t->new_node(0);  // root (src/trie.c:20-25)
        int32_t node = 0;
        for (size_t i = 0; i < len; i++) {
            int32_t c = t->child(node, key[i]);
            if (c < 0) {
                c = t->new_node(key[i]);
                t->link_child(node, c);
            }
            node = c;
        }
        Node& n = t->nodes[node];
        bool fresh = !n.eow;
        if (fresh) { n.eow = 1; t->count += 1; }        // src/trie.c:52-58
        n.value = value;                                // src/Automaton.c:268-279
        t->kind = ACX_KIND_TRIE;                        // src/trie.c:60
        if (fresh) {
            t->version += 1;                            // src/Automaton.c:283-286
            if ((int64_t)len > t->longest_word) t->longest_word = (int64_t)len;
        }
        if (is_new) *is_new = fresh ? 1 : 0;
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_add_word: out of memory");
    }
    return ACX_OK;
}

int acx_trie_add_words(acx_trie_t* t, const uint8_t* keys, const int64_t* key_off, const int64_t* values, int64_t n,
                       int value_mode, int64_t* n_new) {
    if (!t || !key_off || n < 0 || (!keys && n && key_off[n] > key_off[0])) return acx_fail(ACX_E_INVAL, "acx_trie_add_words: bad argument");
    if (!values && value_mode != 1 && value_mode != 2) return acx_fail(ACX_E_INVAL, "acx_trie_add_words: values == NULL needs value_mode 1 or 2");
    int64_t fresh = 0;
    if (n > 0 && key_off[n] > key_off[0]) {
        // room for every node these keys can add (one per byte at most): the arena does not move while they go in — a
        // million signatures are 800 MB of nodes, which doubling would copy twice over.  Address space only until used.
        // (only when they do not fit what is there, and then at least twice the capacity: many small batches must not
        //  copy the arena once per call — reserve() allocates exactly what it is asked for)
        const size_t need = t->nodes.size() + (size_t)(key_off[n] - key_off[0]) + 1, have = t->nodes.capacity();
        if (need > have) { try { t->nodes.reserve(need > 2 * have ? need : 2 * have); } catch (const std::bad_alloc&) {} }
    }
    for (int64_t i = 0; i < n; i++) {
        if (key_off[i + 1] < key_off[i]) return acx_fail(ACX_E_INVAL, "acx_trie_add_words: offsets not monotone at %lld", (long long)i);
        const size_t len = (size_t)(key_off[i + 1] - key_off[i]);
        const int64_t v = values ? values[i] : (value_mode == 1 ? t->count + 1 : (int64_t)len);
        int is_new = 0;
        const int rc = acx_trie_add_word(t, keys + key_off[i], len, v, &is_new);
        if (rc) { if (n_new) *n_new = fresh; return rc; }
        fresh += is_new;
    }
    if (n_new) *n_new = fresh;
    return ACX_OK;
}

static int32_t find_node(const acx_trie* t, const uint8_t* key, size_t len) {
    if (t->kind == ACX_KIND_EMPTY) return -1;
    int32_t node = 0;
    for (size_t i = 0; i < len; i++) {
        node = t->child(node, key[i]);
        if (node < 0) return -1;
    }
    return node;
}

int acx_trie_get(const acx_trie_t* t, const uint8_t* key, size_t len, int* found, int64_t* value) {
    if (!t || !found) return acx_fail(ACX_E_INVAL, "acx_trie_get: NULL argument");
    int32_t node = find_node(t, key, len);
    *found = (node >= 0 && t->nodes[node].eow) ? 1 : 0;
    if (*found && value) *value = t->nodes[node].value;
    return ACX_OK;
}

int acx_trie_longest_prefix(const acx_trie_t* t, const uint8_t* key, size_t len, size_t* out_len) {
    if (!t || !out_len) return acx_fail(ACX_E_INVAL, "acx_trie_longest_prefix: NULL argument");
    size_t n = 0;
    if (t->kind != ACX_KIND_EMPTY) {
        int32_t node = 0;
        for (size_t i = 0; i < len; i++) {              // trie_longest, src/trie.c:155-173
            node = t->child(node, key[i]);
            if (node < 0) break;
            n++;
        }
    }
    *out_len = n;
    return ACX_OK;
}

int acx_trie_remove_word(acx_trie_t* t, const uint8_t* key, size_t len, int* found, int64_t* value) {
    if (!t || !found) return acx_fail(ACX_E_INVAL, "acx_trie_remove_word: NULL argument");
    *found = 0;
    if (len == 0 || t->kind == ACX_KIND_EMPTY) return ACX_OK;
    // trie_remove_word, src/trie.c:66-133: remember the deepest node on the path that
    // must survive (has other children, or is itself a key), cut the tail below it.
    int32_t node = 0, last_multiway = 0;
    size_t last_multiway_index = 0;
    for (size_t i = 0; i < len; i++) {
        node = t->child(node, key[i]);
        if (node < 0) return ACX_OK;
        const Node& n = t->nodes[node];
        bool one = n.first_child >= 0 && t->nodes[n.first_child].next_sibling < 0;
        bool many = n.first_child >= 0 && !one;
        if (many || (one && n.eow)) { last_multiway = node; last_multiway_index = i + 1; }
    }
    Node& n = t->nodes[node];
    if (!n.eow) return ACX_OK;
    if (value) *value = n.value;
    if (n.first_child < 0) {
        int32_t tail = t->child(last_multiway, key[last_multiway_index]);
        t->unlink_child(last_multiway, tail);
        t->live_nodes -= (int64_t)(len - last_multiway_index);   // arena slots are not reused
    } else {
        n.eow = 0;
    }
    t->kind = ACX_KIND_TRIE;                            // src/trie.c:131
    t->version += 1;                                    // src/Automaton.c:341-342
    t->count -= 1;
    *found = 1;
    return ACX_OK;
}

int acx_trie_make_automaton(acx_trie_t* t, int* changed) {
    if (!t) return acx_fail(ACX_E_INVAL, "acx_trie_make_automaton: NULL trie");
    if (changed) *changed = 0;
    if (t->kind != ACX_KIND_TRIE) return ACX_OK;        // src/Automaton.c:574-575
    try {
        // Child lookup index for the BFS.  The fail-link chase asks "does `state` have a child
        // on `letter`?" over and over for SHALLOW states, which are exactly the wide ones (256
        // children at the top of a binary-signature trie); walking their sibling lists made
        // make_automaton 162 s for 1M signatures.  Wide nodes (>= 8 children) get a 256-entry
        // direct table, the rest keep the (short) sibling walk.
        const size_t n_arena = t->nodes.size();
        std::vector<int32_t> wide_of(n_arena, -1);
        std::vector<int32_t> wide_tbl;
        {
            std::vector<uint16_t> nchild(n_arena, 0);
            for (size_t i = 0; i < n_arena; i++)
                for (int32_t c = t->nodes[i].first_child; c >= 0; c = t->nodes[c].next_sibling) nchild[i]++;
            size_t n_wide = 0;
            for (size_t i = 0; i < n_arena; i++) if (nchild[i] >= 8) wide_of[i] = (int32_t)n_wide++;
            wide_tbl.assign(n_wide * 256, -1);
            for (size_t i = 0; i < n_arena; i++)
                if (wide_of[i] >= 0)
                    for (int32_t c = t->nodes[i].first_child; c >= 0; c = t->nodes[c].next_sibling)
                        wide_tbl[(size_t)wide_of[i] * 256 + t->nodes[c].letter] = c;
        }
        auto child_fast = [&](int32_t node, uint8_t letter) -> int32_t {
            const int32_t w = wide_of[node];
            if (w >= 0) return wide_tbl[(size_t)w * 256 + letter];
            for (int32_t c = t->nodes[node].first_child; c >= 0; c = t->nodes[c].next_sibling)
                if (t->nodes[c].letter == letter) return c;
            return -1;
        };

        // Level-synchronous BFS (src/Automaton.c:582-637 visits the same nodes in the same order with one queue).
        // Per level: count the children of every node (parallel), prefix sum (the order of the next level), then
        // place the children and compute their failure links (parallel: a link only reads links of shallower
        // levels, which are complete).
        std::vector<int32_t>& q = t->bfs;
        std::vector<int64_t>& lvl = t->level_first;
        q.clear(); lvl.clear();
        q.reserve((size_t)t->live_nodes);
        q.push_back(0);
        lvl.push_back(0); lvl.push_back(1);
        t->nodes[0].fail = -1;                          // root->fail stays NULL (src/trienode.c:19)
        std::vector<uint32_t> cnt;
        for (size_t d = 0; lvl[d] < lvl[d + 1]; d++) {
            const size_t lo = (size_t)lvl[d], hi = (size_t)lvl[d + 1];
            cnt.assign(hi - lo + 1, 0);
            parallel_range(lo, hi, [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    uint32_t k = 0;
                    for (int32_t c = t->nodes[q[i]].first_child; c >= 0; c = t->nodes[c].next_sibling) k++;
                    cnt[i - lo] = k;
                }
            });
            size_t total = 0;
            for (size_t i = 0; i < hi - lo; i++) { const uint32_t k = cnt[i]; cnt[i] = (uint32_t)total; total += k; }
            if (total >= ((size_t)1 << 31)) return acx_fail(ACX_E_UNSUPPORTED, "acx_trie_make_automaton: too many nodes");
            q.resize(hi + total);
            parallel_range(lo, hi, [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    const int32_t node = q[i];
                    size_t o = hi + cnt[i - lo];
                    for (int32_t c = t->nodes[node].first_child; c >= 0; c = t->nodes[c].next_sibling) {
                        q[o++] = c;
                        if (node == 0) { t->nodes[c].fail = 0; continue; }     // src/Automaton.c:582-596
                        const uint8_t letter = t->nodes[c].letter;
                        int32_t state = t->nodes[node].fail;
                        while (state != 0 && child_fast(state, letter) < 0) state = t->nodes[state].fail;
                        const int32_t f = child_fast(state, letter);
                        t->nodes[c].fail = f < 0 ? 0 : f;
                    }
                }
            });
            lvl.push_back((int64_t)(hi + total));
        }
        lvl.pop_back();                                 // (the last level is empty: lvl[d] .. lvl[d + 1] for d < size - 1)
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_make_automaton: out of memory");
    }
    t->kind = ACX_KIND_AHOCORASICK;
    t->version += 1;                                    // src/Automaton.c:639-640
    if (changed) *changed = 1;
    return ACX_OK;
}

int     acx_trie_kind(const acx_trie_t* t)         { return t ? t->kind : ACX_KIND_EMPTY; }
int64_t acx_trie_num_keys(const acx_trie_t* t)     { return t ? t->count : 0; }
int64_t acx_trie_num_nodes(const acx_trie_t* t)    { return t ? t->live_nodes : 0; }
int64_t acx_trie_longest_word(const acx_trie_t* t) { return t ? t->longest_word : 0; }
int64_t acx_trie_version(const acx_trie_t* t)      { return t ? t->version : 0; }

// ------------------------------------------------------------------------------------
// Flattener
// ------------------------------------------------------------------------------------
static inline size_t align_up(size_t x) { return (x + ACX_BLOB_ALIGN - 1) & ~(size_t)(ACX_BLOB_ALIGN - 1); }

uint64_t acx_fnv1a64(const uint8_t* p, size_t n) {
    // Checksum of an image (multi-GB for a million signatures): FNV-1a's xor-multiply step, but on 8-byte
    // words in 8 independent lanes (one byte at a time is a 4-cycle dependent chain per BYTE: seconds per GB),
    // the lanes and the tail folded bytewise at the end.  Only flatten and validate compute it.
    const uint64_t prime = 0x100000001b3ull;
    uint64_t lane[8];
    for (int k = 0; k < 8; k++) lane[k] = 0xcbf29ce484222325ull + (uint64_t)k;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        uint64_t w[8];
        memcpy(w, p + i, 64);
        for (int k = 0; k < 8; k++) { lane[k] ^= w[k]; lane[k] *= prime; }
    }
    uint64_t h = 0xcbf29ce484222325ull;
    for (int k = 0; k < 8; k++)
        for (int b = 0; b < 8; b++) { h ^= (lane[k] >> (8 * b)) & 0xFFu; h *= prime; }
    for (; i < n; i++) { h ^= p[i]; h *= prime; }
    return h;
}

int acx_flatten(const acx_trie_t* t, void** blob_out, size_t* nbytes_out) { return acx_flatten_ex(t, 0u, blob_out, nbytes_out); }

int acx_flatten_ex(const acx_trie_t* t, uint32_t flags, void** blob_out, size_t* nbytes_out) {
    if (!t || !blob_out || !nbytes_out) return acx_fail(ACX_E_INVAL, "acx_flatten: NULL argument");
    if ((flags & ACX_FLATTEN_TABLE_HOST) && (flags & ACX_FLATTEN_TABLE_DEVICE)) return acx_fail(ACX_E_INVAL, "acx_flatten_ex: the table is built on the host or on the device");
    if (t->kind != ACX_KIND_AHOCORASICK)
        return acx_fail(ACX_E_STATE, "acx_flatten: not an Aho-Corasick automaton yet: call make_automaton first");
    const bool timing = acx_tune_env("ACX_FLATTEN_TIMING") != nullptr;
    struct timespec lap_t0; clock_gettime(CLOCK_MONOTONIC, &lap_t0);
    auto lap = [&](const char* what) {
        if (!timing) return;
        struct timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "[acx_flatten] %s: %.3f s\n", what, (double)(t1.tv_sec - lap_t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - lap_t0.tv_nsec));
        lap_t0 = t1;
    };
    const size_t n = t->bfs.size();
    if (n >= ((size_t)1 << ACX_STATE_BITS_WIDE))
        return acx_fail(ACX_E_UNSUPPORTED, "acx_flatten: %zu states exceed the %d-bit state field of the wide image layout",
                        n, ACX_STATE_BITS_WIDE);

    // 1. byte classes: class 0 = bytes used by no key (they all lead to the root)
    bool used[256] = {false};
    {
        std::mutex mu;
        parallel_range(1, n, [&](size_t a, size_t b2) {
            bool mine[256] = {false};
            for (size_t i = a; i < b2; i++) mine[t->nodes[t->bfs[i]].letter] = true;
            std::lock_guard<std::mutex> g(mu);
            for (int b = 0; b < 256; b++) used[b] = used[b] || mine[b];
        });
    }
    uint8_t cls[256];
    unsigned n_used = 0;
    for (int b = 0; b < 256; b++) n_used += used[b];
    uint32_t K;
    if (n_used == 256) { K = 256; for (int b = 0; b < 256; b++) cls[b] = (uint8_t)b; }
    else { K = n_used + 1; unsigned k = 1; for (int b = 0; b < 256; b++) cls[b] = used[b] ? (uint8_t)k++ : 0; }

    // narrow layout whenever it fits: 24-bit states and 32-bit byte offsets into the table
    // (ACX_FLATTEN_WIDE asks for the wide layout on small automata)
    auto narrow_fits = [&](size_t rows) -> bool {
        return rows < ((size_t)1 << ACX_STATE_BITS_NARROW) && rows * (size_t)K * 4 < ((size_t)1 << 32);
    };
    const uint32_t SB = (narrow_fits(n) && !(flags & ACX_FLATTEN_WIDE)) ? ACX_STATE_BITS_NARROW : ACX_STATE_BITS_WIDE;
    const uint32_t ESC = ACX_ENTRY_CNT_ESCAPE(SB);

    // 1b. state numbering and the implicit top-of-trie (include/acx_blob.h "itop").
    //     Numbering = BFS order, except that levels 1..D+1 are re-sorted by k-gram code (any order
    //     that keeps shallower states at smaller ids works for everything below: fail(s) is
    //     always shallower), so the children of a level-D node are consecutive, in symbol order.
    const bool has_other = n_used != 256;
    const uint32_t sigma = has_other ? K - 1 : 256;
    uint32_t itop_b = 1, itop_D = 0, itop_complete = 0;
    while ((1u << itop_b) < sigma) itop_b++;
    std::vector<int32_t> order;       // position (= state id) -> arena index
    std::vector<uint32_t> acode;      // arena index -> code (valid for depth <= D + 1)
    std::vector<int32_t> adepth;      // arena index -> depth
    std::vector<uint32_t> lvl_first;  // first id of depth d, d = 0..D+2
    // one bitmap addresses levels 0..D: node (d, code) lives at bit (1 << b*d) | code
    auto itop_bm_words = [&](uint32_t D) -> size_t { return (size_t)((2ull << (itop_b * D)) / 32); };   // needs b*D >= 5
    auto itop_nd_words = [&](uint32_t D) -> size_t { return (size_t)((1ull << (itop_b * D)) / 8); };    // 4 bits per history
    auto itop_cost = [&](uint32_t D) -> size_t {      // LDS image size in words: header, ND4
        return ACX_ITOP_HDR_WORDS + itop_nd_words(D);
    };
    try {
        order = t->bfs;
        adepth.assign(t->nodes.size(), 0);
        const std::vector<int64_t>& lvl = t->level_first;           // level d = ids lvl[d] .. lvl[d + 1] (make_automaton)
        const int32_t max_depth = (int32_t)lvl.size() - 2;
        for (int32_t d = 0; d <= max_depth; d++)
            parallel_range((size_t)lvl[d], (size_t)lvl[d + 1], [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) adepth[order[i]] = d; });
        const size_t budget_words = (size_t)156 * 1024 / 4;          // of the CU's 160 KiB of LDS (+1 KiB class map)
        if (SB == ACX_STATE_BITS_NARROW && sigma <= 16 && !(flags & ACX_FLATTEN_NO_ITOP)) {
            // complete levels: every k-gram over the key alphabet is a node (the warm-up path needs
            // no probe there).  dense levels: at least 95 % of them are.  In the steady state the
            // automaton is hardly ever shallower than the last dense level, so with D <= dense + 2
            // the 2-bit depth field of ND4 (D - depth: 0, 1, 2; 3 = shallower, resolved by probing)
            // almost never escapes.
            std::vector<uint64_t> per_level((size_t)max_depth + 1, 0);
            for (int32_t d = 0; d <= max_depth; d++) per_level[d] = (uint64_t)(lvl[d + 1] - lvl[d]);
            uint64_t full = 1;
            while ((int32_t)itop_complete < max_depth && full * sigma == per_level[itop_complete + 1]) { full *= sigma; itop_complete++; }
            uint32_t dense = itop_complete;
            while ((int32_t)dense < max_depth && full <= ((uint64_t)1 << 40) && per_level[dense + 1] * 100 >= full * sigma * 95) { full *= sigma; dense++; }
            // deepest such D whose ND4 fits LDS, whose codes of level D+1 fit 24 bits and whose
            // pseudo states (n + sentinel index, see first_val) fit the state field
            while (itop_D < ACX_ITOP_MAX_LEVELS && (int32_t)itop_D < max_depth && itop_D < dense + 2 &&
                   itop_b * (itop_D + 2) <= 24 && n + ((size_t)2 << (itop_b * (itop_D + 1))) < ((size_t)1 << ACX_STATE_BITS_NARROW) &&
                   (itop_b * (itop_D + 1) < 5 || itop_cost(itop_D + 1) <= budget_words))
                itop_D++;
            if (const char* cap = acx_tune_env("ACX_ITOP_MAX_D")) { const int v = atoi(cap); if (v > 0 && (uint32_t)v < itop_D) itop_D = (uint32_t)v; }   // tuning hook
            if (itop_b * itop_D < 5) itop_D = 0;      // a trie this small does not need it (ND4 needs whole words)
        }
        if (itop_D > 0) {
            acode.assign(t->nodes.size(), 0);
            lvl_first.assign(itop_D + 3, (uint32_t)n);
            for (size_t i = n; i-- > 0;) { const int32_t d = adepth[order[i]]; if (d <= (int32_t)itop_D + 2) lvl_first[d] = (uint32_t)i; }
            for (uint32_t d = itop_D + 2; d-- > 0;) if (lvl_first[d] > lvl_first[d + 1]) lvl_first[d] = lvl_first[d + 1];
            for (uint32_t d = 1; d <= itop_D + 1; d++) {
                // parents (level d-1) already have their final code and position
                for (size_t i = lvl_first[d - 1]; i < lvl_first[d]; i++)
                    for (int32_t ch = t->nodes[order[i]].first_child; ch >= 0; ch = t->nodes[ch].next_sibling)
                        acode[ch] = (acode[order[i]] << itop_b) | (uint32_t)(cls[t->nodes[ch].letter] - (has_other ? 1 : 0));
                std::sort(order.begin() + lvl_first[d], order.begin() + lvl_first[d + 1],
                          [&](int32_t a, int32_t b2) { return acode[a] < acode[b2]; });
            }
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_flatten: out of memory");
    }

    std::vector<int32_t> id;          // arena index -> state id
    std::vector<uint32_t> out_cnt;    // per state id
    uint64_t n_out = 0;
    uint32_t max_cnt = 0;
    try {
        id.assign(t->nodes.size(), -1);
        parallel_range(0, n, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) id[order[i]] = (int32_t)i; });
        out_cnt.assign(n, 0);
        const std::vector<int64_t>& lvl = t->level_first;
        for (size_t d = 1; d + 1 < lvl.size(); d++) {     // level by level: fail(s) is shallower, hence already done
            std::vector<uint64_t> part_sum(64, 0); std::vector<uint32_t> part_max(64, 0);
            std::atomic<unsigned> slot{0};
            parallel_range((size_t)lvl[d], (size_t)lvl[d + 1], [&](size_t a, size_t b) {
                uint64_t sum = 0; uint32_t mx = 0;
                for (size_t i = a; i < b; i++) {
                    const Node& nd = t->nodes[order[i]];
                    const uint32_t c = (nd.eow ? 1u : 0u) + out_cnt[id[nd.fail]];
                    out_cnt[i] = c;
                    sum += c;
                    if (c > mx) mx = c;
                }
                const unsigned k = slot.fetch_add(1);
                part_sum[k] = sum; part_max[k] = mx;
            });
            for (unsigned k = 0; k < slot.load(); k++) { n_out += part_sum[k]; if (part_max[k] > max_cnt) max_cnt = part_max[k]; }
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_flatten: out of memory");
    }
    if (n_out >= ((uint64_t)1 << 32))
        return acx_fail(ACX_E_UNSUPPORTED, "acx_flatten: %llu output entries exceed uint32 CSR offsets",
                        (unsigned long long)n_out);

    lap("numbering + output counts");
    // 1c. position-parallel scan image (acx_ppm.cpp): its own relocatable section, appended last
    uint8_t* ppm = nullptr;
    size_t ppm_bytes = 0;
    {
        const int rcp = (flags & ACX_FLATTEN_NO_PPM) ? ACX_OK : acx_ppm_build(t, cls, K, has_other, &ppm, &ppm_bytes, (flags & ACX_FLATTEN_HOT12) != 0);
        if (rcp) return rcp;
    }
    struct PpmFree { uint8_t* p; ~PpmFree() { free(p); } } ppm_guard{ppm};

    lap("position-parallel section");
    // 2. layout
    acx_blob_header h;
    memset(&h, 0, sizeof h);
    // Where is the dense table built?  On the host (it is then part of the blob) or on the device
    // from the sparse edge lists (the blob is ~K x smaller; acx_image_upload/adopt run the build
    // kernels).  ACX_FLATTEN_TABLE_HOST / _DEVICE override; default: device once it exceeds 64 MiB.
    const size_t table_entries = n * (size_t)K;
    const uint32_t itop_cell_bytes = itop_D ? (sigma <= 4 ? 4u : 8u) : 0u;
    bool table_in_blob = table_entries * 4 < ((size_t)64 << 20);
    if (flags & ACX_FLATTEN_TABLE_HOST) table_in_blob = true;
    if (flags & ACX_FLATTEN_TABLE_DEVICE) table_in_blob = false;
    const uint32_t n_levels = (uint32_t)t->level_first.size() - 1;
    const size_t n_edges = n - 1;

    size_t off = ACX_BLOB_HEADER_BYTES;
    h.off_cls = off;        off = align_up(off + 256);
    if (table_in_blob) { h.off_table = off; off = align_up(off + table_entries * 4); }
    h.off_edge_off = off;   off = align_up(off + (n + 1) * 4);
    h.off_edge_cls = off;   off = align_up(off + n_edges + 1);
    h.off_edge_dst = off;   off = align_up(off + n_edges * 4 + 4);
    h.off_tflags = off;     off = align_up(off + n * 4);
    h.off_lvl_first = off;  off = align_up(off + ((size_t)n_levels + 1) * 4);
    h.off_fail = off;       off = align_up(off + n * 4);
    h.off_node_val = off;   off = align_up(off + n * 4);
    h.off_node_flags = off; off = align_up(off + n);
    h.off_out_off = off;    off = align_up(off + (n + 1) * 4);
    h.off_out_val = off;    off = align_up(off + (size_t)n_out * 4 + 4);
    // first_val has an entry per state and, with an itop, one per implicit node after them
    // ("pseudo state" n + x for sentinel index x): the walk reports an implicit node that has
    // exactly one output as that pseudo state with count 1 and never fetches its real entry
    const size_t itop_nx = itop_D ? (size_t)2 << (itop_b * itop_D) : 0;
    h.off_first_val = off;  off = align_up(off + (n + itop_nx) * 4);
    size_t itop_lds_words = 0, itop_entries = 0;
    if (itop_D > 0) {
        itop_lds_words = itop_cost(itop_D);
        itop_entries = (size_t)2 << (itop_b * itop_D);      // indexed by the same sentinel index as the bitmaps
        h.off_itop_lds = off;    off = align_up(off + itop_lds_words * 4);
        h.off_itop_entry = off;  off = align_up(off + itop_entries * 4);
        h.off_itop_ebits = off;  off = align_up(off + itop_bm_words(itop_D) * 4);
        h.off_itop_cells = off;  off = align_up(off + ((size_t)itop_cell_bytes << (itop_b * itop_D)));
    }
    if (ppm) { h.off_ppm = off; off = align_up(off + ppm_bytes); }
    const size_t total = off;

    uint8_t* blob = (uint8_t*)calloc(1, total);
    if (!blob) return acx_fail(ACX_E_NOMEM, "acx_flatten: cannot allocate %zu bytes for the image", total);

    memcpy(blob + h.off_cls, cls, 256);
    if (ppm) memcpy(blob + h.off_ppm, ppm, ppm_bytes);
    lap("allocate + copy the section");
    uint32_t* table   = (uint32_t*)(blob + h.off_table);
    int32_t*  fail    = (int32_t*)(blob + h.off_fail);
    int32_t*  nval    = (int32_t*)(blob + h.off_node_val);
    uint8_t*  nflags  = (uint8_t*)(blob + h.off_node_flags);
    uint32_t* out_off = (uint32_t*)(blob + h.off_out_off);
    int32_t*  out_val = (int32_t*)(blob + h.off_out_val);
    int32_t*  first_val = (int32_t*)(blob + h.off_first_val);

    // 3. per-target facts + CSR outputs (chain order: s first, then fail(s)'s list)
    std::vector<uint32_t> tflags;
    try { tflags.assign(n, 0); } catch (const std::bad_alloc&) { free(blob); return acx_fail(ACX_E_NOMEM, "acx_flatten: out of memory"); }
    {
        // offsets: a prefix sum of the counts (contiguous); then every level in parallel: a state's list is its own
        // value followed by a copy of fail(s)'s list, which belongs to a shallower level and is complete
        fail[0] = -1;
        {
            uint32_t o = 0;
            for (size_t i = 0; i < n; i++) { out_off[i] = o; o += out_cnt[i]; }
            out_off[n] = o;
        }
        const std::vector<int64_t>& lvl = t->level_first;
        for (size_t d = 1; d + 1 < lvl.size(); d++)
            parallel_range((size_t)lvl[d], (size_t)lvl[d + 1], [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    const Node& nd = t->nodes[order[i]];
                    uint32_t o = out_off[i];
                    const int32_t f = id[nd.fail];
                    fail[i] = f;
                    nval[i] = nd.eow ? (int32_t)(uint32_t)(uint64_t)nd.value : 0;   // "ii" truncation, src/AutomatonSearchIter.c:180-184
                    nflags[i] = nd.eow ? 1 : 0;
                    if (nd.eow) out_val[o++] = nval[i];
                    const uint32_t fc = out_cnt[f];
                    if (fc) memcpy(out_val + o, out_val + out_off[f], (size_t)fc * 4);
                    uint32_t fl = 0;
                    if (nd.eow) fl |= ACX_ENTRY_EOW(SB);
                    if (f != 0 && t->nodes[nd.fail].eow) fl |= ACX_ENTRY_FAILEOW(SB);   // src/AutomatonSearchIterLong.c:123
                    const uint32_t c = out_cnt[i];
                    fl |= (c >= ESC ? ESC : c) << ACX_ENTRY_CNT_SHIFT(SB);
                    tflags[i] = fl;
                    first_val[i] = c ? out_val[out_off[i]] : 0;
                }
            });
    }

    lap("outputs");
    // 4. dense fail-resolved rows: row(s) = row(fail(s)) with EDGE cleared, then own edges.
    //    row(root): every class loops to the root except its own edges.
    //    The sparse form (edge CSR + per-target bits + level boundaries) is always written; the
    //    dense rows only when the table travels inside the blob.
    {
        uint32_t* edge_off = (uint32_t*)(blob + h.off_edge_off);
        uint8_t*  edge_cls = blob + h.off_edge_cls;
        uint32_t* edge_dst = (uint32_t*)(blob + h.off_edge_dst);
        uint32_t* tfl      = (uint32_t*)(blob + h.off_tflags);
        uint32_t* lvl      = (uint32_t*)(blob + h.off_lvl_first);
        // children per state (parallel), prefix sum, then the edge lists (parallel)
        parallel_range(0, n, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                uint32_t k = 0;
                for (int32_t ch = t->nodes[order[i]].first_child; ch >= 0; ch = t->nodes[ch].next_sibling) k++;
                edge_off[i] = k;
                tfl[i] = (uint32_t)i | tflags[i];      // (every reader ORs the id in anyway; the itop walk takes the entry whole)
            }
        });
        uint32_t ne = 0;
        for (size_t i = 0; i < n; i++) { const uint32_t k = edge_off[i]; edge_off[i] = ne; ne += k; }
        edge_off[n] = ne;
        parallel_range(0, n, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                uint32_t e = edge_off[i];
                for (int32_t ch = t->nodes[order[i]].first_child; ch >= 0; ch = t->nodes[ch].next_sibling) {
                    edge_cls[e] = cls[t->nodes[ch].letter];
                    edge_dst[e] = (uint32_t)id[ch];
                    e++;
                }
            }
        });
        for (uint32_t d = 0; d <= n_levels; d++) lvl[d] = (uint32_t)t->level_first[d];      // (level_first[n_levels] = n)
    }
    if (table_in_blob) {
        for (size_t i = 0; i < n; i++) {
            uint32_t* row = table + i * K;
            if (i > 0) {
                const uint32_t* frow = table + (size_t)fail[i] * K;
                for (uint32_t c = 0; c < K; c++) row[c] = frow[c] & ~ACX_ENTRY_EDGE(SB);
            }
            for (int32_t ch = t->nodes[order[i]].first_child; ch >= 0; ch = t->nodes[ch].next_sibling) {
                const uint32_t tid = (uint32_t)id[ch];
                row[cls[t->nodes[ch].letter]] = tid | tflags[tid] | ACX_ENTRY_EDGE(SB);
            }
        }
    }

    lap("edges + table");
    // 5. implicit top-of-trie: ND4 (LDS), the existence bitmap and the packed entry of every
    //    implicit node (global)  (include/acx_blob.h)
    if (itop_D > 0) {
        uint32_t* lds = (uint32_t*)(blob + h.off_itop_lds);
        uint32_t* ient = (uint32_t*)(blob + h.off_itop_entry);
        uint32_t* E = (uint32_t*)(blob + h.off_itop_ebits);       // global: slow path only
        const size_t ndb = ACX_ITOP_HDR_WORDS;
        E[0] |= 1u << 1;                                          // the root: (d = 0, code = 0) -> bit 1
        const uint32_t complete = itop_complete < itop_D ? itop_complete : itop_D;   // levels 1..complete hold every k-gram
        for (uint32_t d = 1; d <= itop_D; d++) {
            for (uint32_t i = lvl_first[d]; i < lvl_first[d + 1]; i++) {
                const uint32_t x = (1u << (itop_b * d)) | acode[order[i]];     // sentinel index
                E[x >> 5] |= 1u << (x & 31);
                ient[x] = i | tflags[i];
            }
        }
        // ND4: for every history of D symbols, how far below D the longest k-gram node ending
        // here is (0..2; 3 = more: resolved by probing E) and its outputs (0 none, 1 exactly one, 2 more)
        const uint32_t n_hist = 1u << (itop_b * itop_D);
        for (uint32_t hh = 0; hh < n_hist; hh++) {
            uint32_t dd = itop_D, x = 1;
            for (;; dd--) {
                x = (1u << (itop_b * dd)) | (hh & ((1u << (itop_b * dd)) - 1u));
                if ((E[x >> 5] >> (x & 31)) & 1u) break;          // dd = 0 (the root) always exists
            }
            const uint32_t oc = dd == 0 ? 0 : out_cnt[ient[x] & ACX_ENTRY_STATE_MASK(SB)];
            const uint32_t q = itop_D - dd > 2 ? 3u : ((itop_D - dd) | ((oc > 2 ? 2u : oc) << 2));   // 3: shallower than D - 2
            lds[ndb + (hh >> 3)] |= q << ((hh & 7) * 4);
        }
        // pseudo states: first_val[n + x] = the first output of implicit node x
        for (uint32_t d = 1; d <= itop_D; d++)
            for (uint32_t i = lvl_first[d]; i < lvl_first[d + 1]; i++)
                first_val[n + ((1u << (itop_b * d)) | acode[order[i]])] = first_val[i];
        // cells: children of the level-D node with a given code (consecutive ids, symbol order)
        {
            uint8_t* cells = blob + h.off_itop_cells;
            for (uint32_t i = lvl_first[itop_D]; i < lvl_first[itop_D + 1]; i++) {
                uint32_t first = 0, mask = 0, outs = 0;
                for (int32_t ch = t->nodes[order[i]].first_child; ch >= 0; ch = t->nodes[ch].next_sibling) {
                    const uint32_t sym = (uint32_t)(cls[t->nodes[ch].letter] - (has_other ? 1 : 0));
                    const uint32_t cid = (uint32_t)id[ch];
                    mask |= 1u << sym;
                    if (out_cnt[cid]) outs |= 1u << sym;
                    if (first == 0 || cid < first) first = cid;
                }
                const uint32_t code = acode[order[i]];
                if (itop_cell_bytes == 4) ((uint32_t*)cells)[code] = first | (mask << 24) | (outs << 28);
                else { ((uint32_t*)cells)[2 * (size_t)code] = first; ((uint32_t*)cells)[2 * (size_t)code + 1] = mask | (outs << 16); }
            }
        }
        lds[0] = itop_b; lds[1] = itop_D; lds[2] = lvl_first[itop_D + 1]; lds[3] = itop_cell_bytes; lds[4] = (uint32_t)n;
        lds[5] = has_other ? 1u : 0u; lds[6] = (uint32_t)itop_lds_words; lds[7] = (uint32_t)((1ull << (itop_b * itop_D)) - 1);
        lds[8] = (uint32_t)ndb;
        lds[11] = itop_b * complete;                           // shifts up to this one always hit: no probe needed
        {   // shift of the shallowest level that has a node with outputs = the shortest key (>= 1 symbol)
            uint32_t hmin = 0xFFFFu;
            for (size_t i = 1; i < n && hmin == 0xFFFFu; i++) if (out_cnt[i]) hmin = (uint32_t)adepth[order[i]];
            lds[12] = hmin == 0xFFFFu ? 0xFFFFu : itop_b * hmin;
        }
        h.itop_depth = itop_D; h.itop_bits = itop_b; h.itop_lds_bytes = (uint32_t)(itop_lds_words * 4);
        h.itop_cell_bytes = itop_cell_bytes;
        h.itop_flags = ((itop_complete + 2 >= itop_D) ? ACX_ITOP_FLAG_NOESC : 0u) | ACX_ITOP_FLAG_TFLAGS_ID;
    }

    lap("itop");
    h.magic = ACX_BLOB_MAGIC;
    h.version = ACX_BLOB_VERSION;
    h.header_bytes = ACX_BLOB_HEADER_BYTES;
    h.total_bytes = total;
    h.n_states = (uint32_t)n;
    h.n_classes = K;
    h.n_keys = (uint32_t)t->count;
    h.longest_word = (uint32_t)t->longest_word;
    h.max_out_count = max_cnt;
    h.has_escape = max_cnt >= ESC ? 1 : 0;
    h.state_bits = SB;
    h.n_levels = n_levels; h.n_edges = (uint32_t)n_edges; h.table_in_blob = table_in_blob ? 1 : 0;
    h.n_out = n_out;
    h.trie_version = (uint64_t)t->version;
    h.fnv1a64 = acx_fnv1a64(blob + ACX_BLOB_HEADER_BYTES, total - ACX_BLOB_HEADER_BYTES);
    memcpy(blob, &h, sizeof h);
    lap("checksum");

    *blob_out = blob;
    *nbytes_out = total;
    return ACX_OK;
}

void acx_blob_free(void* blob) { free(blob); }

int acx_blob_check_header(const acx_blob_header* h, size_t nbytes) {
    if (nbytes < ACX_BLOB_HEADER_BYTES) return acx_fail(ACX_E_FORMAT, "image: %zu bytes is shorter than the header", nbytes);
    if (h->magic != ACX_BLOB_MAGIC) return acx_fail(ACX_E_FORMAT, "image: bad magic");
    if (h->version != ACX_BLOB_VERSION) return acx_fail(ACX_E_FORMAT, "image: version %u, this build reads %u", h->version, ACX_BLOB_VERSION);
    if (h->header_bytes != ACX_BLOB_HEADER_BYTES || h->total_bytes != nbytes)
        return acx_fail(ACX_E_FORMAT, "image: size mismatch (header says %llu, got %zu)", (unsigned long long)h->total_bytes, nbytes);
    if (h->state_bits != ACX_STATE_BITS_NARROW && h->state_bits != ACX_STATE_BITS_WIDE)
        return acx_fail(ACX_E_FORMAT, "image: unknown entry layout (state_bits = %u)", h->state_bits);
    if (h->n_states == 0 || h->n_states >= (1u << h->state_bits) || h->n_classes == 0 || h->n_classes > 256)
        return acx_fail(ACX_E_FORMAT, "image: bad n_states/n_classes");
    const uint64_t n = h->n_states, K = h->n_classes;
    const uint64_t n_codes = h->itop_depth ? 1ull << (h->itop_bits * h->itop_depth) : 0;
    if (h->itop_depth && (h->itop_bits == 0 || h->itop_bits > 4 || h->itop_bits * h->itop_depth > 22 ||
                          (h->itop_cell_bytes != 4 && h->itop_cell_bytes != 8) ||
                          h->itop_lds_bytes != (ACX_ITOP_HDR_WORDS + n_codes / 8) * 4))
        return acx_fail(ACX_E_FORMAT, "image: inconsistent implicit-top fields");
    struct { uint64_t off, len; } sec[] = {
        {h->off_cls, 256}, {h->table_in_blob ? h->off_table : (uint64_t)ACX_BLOB_ALIGN, h->table_in_blob ? n * K * 4 : 0},
        {h->off_edge_off, (n + 1) * 4}, {h->off_edge_cls, h->n_edges}, {h->off_edge_dst, (uint64_t)h->n_edges * 4},
        {h->off_tflags, n * 4}, {h->off_lvl_first, ((uint64_t)h->n_levels + 1) * 4},
        {h->off_fail, n * 4}, {h->off_node_val, n * 4},
        {h->off_node_flags, n}, {h->off_out_off, (n + 1) * 4}, {h->off_out_val, h->n_out * 4},
        {h->off_first_val, (n + 2 * n_codes) * 4},
        {h->itop_depth ? h->off_itop_lds : (uint64_t)ACX_BLOB_ALIGN, h->itop_depth ? h->itop_lds_bytes : 0},
        {h->itop_depth ? h->off_itop_entry : (uint64_t)ACX_BLOB_ALIGN, n_codes * 8},
        {h->itop_depth ? h->off_itop_ebits : (uint64_t)ACX_BLOB_ALIGN, n_codes / 4},
        {h->itop_depth ? h->off_itop_cells : (uint64_t)ACX_BLOB_ALIGN, n_codes * h->itop_cell_bytes},
    };
    for (auto& s : sec)
        if (s.off % ACX_BLOB_ALIGN || s.off < ACX_BLOB_HEADER_BYTES || s.off + s.len > nbytes)
            return acx_fail(ACX_E_FORMAT, "image: section out of bounds");
    if (h->off_ppm && (h->off_ppm % ACX_BLOB_ALIGN || h->off_ppm < ACX_BLOB_HEADER_BYTES || h->off_ppm + sizeof(acx_ppm_header) > nbytes))
        return acx_fail(ACX_E_FORMAT, "image: ppm section out of bounds");
    if (h->state_bits == ACX_STATE_BITS_NARROW && n * K * 4 >= (1ull << 32))
        return acx_fail(ACX_E_FORMAT, "image: table too large for the narrow layout's 32-bit offsets");
    return ACX_OK;
}

int acx_blob_validate(const void* blob, size_t nbytes) {
    if (!blob) return acx_fail(ACX_E_INVAL, "acx_blob_validate: NULL blob");
    acx_blob_header h;
    if (nbytes < sizeof h) return acx_fail(ACX_E_FORMAT, "image: truncated");
    memcpy(&h, blob, sizeof h);
    int rc = acx_blob_check_header(&h, nbytes);
    if (rc) return rc;
    const uint8_t* b = (const uint8_t*)blob;
    if (acx_fnv1a64(b + ACX_BLOB_HEADER_BYTES, nbytes - ACX_BLOB_HEADER_BYTES) != h.fnv1a64)
        return acx_fail(ACX_E_FORMAT, "image: checksum mismatch");
    // structural checks: every entry targets a valid state; CSR is monotone and ends at n_out
    if (h.table_in_blob) {
        const uint32_t* table = (const uint32_t*)(b + h.off_table);
        const uint64_t ne = (uint64_t)h.n_states * h.n_classes;
        for (uint64_t i = 0; i < ne; i++)
            if ((table[i] & ACX_ENTRY_STATE_MASK(h.state_bits)) >= h.n_states) return acx_fail(ACX_E_FORMAT, "image: entry %llu targets a state out of range", (unsigned long long)i);
    }
    {   // sparse form: edges target valid, deeper states; fail links point to shallower ones
        const uint32_t* eo = (const uint32_t*)(b + h.off_edge_off);
        const uint32_t* ed = (const uint32_t*)(b + h.off_edge_dst);
        const uint8_t* ec = b + h.off_edge_cls;
        const int32_t* fl = (const int32_t*)(b + h.off_fail);
        if (eo[h.n_states] != h.n_edges) return acx_fail(ACX_E_FORMAT, "image: edge CSR end does not match n_edges");
        for (uint32_t s2 = 0; s2 < h.n_states; s2++) {
            if (eo[s2] > eo[s2 + 1]) return acx_fail(ACX_E_FORMAT, "image: edge CSR not monotone at state %u", s2);
            for (uint32_t k = eo[s2]; k < eo[s2 + 1]; k++)
                if (ed[k] <= s2 || ed[k] >= h.n_states || ec[k] >= h.n_classes)
                    return acx_fail(ACX_E_FORMAT, "image: bad edge %u of state %u", k, s2);
            if (s2 > 0 && (fl[s2] < 0 || (uint32_t)fl[s2] >= s2)) return acx_fail(ACX_E_FORMAT, "image: fail link of state %u is not shallower", s2);
        }
    }
    const uint32_t* oo = (const uint32_t*)(b + h.off_out_off);
    for (uint32_t s = 0; s < h.n_states; s++)
        if (oo[s] > oo[s + 1]) return acx_fail(ACX_E_FORMAT, "image: CSR offsets not monotone at state %u", s);
    if (oo[h.n_states] != h.n_out) return acx_fail(ACX_E_FORMAT, "image: CSR end does not match n_out");
    return ACX_OK;
}

}  // extern "C"
