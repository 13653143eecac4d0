// acx_trie_impl.h — the arena trie behind acx_trie_t (private to libacx: acx_trie.cpp builds,
// walks and flattens it, acx_persist.cpp reads and writes the reference's persistence formats).
#ifndef ACX_TRIE_IMPL_H_INCLUDED
#define ACX_TRIE_IMPL_H_INCLUDED

#include "acx_internal.h"

#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

struct Node {
    int64_t value;
    int32_t first_child;
    int32_t next_sibling;
    int32_t fail;
    uint8_t letter;
    uint8_t eow;
    uint16_t wide;      // 1: the node has many children and a direct index (acx_trie::wide_index)
};
static_assert(sizeof(Node) == 24, "Node layout");

struct acx_trie {
    std::vector<Node> nodes;        // nodes[0] = root once kind != EMPTY
    int32_t root_child[256];        // direct index for the root's children (-1 = none)
    std::vector<int32_t> bfs;       // BFS order recorded by make_automaton (root first)
    std::vector<int64_t> level_first;   // ... and where its levels start: level d = bfs[level_first[d] .. level_first[d + 1])
    int kind = ACX_KIND_EMPTY;
    int64_t count = 0;
    int64_t longest_word = 0;
    int64_t version = 0;
    int64_t live_nodes = 0;
    // Sibling lists keep the insertion order (what keys() / items() iterate in, like the reference's child arrays), but a
    // node near the root of a signature trie has up to 256 children: walking its list for every key made add_word
    // quadratic in practice (a million signatures: 3.4 s).  A node that gets its 17th child also gets a direct index.
    struct Wide { int32_t child[256]; int32_t last; };
    std::unordered_map<int32_t, Wide> wide_index;

    acx_trie() { for (auto& c : root_child) c = -1; }

    int32_t child(int32_t node, uint8_t letter) const {
        if (node == 0) return root_child[letter];
        if (nodes[node].wide) return wide_index.find(node)->second.child[letter];
        for (int32_t c = nodes[node].first_child; c >= 0; c = nodes[c].next_sibling)
            if (nodes[c].letter == letter) return c;
        return -1;
    }

    int32_t new_node(uint8_t letter) {
        Node n;
        n.value = 0; n.first_child = -1; n.next_sibling = -1; n.fail = -1;
        n.letter = letter; n.eow = 0; n.wide = 0;
        nodes.push_back(n);
        live_nodes++;
        return (int32_t)nodes.size() - 1;
    }

    // append `c` at the end of `parent`'s sibling list (insertion order, like
    // trienode_set_next, src/trienode.c:124-147)
    void link_child(int32_t parent, int32_t c) {
        if (parent != 0 && nodes[parent].wide) {
            Wide& w = wide_index.find(parent)->second;
            if (w.last >= 0) nodes[w.last].next_sibling = c; else nodes[parent].first_child = c;
            w.last = c; w.child[nodes[c].letter] = c;
            return;
        }
        int32_t* slot = &nodes[parent].first_child;
        int n = 0;
        while (*slot >= 0) { slot = &nodes[*slot].next_sibling; n++; }
        *slot = c;
        if (parent == 0) root_child[nodes[c].letter] = c;
        else if (n >= 16) {                                           // the 17th child: index the node
            Wide w;
            for (auto& x : w.child) x = -1;
            w.last = -1;
            for (int32_t k = nodes[parent].first_child; k >= 0; k = nodes[k].next_sibling) { w.child[nodes[k].letter] = k; w.last = k; }
            wide_index.emplace(parent, w);
            nodes[parent].wide = 1;
        }
    }

    void unlink_child(int32_t parent, int32_t c) {
        int32_t* slot = &nodes[parent].first_child;
        int32_t prev = -1;
        while (*slot >= 0 && *slot != c) { prev = *slot; slot = &nodes[*slot].next_sibling; }
        if (*slot == c) *slot = nodes[c].next_sibling;
        if (parent == 0) root_child[nodes[c].letter] = -1;
        else if (nodes[parent].wide) {
            Wide& w = wide_index.find(parent)->second;
            w.child[nodes[c].letter] = -1;
            if (w.last == c) w.last = prev;
        }
    }
};



// Host loops over tens of millions of trie nodes are bound by cache misses (the arena is in insertion order, the
// passes run in BFS order): they are cut into contiguous ranges for a handful of threads.  f(a, b) gets disjoint
// sub-ranges of [lo, hi); small ranges run inline.  ACX_HOST_THREADS overrides the thread count (1: serial).
static inline unsigned acx_host_threads() {
    static const unsigned n = [] {
        const char* e = getenv("ACX_HOST_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        if (v < 1) v = 1;
        return v > 32 ? 32u : v;
    }();
    return n;
}
template <typename F>
static inline void parallel_range(size_t lo, size_t hi, F&& f) {
    const size_t n = hi > lo ? hi - lo : 0, grain = 16384;
    size_t T = acx_host_threads();
    if (T > n / grain) T = n / grain;
    if (T <= 1) { if (n) f(lo, hi); return; }
    const size_t step = (n + T - 1) / T;
    std::vector<std::thread> th;
    th.reserve(T - 1);
    // an exception in a worker (std::bad_alloc) would end the process: it is carried to the calling thread instead
    std::exception_ptr err;
    std::mutex err_mu;
    auto guarded = [&](size_t a, size_t b) {
        try { f(a, b); }
        catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
    };
    for (size_t k = 1; k < T; k++) {
        const size_t a = lo + k * step, b = a + step < hi ? a + step : hi;
        if (a < b) th.emplace_back([&guarded, a, b] { guarded(a, b); });
    }
    guarded(lo, lo + step < hi ? lo + step : hi);
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}


// one whole letter below a node: the trie stores a multi-byte letter (UTF-8, continued to 6-byte
// forms for KEY_SEQUENCE integers) as a chain of byte nodes.  acx_items.cpp.
struct AcxLetterChild {
    int32_t node;        // arena index of the letter's last byte node
    uint8_t len;         // bytes of the letter
    uint8_t b[6];
};
// children of `parent` one letter down, in the order their letters were first added (= arena order)
void acx_letter_children(const acx_trie* t, int32_t parent, bool multibyte, std::vector<AcxLetterChild>& out);
uint32_t acx_letter_value(const uint8_t* b, int len);     // decode one stored letter

// position-parallel scan image (acx_ppm.cpp; layout: include/acx_blob.h).  *out = nullptr (and ACX_OK)
// when the automaton gets none.  malloc'd.
int acx_ppm_build(const acx_trie* t, const uint8_t* cls, uint32_t n_classes, bool has_other, uint8_t** out, size_t* nbytes, bool hot12 = false);

#endif
