// acx_items.cpp — key enumeration with prefix / wildcard patterns: what Automaton.keys(),
// values(), items() and __iter__ of the reference iterate over (the dict-like
// methods).  CPU only.
//
// Same traversal as automaton_items_iter_next (src/AutomatonItemsIter.c:124-209): a LIFO stack
// of (node, depth); popping a node pushes all its children when the pattern is exhausted or its
// letter at this depth is the wildcard, else only the child on the pattern's letter; children are
// pushed in array order, hence visited last child first.  `how` (src/Automaton.h:43-47):
//   MATCH_EXACT_LENGTH (0): keys as long as the pattern;  MATCH_AT_MOST_PREFIX (1): not longer;
//   MATCH_AT_LEAST_PREFIX (2): not shorter (the default without a wildcard).
// The result is materialised (keys back to back + offsets + values); the host iterators hand it
// out one by one and re-check the trie version like the reference's iterator does.
#include "acx_trie_impl.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

extern "C" int acx_trie_items(const acx_trie_t* t, const uint8_t* pattern, size_t plen, int use_wildcard, uint8_t wildcard,
                              int how, uint8_t** keys, int64_t** key_off, int64_t** values, int64_t* n) {
    if (!t || !keys || !key_off || !values || !n || (plen && !pattern)) return acx_fail(ACX_E_INVAL, "acx_trie_items: bad argument");
    if (how < 0 || how > 2) return acx_fail(ACX_E_INVAL, "acx_trie_items: bad match type %d", how);
    struct Item { int32_t node; int32_t depth; };
    std::vector<uint8_t> kbuf;
    std::vector<int64_t> koff, vals;
    try {
        koff.push_back(0);
        if (t->kind != ACX_KIND_EMPTY && !t->nodes.empty()) {
            std::vector<Item> stack;
            std::vector<uint8_t> path((size_t)t->longest_word + 2, 0);      // path[d] = letter leading to depth d
            std::vector<int32_t> kids;
            stack.push_back({0, 0});
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                const size_t depth = (size_t)it.depth;
                if (how != 2 && depth > plen) continue;
                const bool output = how == 0 ? depth == plen : (how == 1 ? depth <= plen : depth >= plen);
                const Node& nd = t->nodes[it.node];
                if (depth >= plen || (use_wildcard && pattern[depth] == wildcard)) {
                    kids.clear();
                    for (int32_t c = nd.first_child; c >= 0; c = t->nodes[c].next_sibling) kids.push_back(c);
                    for (int32_t c : kids) stack.push_back({c, it.depth + 1});     // popped in reverse: last child first
                } else {
                    const int32_t c = t->child(it.node, pattern[depth]);
                    if (c >= 0) stack.push_back({c, it.depth + 1});
                }
                if (depth >= path.size()) path.resize(depth + 1, 0);
                path[depth] = nd.letter;
                if (output && nd.eow) {
                    kbuf.insert(kbuf.end(), path.begin() + 1, path.begin() + 1 + (ptrdiff_t)depth);
                    koff.push_back((int64_t)kbuf.size());
                    vals.push_back(nd.value);
                }
            }
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_items: out of memory");
    }
    uint8_t* kb = (uint8_t*)malloc(kbuf.size() + 1);
    int64_t* ko = (int64_t*)malloc(koff.size() * sizeof(int64_t));
    int64_t* vv = (int64_t*)malloc((vals.size() + 1) * sizeof(int64_t));
    if (!kb || !ko || !vv) { free(kb); free(ko); free(vv); return acx_fail(ACX_E_NOMEM, "acx_trie_items: out of memory"); }
    if (!kbuf.empty()) memcpy(kb, kbuf.data(), kbuf.size());
    memcpy(ko, koff.data(), koff.size() * sizeof(int64_t));
    if (!vals.empty()) memcpy(vv, vals.data(), vals.size() * sizeof(int64_t));
    *keys = kb; *key_off = ko; *values = vv; *n = (int64_t)vals.size();
    return ACX_OK;
}

// what Automaton.get_stats() reports (src/Automaton.c:1044-1096): node / key / edge counts, the
// longest path, and the memory the REFERENCE's pointer trie would take for this automaton
// (32-byte TrieNode + one 8-byte slot per edge: trienode_get_size, src/trie.c:228-231)
extern "C" int acx_trie_stats(const acx_trie_t* t, int64_t* nodes, int64_t* words, int64_t* longest, int64_t* links,
                              int64_t* sizeof_node, int64_t* total_size) {
    if (!t) return acx_fail(ACX_E_INVAL, "acx_trie_stats: NULL trie");
    int64_t nn = 0, nw = 0, nl = 0, deepest = 0;
    if (t->kind != ACX_KIND_EMPTY && !t->nodes.empty()) {
        try {
            struct Item { int32_t node; int32_t depth; };
            std::vector<Item> stack;
            stack.push_back({0, 0});
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                nn++;
                nw += t->nodes[it.node].eow;
                if (it.depth > deepest) deepest = it.depth;
                for (int32_t c = t->nodes[it.node].first_child; c >= 0; c = t->nodes[c].next_sibling) { nl++; stack.push_back({c, it.depth + 1}); }
            }
        } catch (const std::bad_alloc&) {
            return acx_fail(ACX_E_NOMEM, "acx_trie_stats: out of memory");
        }
    }
    if (nodes) *nodes = nn;
    if (words) *words = nw;
    if (longest) *longest = deepest;
    if (links) *links = nl;
    if (sizeof_node) *sizeof_node = 32;
    if (total_size) *total_size = nn * 32 + nl * 8;
    return ACX_OK;
}
