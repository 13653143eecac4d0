// acx_items.cpp — key enumeration with prefix / wildcard patterns: what Automaton.keys(),
// values(), items() and __iter__ of the reference iterate over (the dict-like
// methods).  CPU only.
//
// Same traversal as automaton_items_iter_next (src/AutomatonItemsIter.c:124-209): a LIFO stack
// of (node, depth); popping a node pushes all its children when the pattern is exhausted or its
// letter at this depth is the wildcard, else only the child on the pattern's letter; children are
// pushed in array (= insertion) order, hence visited last child first.  `how`
// (src/Automaton.h:43-47):
//   MATCH_EXACT_LENGTH (0): keys as long as the pattern;  MATCH_AT_MOST_PREFIX (1): not longer;
//   MATCH_AT_LEAST_PREFIX (2): not shorter (the default without a wildcard).
// letters_utf8 = 0: one letter = one byte (bytes build).
// letters_utf8 = 1: one letter = one UTF-8 sequence (the str build keeps UTF-8 in the byte trie).
//   The "children" of a node are then the nodes one whole character below it, ordered by creation
//   (arena index of the character's last byte node = when that letter was first added under this
//   parent), which is the reference's child-array order; depth and the pattern count characters.
// The result is materialised (keys back to back + offsets + values); the host iterators hand it
// out one by one and re-check the trie version like the reference's iterator does.
#include "acx_trie_impl.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

inline bool char_start(uint8_t b) { return (b & 0xC0) != 0x80; }
// (5- and 6-byte forms: the str build's hosts also store KEY_SEQUENCE letters up to 31 bits this way)
inline int utf8_len(uint8_t lead) { return lead < 0x80 ? 1 : lead >= 0xFC ? 6 : lead >= 0xF8 ? 5 : lead >= 0xF0 ? 4 : lead >= 0xE0 ? 3 : 2; }

struct Item {            // a node one whole letter below its parent item
    int32_t node;
    int32_t depth;       // letters
    int32_t nbytes;      // key bytes up to and including this letter
    uint8_t len;         // bytes of this letter (0 for the root)
    uint8_t b[6];
};

// the nodes one letter below `parent` (for UTF-8: 1-4 byte levels down), in creation order
void letter_children(const acx_trie* t, const Item& parent, bool utf8, std::vector<Item>& out) {
    out.clear();
    struct Walk { int32_t node; int left; Item it; };
    std::vector<Walk> todo;
    for (int32_t c = t->nodes[parent.node].first_child; c >= 0; c = t->nodes[c].next_sibling) {
        Item k;
        k.node = c; k.depth = parent.depth + 1; k.len = 1; memset(k.b, 0, sizeof k.b); k.b[0] = t->nodes[c].letter;
        const int len = utf8 ? utf8_len(t->nodes[c].letter) : 1;
        k.nbytes = parent.nbytes + len;
        if (len == 1) out.push_back(k);
        else todo.push_back({c, len - 1, k});
    }
    while (!todo.empty()) {
        const Walk w = todo.back();
        todo.pop_back();
        for (int32_t c = t->nodes[w.node].first_child; c >= 0; c = t->nodes[c].next_sibling) {
            Item k = w.it;
            k.node = c; k.b[k.len] = t->nodes[c].letter; k.len++;
            if (w.left == 1) out.push_back(k);
            else todo.push_back({c, w.left - 1, k});
        }
    }
    if (utf8) std::sort(out.begin(), out.end(), [](const Item& x, const Item& y) { return x.node < y.node; });
}

}  // namespace

void acx_letter_children(const acx_trie* t, int32_t parent, bool multibyte, std::vector<AcxLetterChild>& out) {
    Item p;
    p.node = parent; p.depth = 0; p.nbytes = 0; p.len = 0; memset(p.b, 0, sizeof p.b);
    std::vector<Item> kids;
    letter_children(t, p, multibyte, kids);
    out.clear();
    for (const Item& k : kids) {
        AcxLetterChild c;
        c.node = k.node; c.len = k.len; memcpy(c.b, k.b, sizeof c.b);
        out.push_back(c);
    }
}

uint32_t acx_letter_value(const uint8_t* b, int len) {
    if (len <= 1) return b[0];
    uint32_t x = b[0] & (0xFFu >> (len + 1));
    for (int k = 1; k < len; k++) x = (x << 6) | (b[k] & 0x3Fu);
    return x;
}

extern "C" int acx_trie_items(const acx_trie_t* t, const uint8_t* pattern, size_t plen, const uint8_t* wildcard, size_t wlen,
                              int how, int letters_utf8, uint8_t** keys, int64_t** key_off, int64_t** values, int64_t* n) {
    if (!t || !keys || !key_off || !values || !n || (plen && !pattern) || (wlen && !wildcard) || wlen > 6)
        return acx_fail(ACX_E_INVAL, "acx_trie_items: bad argument");
    if (how < 0 || how > 2) return acx_fail(ACX_E_INVAL, "acx_trie_items: bad match type %d", how);
    const bool utf8 = letters_utf8 != 0;
    std::vector<uint8_t> kbuf;
    std::vector<int64_t> koff, vals;
    try {
        std::vector<size_t> poff;                                    // byte offset of each letter of the pattern (+ end)
        for (size_t i = 0; i < plen; i++) if (!utf8 || char_start(pattern[i])) poff.push_back(i);
        const size_t pletters = poff.size();
        poff.push_back(plen);
        koff.push_back(0);
        if (t->kind != ACX_KIND_EMPTY && !t->nodes.empty()) {
            std::vector<Item> stack, kids;
            std::vector<uint8_t> path((size_t)t->longest_word + 8, 0);    // key bytes of the current branch
            Item root;
            root.node = 0; root.depth = 0; root.nbytes = 0; root.len = 0; memset(root.b, 0, sizeof root.b);
            stack.push_back(root);
            while (!stack.empty()) {
                const Item it = stack.back();
                stack.pop_back();
                const size_t depth = (size_t)it.depth;
                if (how != 2 && depth > pletters) continue;
                const bool output = how == 0 ? depth == pletters : (how == 1 ? depth <= pletters : depth >= pletters);
                bool expand_all = depth >= pletters;
                if (!expand_all && wlen) {
                    const size_t l0 = poff[depth], l1 = poff[depth + 1];
                    expand_all = l1 - l0 == wlen && memcmp(pattern + l0, wildcard, wlen) == 0;
                }
                if (expand_all) {
                    letter_children(t, it, utf8, kids);
                    for (const Item& k : kids) stack.push_back(k);                  // popped in reverse: last child first
                } else {
                    const size_t l0 = poff[depth], l1 = poff[depth + 1];
                    int32_t c = it.node;
                    for (size_t i = l0; i < l1 && c >= 0; i++) c = t->child(c, pattern[i]);
                    if (c >= 0 && l1 - l0 <= 6) {
                        Item k;
                        k.node = c; k.depth = it.depth + 1; k.nbytes = it.nbytes + (int32_t)(l1 - l0); k.len = (uint8_t)(l1 - l0);
                        memset(k.b, 0, sizeof k.b);
                        memcpy(k.b, pattern + l0, l1 - l0);
                        stack.push_back(k);
                    }
                }
                if ((size_t)it.nbytes + 1 > path.size()) path.resize((size_t)it.nbytes + 64, 0);
                if (it.len) memcpy(path.data() + it.nbytes - it.len, it.b, it.len);   // ancestors' bytes are still in place
                const Node& nd = t->nodes[it.node];
                if (output && nd.eow) {
                    kbuf.insert(kbuf.end(), path.begin(), path.begin() + it.nbytes);
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
// (32-byte TrieNode + one 8-byte slot per edge: trienode_get_size, src/trie.c:228-231).
// (For a str build the counts are those of the UTF-8 byte trie.)
extern "C" int acx_trie_stats(const acx_trie_t* t, int64_t* nodes, int64_t* words, int64_t* longest, int64_t* links,
                              int64_t* sizeof_node, int64_t* total_size) {
    if (!t) return acx_fail(ACX_E_INVAL, "acx_trie_stats: NULL trie");
    int64_t nn = 0, nw = 0, nl = 0, deepest = 0;
    if (t->kind != ACX_KIND_EMPTY && !t->nodes.empty()) {
        try {
            struct It { int32_t node; int32_t depth; };
            std::vector<It> stack;
            stack.push_back({0, 0});
            while (!stack.empty()) {
                const It it = stack.back();
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
