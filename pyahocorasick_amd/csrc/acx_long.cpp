// acx_long.cpp — the dictionary behind the position-parallel form of iter_long (host side).
//
// The reference's iter_long (automaton_search_iter_long_next, /root/reference/src/AutomatonSearchIterLong.c:89-153) is a
// serial state machine; oracle/ac_oracle.c orc_iter_long restates it.  Its state after a restart at r is not history but a
// property of the text: the longest suffix of text[r..i] that is a path of the trie.  What it reports only ever depends on
// three kinds of trie nodes on the chain  LS(i) -> fail -> fail ...  of the text at i:
//     E   the node ends a key                                            (…IterLong.c:118-121: remembered)
//     FE  it does not, its fail node is not the root and ends a key      (:122-126: reported at once, the fail node's value)
//     U   its fail node is an E or FE node: the next LONGER path of the trie that ends where such a node ends
// (tests/test_iter_long_plan_cpu.py pins the rule against the oracle: after a restart at r the record (end i, length l,
// next longer path l_up) is where the walk stops looking iff  i - l_up + 1 < r <= i - l + 1.)
// So iter_long = ACX_SCAN_ALL over the dictionary D = E + FE + U — the position-parallel kernels as they are: records
// (end, value) per haystack, position ascending, longest first within a position, so that the record in front of one with
// the same end IS its next longer path — followed by one sweep over the records of every haystack (k_long_sweep,
// acx_long.hip).  The value of a D key packs what the sweep needs:  index | length << 24 | kind << 30  (kind 0 = U,
// 1 = E, 2 = FE, 3 = an E node with no E or FE node below it in the trie; index into `real`: what iter_long reports for the
// node, first_val of the blob; a dictionary of fewer than 2^18 entries also carries, in bits 18-23, how far below the node the
// deepest E / FE node of its subtree lies).
#include "acx_internal.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

extern "C" int acx_blob_long_trie(const void* blob_v, size_t nbytes, acx_trie_t** out_trie, int32_t** real_vals, int64_t* n_out, int32_t* longest_out) {
    if (!blob_v || !out_trie || !real_vals || !n_out) return acx_fail(ACX_E_INVAL, "acx_blob_long_trie: NULL argument");
    *out_trie = nullptr; *real_vals = nullptr; *n_out = 0;
    if (longest_out) *longest_out = 0;
    const uint8_t* blob = (const uint8_t*)blob_v;
    if (nbytes < sizeof(acx_blob_header)) return acx_fail(ACX_E_FORMAT, "acx_blob_long_trie: truncated");
    acx_blob_header h;
    memcpy(&h, blob, sizeof h);
    int rc = acx_blob_check_header(&h, nbytes);
    if (rc) return rc;
    const size_t n = h.n_states;
    if (n < 2 || h.n_keys == 0) return ACX_OK;                          // (nothing to scan for: *n_out = 0)
    const uint8_t* cls = blob + h.off_cls;
    const int32_t* fail = (const int32_t*)(blob + h.off_fail);
    const uint8_t* flags = blob + h.off_node_flags;
    const int32_t* first_val = (const int32_t*)(blob + h.off_first_val);
    const uint32_t* edge_off = (const uint32_t*)(blob + h.off_edge_off);
    const uint8_t* edge_cls = blob + h.off_edge_cls;
    const uint32_t* edge_dst = (const uint32_t*)(blob + h.off_edge_dst);
    if (!h.off_edge_off || !h.off_edge_cls || !h.off_edge_dst || !h.off_fail || !h.off_node_flags || !h.off_first_val)
        return acx_fail(ACX_E_FORMAT, "acx_blob_long_trie: the blob lacks the sparse form");
    uint8_t byte_of[256];
    memset(byte_of, 0, sizeof byte_of);
    for (int b = 0; b < 256; b++) if (cls[b]) byte_of[cls[b]] = (uint8_t)b;      // (every byte of a key has a class of its own)
    std::vector<uint32_t> parent(n, 0);
    std::vector<uint8_t> pbyte(n, 0), kind(n, 0);
    std::vector<uint32_t> depth(n, 0);
    for (size_t s = 0; s < n; s++) {                                    // BFS numbering: a parent comes before its children
        for (uint32_t e = edge_off[s]; e < edge_off[s + 1]; e++) {
            const uint32_t d = edge_dst[e];
            if (d >= n || d <= s) return acx_fail(ACX_E_FORMAT, "acx_blob_long_trie: edges are not in BFS order");
            parent[d] = (uint32_t)s; pbyte[d] = byte_of[edge_cls[e]];
            depth[d] = depth[s] + 1u;
        }
    }
    // kinds: 1 = E, 2 = FE, then U = 4 for the nodes whose fail node is an event node (an FE node is its key's U as well)
    for (size_t s = 1; s < n; s++) {
        if (flags[s] & 1u) kind[s] = 1;
        else { const int32_t f = fail[s]; if (f > 0 && (flags[f] & 1u)) kind[s] = 2; }
    }
    size_t nd = 0;
    for (size_t s = 1; s < n; s++) {
        const int32_t f = fail[s];
        if (f > 0 && (kind[f] & 3u)) kind[s] |= 4u;
        if (kind[s]) nd++;
    }
    // below[s]: how far below node s the deepest E or FE node of its subtree lies (0: none).  The walk that remembers an E node
    // (…IterLong.c:118-121) goes on down the trie; what it remembers can only be replaced within that many letters.  An E node with
    // nothing below (kind 3 in the packed value) is certain to be reported the moment it is reached; for the others the sweep looks
    // ahead along the path for `below` letters (6 bits of the value when the dictionary has fewer than 2^18 entries, else for
    // longest - 1 letters).
    std::vector<uint8_t> below(n, 0);
    for (size_t s = n; s-- > 1;) {                                      // (BFS numbering: children behind their parents)
        if (!(kind[s] & 3u) && !below[s]) continue;
        const uint32_t c = 1u + below[s];
        const uint32_t par = parent[s];
        if (c > below[par]) below[par] = (uint8_t)(c > 63u ? 63u : c);
    }
    for (size_t s = 1; s < n; s++) if (kind[s] && depth[s] > 63u) return ACX_OK;     // (deeper than the 6 bits of the length field: the serial walk stays)
    if (nd == 0 || nd >= ((size_t)1 << 24)) return ACX_OK;
    std::vector<uint8_t> keys;
    std::vector<int64_t> key_off, values;
    int32_t* real = (int32_t*)malloc(nd * sizeof(int32_t));
    if (!real) return acx_fail(ACX_E_NOMEM, "acx_blob_long_trie: out of memory");
    uint32_t longest = 0;
    try {
        key_off.reserve(nd + 1); values.reserve(nd);
        key_off.push_back(0);
        size_t idx = 0;
        uint8_t tmp[64];
        for (size_t s = 1; s < n; s++) {
            if (!kind[s]) continue;
            const uint32_t len = depth[s];
            if (len == 0 || len > 63) { free(real); return ACX_OK; }
            size_t x = s;
            for (uint32_t i = len; i-- > 0;) { tmp[i] = pbyte[x]; x = parent[x]; }
            keys.insert(keys.end(), tmp, tmp + len);
            key_off.push_back((int64_t)keys.size());
            uint32_t k = kind[s] & 3u;
            if (k == 1u && !below[s]) k = 3u;
            const uint32_t hb = nd < ((size_t)1 << ACX_LONG_SMALL_BITS) ? (uint32_t)below[s] << ACX_LONG_SMALL_BITS : 0u;
            values.push_back((int64_t)(int32_t)((uint32_t)idx | hb | (len << 24) | (k << 30)));
            real[idx] = first_val[s];                                   // (E: its own value; FE: its fail node's — the first output)
            if (len > longest) longest = len;
            idx++;
        }
    } catch (const std::bad_alloc&) { free(real); return acx_fail(ACX_E_NOMEM, "acx_blob_long_trie: out of memory"); }
    acx_trie_t* t = nullptr;
    if ((rc = acx_trie_new(&t))) { free(real); return rc; }
    int64_t n_new = 0;
    rc = acx_trie_add_words(t, keys.data(), key_off.data(), values.data(), (int64_t)nd, 0, &n_new);
    int changed = 0;
    if (!rc) rc = acx_trie_make_automaton(t, &changed);
    if (rc || n_new != (int64_t)nd) { acx_trie_free(t); free(real); return rc ? rc : acx_fail(ACX_E_FORMAT, "acx_blob_long_trie: duplicate nodes"); }
    *out_trie = t; *real_vals = real; *n_out = (int64_t)nd;
    if (longest_out) *longest_out = (int32_t)longest;
    return ACX_OK;
}
