// acx_persist.cpp — the reference's two persistence formats <-> the arena trie.
//
// CPU only.  Both formats are dumps of the reference's pointer trie, one record per node (the
// reference writes them in trie_traverse order: pre-order, children in array order,
// src/trie.c:197-225; its loaders accept any order):
//
//   record  = 24 bytes  { u64 output; u64 fail; u32 n; u8 eow; 3 bytes of padding (garbage) }
//             = PICKLE_TRIENODE_SIZE, src/pickle/pickle.h:7 (TrieNode without its `next` pointer,
//               src/trienode.h:27-41; LP64; the bytes build stores letters as uint16:
//               src/common.h:63-67, hence the 32-bit n)
//           + n x 10 bytes { u16 letter; u64 child }      (packed Pair, src/trienode.h:19-24; a key byte
//             >= 0x80 is stored sign-extended: `char` widened to uint16)
//
//   pickle  (Automaton.__reduce__, src/Automaton_pickle.c:128-285): a list of byte chunks, each
//           { i64 nodes in this chunk; records }.  Links are 1-based node numbers in dump order
//           (0 = NULL, the root is node 1); `output` is the integer value, or 0 when the values
//           travel as a separate Python list (STORE_ANY: one item per eow node, in dump order).
//   save    (Automaton.save, src/custompickle/save/automaton_save.c:36-138): one file
//           { magic "pyahocorasick002"; i32 kind, store, key_type; u64 words; i32 longest }   48 bytes
//           then per node { u64 address of the node; record; [STORE_ANY and eow: `output` bytes of
//           serialized value] } with links = raw addresses, then { u64 nodes; magic }.
//
// Loading rebuilds the arena trie node for node (same child order, so keys() order and every
// search result are the reference's); fail links are NOT taken from the file — the host calls
// make_automaton again when the dump was an automaton.  Dumps written here carry the fail links
// the reference's loader expects, and every dump written here is a valid dump of the corresponding
// reference build: bytes build (uint16 letters) for byte keys and bytes-flavour KEY_SEQUENCE, unicode
// build (uint32 letters) for the str flavour — multi-byte letters are decoded on the way out.
#include "acx_trie_impl.h"

#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

namespace {

constexpr size_t REC = 24, PAIR = 10, PAIR_UCS4 = 12, SAVE_HEADER = 48, SAVE_FOOTER = 24;
const char MAGIC[16] = {'p', 'y', 'a', 'h', 'o', 'c', 'o', 'r', 'a', 's', 'i', 'c', 'k', '0', '0', '2'};

inline uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
inline uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t rd16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline void wr64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
inline void wr32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
inline void wr16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }

struct RawNode {            // one parsed record; links still in the file's own naming
    uint64_t output;
    uint32_t n;
    uint8_t eow;
    const uint8_t* pairs;   // n x PAIR bytes
};

// Build the arena trie from parsed records.  child_index(k, j) = index (0-based, dump order) of
// the j-th child of node k, or -1 if the link is invalid.  Node 0 is the root.
template <class ChildIndex>
int build_trie(const std::vector<RawNode>& raw, bool values_by_position, int64_t longest_word, ChildIndex child_index,
               acx_trie** out, int64_t* n_eow_out) {
    const size_t n = raw.size();
    if (n == 0 || n >= ((size_t)1 << 31)) return acx_fail(ACX_E_FORMAT, "reference dump: %zu nodes", n);
    acx_trie* t = new (std::nothrow) acx_trie();
    if (!t) return acx_fail(ACX_E_NOMEM, "reference dump: out of memory");
    int rc = ACX_OK;
    try {
        t->nodes.resize(n);
        std::vector<int32_t> depth(n, -1);
        std::vector<uint8_t> linked(n, 0);
        for (size_t k = 0; k < n; k++) {
            Node& nd = t->nodes[k];
            nd.value = 0; nd.first_child = -1; nd.next_sibling = -1; nd.fail = -1; nd.letter = 0; nd.eow = raw[k].eow ? 1 : 0; nd.wide = 0;
        }
        depth[0] = 0;
        int64_t eow_seen = 0, longest = 0;
        // dump order is pre-order, so a parent always precedes its children: one forward pass
        for (size_t k = 0; k < n && rc == ACX_OK; k++) {
            if (depth[k] < 0) { rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu is not reachable from the root", k); break; }
            Node& nd = t->nodes[k];
            if (nd.eow) {
                nd.value = values_by_position ? eow_seen : (int64_t)raw[k].output;
                eow_seen++;
                if (depth[k] > longest) longest = depth[k];
                if (k == 0) { rc = acx_fail(ACX_E_FORMAT, "reference dump: the root is marked as a key"); break; }
            }
            int32_t last = -1;
            for (uint32_t j = 0; j < raw[k].n; j++) {
                const int64_t c = child_index(k, j);
                if (c <= (int64_t)k || c >= (int64_t)n || linked[c]) {
                    rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu has a malformed link #%u", k, j);
                    break;
                }
                // the bytes build widens `char` key bytes to uint16 letters, so bytes >= 0x80 arrive
                // sign-extended (0xFF80..0xFFFF)
                const uint16_t letter = rd16(raw[k].pairs + (size_t)j * PAIR);
                // ... or zero-extended (0x0080..0x00FF) where the reference was built with an unsigned `char` (aarch64)
                if (letter > 0xFF && letter < 0xFF80) { rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu has letter %u: not a bytes-build dump", k, letter); break; }
                for (int32_t sib = nd.first_child; sib >= 0 && rc == ACX_OK; sib = t->nodes[sib].next_sibling)
                    if (t->nodes[sib].letter == (uint8_t)letter) rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu has two edges with letter %u", k, (unsigned)(uint8_t)letter);
                if (rc != ACX_OK) break;
                linked[c] = 1;
                depth[c] = depth[k] + 1;
                t->nodes[c].letter = (uint8_t)letter;
                if (last < 0) nd.first_child = (int32_t)c; else t->nodes[last].next_sibling = (int32_t)c;
                last = (int32_t)c;
                if (k == 0) {
                    if (t->root_child[t->nodes[c].letter] >= 0) { rc = acx_fail(ACX_E_FORMAT, "reference dump: duplicate root edge"); break; }
                    t->root_child[t->nodes[c].letter] = (int32_t)c;
                }
            }
        }
        if (rc == ACX_OK) {
            t->kind = ACX_KIND_TRIE;
            t->count = eow_seen;
            // the reference never lowers longest_word (remove_word leaves it): keep the dump's value,
            // but never below the truth (the chunk halo of the scan is longest_word - 1)
            t->longest_word = longest_word > longest ? longest_word : longest;
            t->live_nodes = (int64_t)n;
            t->version = 1;
            if (n_eow_out) *n_eow_out = eow_seen;
        }
    } catch (const std::bad_alloc&) {
        rc = acx_fail(ACX_E_NOMEM, "reference dump: out of memory");
    }
    if (rc != ACX_OK) { delete t; return rc; }
    *out = t;
    return ACX_OK;
}

// The unicode build's dumps: same records, but a pair is { u32 letter (code point); u64 child } = 12
// bytes (src/common.h:50-56: TRIE_LETTER_TYPE uint32_t).  The byte trie here holds UTF-8, so the
// code-point trie is not copied node for node: its keys are re-inserted, in dump (pre-order) order,
// which also recreates the reference's child order letter by letter (a letter's node is created
// when the first key through it is added).
inline size_t utf8_encode(uint32_t cp, uint8_t* o) {
    if (cp < 0x80) { o[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { o[0] = (uint8_t)(0xC0 | (cp >> 6)); o[1] = (uint8_t)(0x80 | (cp & 0x3F)); return 2; }
    if (cp < 0x10000) { o[0] = (uint8_t)(0xE0 | (cp >> 12)); o[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); o[2] = (uint8_t)(0x80 | (cp & 0x3F)); return 3; }
    o[0] = (uint8_t)(0xF0 | (cp >> 18)); o[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F)); o[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F)); o[3] = (uint8_t)(0x80 | (cp & 0x3F));
    return 4;
}

// any letter up to 31 bits as one self-synchronising sequence: UTF-8 continued with its original
// 5- and 6-byte forms (KEY_SEQUENCE letters; the hosts encode the same way)
inline size_t letter_encode(uint32_t v, uint8_t* o) {
    if (v < 0x200000) return utf8_encode(v, o);
    const int n = v < 0x4000000 ? 5 : 6;
    o[0] = (uint8_t)((n == 5 ? 0xF8 : 0xFC) | (v >> (6 * (n - 1))));
    for (int k = n - 2, i = 1; k >= 0; k--, i++) o[i] = (uint8_t)(0x80 | ((v >> (6 * k)) & 0x3F));
    return (size_t)n;
}

template <class ChildIndex>
int build_trie_reinsert(const std::vector<RawNode>& raw, bool values_by_position, size_t pair_size, int letter_bytes, bool sequence,
                        ChildIndex child_index, acx_trie** out, int64_t* n_eow_out) {
    const size_t n = raw.size();
    if (n == 0 || n >= ((size_t)1 << 31)) return acx_fail(ACX_E_FORMAT, "reference dump: %zu nodes", n);
    acx_trie* t = nullptr;
    int rc = acx_trie_new(&t);
    if (rc) return rc;
    try {
        std::vector<uint8_t> linked(n, 0);
        struct Frame { size_t node; uint32_t next; size_t key_len; };
        std::vector<Frame> stack;
        std::vector<uint8_t> key;
        int64_t eow_seen = 0;
        size_t visited = 0;
        stack.push_back({0, 0, 0});
        // pre-order = dump order, so the i-th node entered is node i and eow ordinals follow the dump
        bool entering = true;
        while (!stack.empty() && rc == ACX_OK) {
            Frame& f = stack.back();
            if (entering) {
                if (f.node != visited) { rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu is not in pre-order", f.node); break; }
                visited++;
                if (raw[f.node].eow) {
                    if (f.node == 0) { rc = acx_fail(ACX_E_FORMAT, "reference dump: the root is marked as a key"); break; }
                    const int64_t v = values_by_position ? eow_seen : (int64_t)raw[f.node].output;
                    eow_seen++;
                    rc = acx_trie_add_word(t, key.data(), key.size(), v, nullptr);
                    if (rc) break;
                }
                entering = false;
            }
            if (f.next < raw[f.node].n) {
                const uint32_t j = f.next++;
                const int64_t c = child_index(f.node, j);
                if (c <= (int64_t)f.node || c >= (int64_t)n || linked[c]) { rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu has a malformed link #%u", f.node, j); break; }
                linked[c] = 1;
                const uint8_t* lp = raw[f.node].pairs + (size_t)j * pair_size;
                const uint32_t cp = letter_bytes == 4 ? rd32(lp) : rd16(lp);
                if (sequence ? cp > 0x7FFFFFFFu : cp > 0x10FFFF) { rc = acx_fail(ACX_E_FORMAT, "reference dump: node #%zu has letter %u: out of range", f.node, cp); break; }
                uint8_t enc[6];
                const size_t el = letter_encode(cp, enc);
                const size_t kl = key.size();
                key.insert(key.end(), enc, enc + el);
                stack.push_back({(size_t)c, 0, kl});
                entering = true;
            } else {
                key.resize(f.key_len);
                stack.pop_back();
            }
        }
        if (rc == ACX_OK && visited != n) rc = acx_fail(ACX_E_FORMAT, "reference dump: %zu of %zu nodes are not reachable from the root", n - visited, n);
        if (rc == ACX_OK && n_eow_out) *n_eow_out = eow_seen;
    } catch (const std::bad_alloc&) {
        rc = acx_fail(ACX_E_NOMEM, "reference dump: out of memory");
    }
    if (rc != ACX_OK) { acx_trie_free(t); return rc; }
    *out = t;
    return ACX_OK;
}

// The live nodes (reachable from the root) in ARENA order = the order they were created in.  Dumps
// are written in this order: a parent is always older than its children (what the loaders need), the
// reference's loader accepts any order, and a reload then recreates the arena as it was — which keeps
// the enumeration order of multi-byte letters (acx_items.cpp orders them by creation).  For a trie
// that was itself loaded from a reference dump this IS the reference's pre-order.
int preorder(const acx_trie* t, std::vector<int32_t>& order) {
    order.clear();
    if (t->kind == ACX_KIND_EMPTY || t->nodes.empty()) return ACX_OK;
    try {
        std::vector<uint8_t> live(t->nodes.size(), 0);
        std::vector<int32_t> stack;
        stack.push_back(0);
        size_t n_live = 0;
        while (!stack.empty()) {
            const int32_t k = stack.back();
            stack.pop_back();
            live[(size_t)k] = 1;
            n_live++;
            for (int32_t c = t->nodes[k].first_child; c >= 0; c = t->nodes[c].next_sibling) stack.push_back(c);
        }
        order.reserve(n_live);
        for (size_t k = 0; k < live.size(); k++) if (live[k]) order.push_back((int32_t)k);
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "reference dump: out of memory");
    }
    return ACX_OK;
}

// The trie as the writers see it: one entry per letter-level node, in dump order, with its
// children as (letter value, child position).
//   multibyte = false (bytes build, string keys): every byte node is a letter node; dump order =
//     creation order; the letter is the byte, sign-extended like the reference widens `char`.
//   multibyte = true (str flavour, KEY_SEQUENCE): a letter is a whole byte sequence; only the nodes
//     that end a letter are dumped, in pre-order with children in creation order — what the
//     re-inserting loader (and the reference) reads back in the same order.
struct LView {
    std::vector<int32_t> node;                                   // arena index of each dumped node
    std::vector<std::vector<std::pair<uint32_t, int64_t>>> kids; // per dumped node: (letter, child position)
    std::vector<int64_t> pos_of;                                 // arena index -> position, -1 if not dumped
};

int letter_view(const acx_trie* t, bool multibyte, LView& v) {
    v.node.clear(); v.kids.clear(); v.pos_of.assign(t->nodes.size(), -1);
    if (t->kind == ACX_KIND_EMPTY || t->nodes.empty()) return ACX_OK;
    try {
        if (!multibyte) {
            int rc = preorder(t, v.node);                        // (creation order)
            if (rc) return rc;
            for (size_t k = 0; k < v.node.size(); k++) v.pos_of[(size_t)v.node[k]] = (int64_t)k;
            v.kids.resize(v.node.size());
            for (size_t k = 0; k < v.node.size(); k++)
                for (int32_t c = t->nodes[v.node[k]].first_child; c >= 0; c = t->nodes[c].next_sibling)
                    v.kids[k].push_back({(uint32_t)(uint16_t)(int16_t)(int8_t)t->nodes[c].letter, v.pos_of[(size_t)c]});
            return ACX_OK;
        }
        struct Frame { int32_t node; size_t pos; std::vector<AcxLetterChild> ch; size_t next; };
        std::vector<Frame> stack;
        auto enter = [&](int32_t node) {
            Frame f;
            f.node = node; f.pos = v.node.size(); f.next = 0;
            acx_letter_children(t, node, true, f.ch);
            v.pos_of[(size_t)node] = (int64_t)f.pos;
            v.node.push_back(node);
            v.kids.emplace_back();
            stack.push_back(std::move(f));
        };
        enter(0);
        while (!stack.empty()) {
            Frame& f = stack.back();
            if (f.next < f.ch.size()) {
                const AcxLetterChild c = f.ch[f.next++];
                const size_t parent_pos = f.pos;
                const uint32_t letter = acx_letter_value(c.b, c.len);
                const size_t child_pos = v.node.size();
                enter(c.node);                                     // (invalidates f)
                v.kids[parent_pos].push_back({letter, (int64_t)child_pos});
            } else {
                stack.pop_back();
            }
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "reference dump: out of memory");
    }
    return ACX_OK;
}

// fail link of a dumped node as a position (+1), 0 = none
inline uint64_t fail_id(const acx_trie* t, const LView& v, size_t k) {
    if (t->kind != ACX_KIND_AHOCORASICK) return 0;
    const int32_t f = t->nodes[v.node[k]].fail;
    if (f < 0) return 0;
    const int64_t p = v.pos_of[(size_t)f];
    return p < 0 ? 1 : (uint64_t)p + 1;                           // (a fail link always ends a letter; the root otherwise)
}

inline void write_pair(uint8_t* p, int letter_bytes, uint32_t letter, uint64_t child) {
    if (letter_bytes == 4) { wr32(p, letter); wr64(p + 4, child); }
    else { wr16(p, (uint16_t)letter); wr64(p + 2, child); }
}

}  // namespace

extern "C" {

int acx_trie_from_ref_pickle(const void* const* chunks, const size_t* chunk_bytes, size_t n_chunks, int values_by_position,
                             int64_t longest_word, int letter_bytes, int sequence, acx_trie_t** out, int64_t* n_eow) {
    if (!chunks || !chunk_bytes || !out || n_chunks == 0) return acx_fail(ACX_E_INVAL, "acx_trie_from_ref_pickle: bad argument");
    if (letter_bytes != 2 && letter_bytes != 4) return acx_fail(ACX_E_INVAL, "acx_trie_from_ref_pickle: letter_bytes must be 2 (bytes build) or 4 (unicode build)");
    const size_t PAIR = letter_bytes == 4 ? PAIR_UCS4 : ::PAIR;
    std::vector<RawNode> raw;
    try {
        // automaton_unpickle__validate_bytes_list (src/Automaton_pickle.c:288-322): every chunk starts with a
        // positive node count; their sum is what links are checked against
        uint64_t count = 0;
        for (size_t c = 0; c < n_chunks; c++) {
            const uint8_t* p = (const uint8_t*)chunks[c];
            if (!p || chunk_bytes[c] < 8) return acx_fail(ACX_E_FORMAT, "Data truncated [parsing the nodes count]: chunk #%zu has %zu bytes", c, chunk_bytes[c]);
            const int64_t cnt = (int64_t)rd64(p);
            if (cnt <= 0) return acx_fail(ACX_E_FORMAT, "Nodes count for item #%zu on the bytes list is not positive (%llu)", c, (unsigned long long)cnt);
            count += (uint64_t)cnt;
        }
        // 1. the records (src/Automaton_pickle.c:362-418)
        std::vector<uint64_t> fail_raw;
        for (size_t c = 0; c < n_chunks; c++) {
            const uint8_t* data = (const uint8_t*)chunks[c];
            const uint8_t* p = data + 8;
            const uint8_t* end = data + chunk_bytes[c];
            const int64_t cnt = (int64_t)rd64(data);
            for (int64_t i = 0; i < cnt; i++) {
                if ((size_t)(end - p) < REC)
                    return acx_fail(ACX_E_FORMAT, "Data truncated [parsing header of node #%lld]: chunk #%zu @ offset %zu, expected at least %zu bytes",
                                    (long long)i, c, (size_t)(p - data), REC);
                RawNode r;
                r.output = rd64(p); r.n = rd32(p + 16); r.eow = p[20]; r.pairs = p + REC;
                fail_raw.push_back(rd64(p + 8));
                p += REC;
                if ((size_t)(end - p) < (size_t)r.n * PAIR)
                    return acx_fail(ACX_E_FORMAT, "Data truncated [parsing children of node #%lld]: chunk #%zu @ offset %zu, expected at least %zu bytes",
                                    (long long)i, c, (size_t)(p - data) + (size_t)i, (size_t)r.n * PAIR);
                p += (size_t)r.n * PAIR;
                raw.push_back(r);
            }
        }
        // 2. the links (src/Automaton_pickle.c:421-456): ids are 1-based, 0 = NULL, at most `count`
        const size_t lb = (size_t)letter_bytes;
        for (size_t k = 0; k < raw.size(); k++) {
            if (fail_raw[k] > count)
                return acx_fail(ACX_E_FORMAT, "Node #%zu malformed: the fail link points to node #%llu, while there are %llu nodes",
                                k, (unsigned long long)fail_raw[k], (unsigned long long)count);
            for (uint32_t j = 0; j < raw[k].n; j++) {
                const uint64_t child = rd64(raw[k].pairs + (size_t)j * PAIR + lb);
                if (child > count)
                    return acx_fail(ACX_E_FORMAT, "Node #%zu malformed: next link #%u points to node #%llu, while there are %llu nodes",
                                    k, j, (unsigned long long)child, (unsigned long long)count);
            }
        }
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_from_ref_pickle: out of memory");
    }
    const std::vector<RawNode>& rr = raw;
    if (letter_bytes == 4 || sequence) {
        const size_t lb = (size_t)letter_bytes;
        return build_trie_reinsert(raw, values_by_position != 0, PAIR, letter_bytes, sequence != 0,
                                   [&rr, PAIR, lb](size_t k, uint32_t j) -> int64_t { return (int64_t)rd64(rr[k].pairs + (size_t)j * PAIR + lb) - 1; },
                                   out, n_eow);
    }
    return build_trie(raw, values_by_position != 0, longest_word,
                      [&rr](size_t k, uint32_t j) -> int64_t { return (int64_t)rd64(rr[k].pairs + (size_t)j * ::PAIR + 2) - 1; },   // 1-based ids
                      out, n_eow);
}

int acx_trie_eow_values(const acx_trie_t* t, int letters_multibyte, int64_t** values, int64_t* n) {
    if (!t || !values || !n) return acx_fail(ACX_E_INVAL, "acx_trie_eow_values: NULL argument");
    LView v;
    int rc = letter_view(t, letters_multibyte != 0, v);
    if (rc) return rc;
    int64_t* out = (int64_t*)malloc(((size_t)t->count + 1) * sizeof(int64_t));
    if (!out) return acx_fail(ACX_E_NOMEM, "acx_trie_eow_values: out of memory");
    int64_t k = 0;
    for (int32_t i : v.node) if (t->nodes[i].eow && k < t->count) out[k++] = t->nodes[i].value;
    *values = out; *n = k;
    return ACX_OK;
}

int acx_trie_to_ref_pickle(const acx_trie_t* t, int values_by_position, size_t chunk_limit, int letter_bytes, int letters_multibyte,
                           void** buf, size_t** chunk_bytes, size_t* n_chunks) {
    if (!t || !buf || !chunk_bytes || !n_chunks) return acx_fail(ACX_E_INVAL, "acx_trie_to_ref_pickle: NULL argument");
    if (letter_bytes != 2 && letter_bytes != 4) return acx_fail(ACX_E_INVAL, "acx_trie_to_ref_pickle: letter_bytes must be 2 or 4");
    const size_t PAIR = letter_bytes == 4 ? PAIR_UCS4 : ::PAIR;
    if (chunk_limit < 8 + REC + 256 * PAIR) chunk_limit = (size_t)16 << 20;                // the reference's array size, src/Automaton_pickle.c:197
    LView v;
    int rc = letter_view(t, letters_multibyte != 0, v);
    if (rc) return rc;
    if (v.node.empty()) return acx_fail(ACX_E_STATE, "acx_trie_to_ref_pickle: empty automaton (the reference pickles it as Automaton())");
    try {
        std::vector<uint8_t> data;
        std::vector<size_t> sizes;
        size_t chunk_start = 0;
        int64_t in_chunk = 0;
        auto open_chunk = [&]() { chunk_start = data.size(); data.resize(data.size() + 8, 0); in_chunk = 0; };
        auto close_chunk = [&]() { wr64(data.data() + chunk_start, (uint64_t)in_chunk); sizes.push_back(data.size() - chunk_start); };
        open_chunk();
        for (size_t k = 0; k < v.node.size(); k++) {
            const Node& nd = t->nodes[v.node[k]];
            const size_t nc = v.kids[k].size();
            const size_t need = REC + nc * PAIR;
            if (in_chunk > 0 && data.size() - chunk_start + need > chunk_limit) { close_chunk(); open_chunk(); }
            const size_t at = data.size();
            data.resize(at + need, 0);
            uint8_t* p = data.data() + at;
            wr64(p, (nd.eow && !values_by_position) ? (uint64_t)nd.value : 0);
            wr64(p + 8, fail_id(t, v, k));
            wr32(p + 16, (uint32_t)nc);
            p[20] = nd.eow;
            p += REC;
            for (const auto& kc : v.kids[k]) { write_pair(p, letter_bytes, kc.first, (uint64_t)kc.second + 1); p += PAIR; }
            in_chunk++;
        }
        close_chunk();
        uint8_t* ob = (uint8_t*)malloc(data.size() ? data.size() : 1);
        size_t* os = (size_t*)malloc(sizes.size() * sizeof(size_t));
        if (!ob || !os) { free(ob); free(os); return acx_fail(ACX_E_NOMEM, "acx_trie_to_ref_pickle: out of memory"); }
        memcpy(ob, data.data(), data.size());
        memcpy(os, sizes.data(), sizes.size() * sizeof(size_t));
        *buf = ob; *chunk_bytes = os; *n_chunks = sizes.size();
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_to_ref_pickle: out of memory");
    }
    return ACX_OK;
}

int acx_trie_from_ref_savefile(const void* data, size_t nbytes, int letter_bytes, acx_trie_t** out, acx_ref_meta_t* meta,
                               int64_t** payload_off, int64_t** payload_len) {
    if (!data || !out || !meta) return acx_fail(ACX_E_INVAL, "acx_trie_from_ref_savefile: NULL argument");
    if (letter_bytes != 2 && letter_bytes != 4) return acx_fail(ACX_E_INVAL, "acx_trie_from_ref_savefile: letter_bytes must be 2 (bytes build) or 4 (unicode build)");
    const uint8_t* b = (const uint8_t*)data;
    if (nbytes < SAVE_HEADER + SAVE_FOOTER || memcmp(b, MAGIC, 16) != 0 || memcmp(b + nbytes - 16, MAGIC, 16) != 0)
        return acx_fail(ACX_E_FORMAT, "save file: bad magic (not a pyahocorasick002 file, or truncated)");   // src/custompickle/custompickle.c:35-52
    memset(meta, 0, sizeof *meta);
    meta->kind = (int32_t)rd32(b + 16); meta->store = (int32_t)rd32(b + 20); meta->key_type = (int32_t)rd32(b + 24);
    meta->count = (int64_t)rd64(b + 32); meta->longest_word = (int32_t)rd32(b + 40);
    const uint64_t n_nodes = rd64(b + nbytes - SAVE_FOOTER);
    meta->n_nodes = (int64_t)n_nodes;
    const bool any = meta->store == ACX_STORE_ANY;
    const bool reinsert = letter_bytes == 4 || meta->key_type == ACX_KEY_SEQUENCE;
    const size_t PAIR = letter_bytes == 4 ? PAIR_UCS4 : ::PAIR;
    *out = nullptr;
    if (payload_off) *payload_off = nullptr;
    if (payload_len) *payload_len = nullptr;
    if (n_nodes == 0) return ACX_OK;                                                       // an empty automaton: header + footer only
    if (n_nodes > (nbytes - SAVE_HEADER - SAVE_FOOTER) / (8 + REC)) return acx_fail(ACX_E_FORMAT, "save file: nodes count does not fit the file");
    std::vector<RawNode> raw;
    std::unordered_map<uint64_t, int64_t> index;
    std::vector<int64_t> poff, plen;
    try {
        raw.reserve(n_nodes);
        index.reserve(n_nodes * 2);
        const uint8_t* p = b + SAVE_HEADER;
        const uint8_t* end = b + nbytes - SAVE_FOOTER;
        for (uint64_t i = 0; i < n_nodes; i++) {
            if ((size_t)(end - p) < 8 + REC) return acx_fail(ACX_E_FORMAT, "save file: truncated at node #%llu", (unsigned long long)i);
            const uint64_t addr = rd64(p);
            p += 8;
            RawNode r;
            r.output = rd64(p); r.n = rd32(p + 16); r.eow = p[20]; r.pairs = p + REC;
            p += REC;
            if ((size_t)(end - p) < (size_t)r.n * PAIR) return acx_fail(ACX_E_FORMAT, "save file: truncated in the children of node #%llu", (unsigned long long)i);
            p += (size_t)r.n * PAIR;
            if (any && r.eow) {                                                            // serialized value follows, `output` bytes long
                if ((uint64_t)(end - p) < r.output) return acx_fail(ACX_E_FORMAT, "save file: truncated in the value of node #%llu", (unsigned long long)i);
                poff.push_back((int64_t)(p - b)); plen.push_back((int64_t)r.output);
                p += r.output;
            }
            if (!index.emplace(addr, (int64_t)i).second) return acx_fail(ACX_E_FORMAT, "save file: node #%llu repeats an address", (unsigned long long)i);
            raw.push_back(r);
        }
        if (p != end) return acx_fail(ACX_E_FORMAT, "save file: %zu stray bytes before the footer", (size_t)(end - p));
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_from_ref_savefile: out of memory");
    }
    const std::vector<RawNode>& rr = raw;
    const std::unordered_map<uint64_t, int64_t>& ix = index;
    int64_t n_eow = 0;
    int rc;
    if (reinsert) {
        const size_t lb = (size_t)letter_bytes;
        rc = build_trie_reinsert(raw, any, PAIR, letter_bytes, meta->key_type == ACX_KEY_SEQUENCE,
                                 [&rr, &ix, PAIR, lb](size_t k, uint32_t j) -> int64_t {
                                     auto it = ix.find(rd64(rr[k].pairs + (size_t)j * PAIR + lb));
                                     return it == ix.end() ? -1 : it->second;
                                 },
                                 out, &n_eow);
    }
    else
        rc = build_trie(raw, any, meta->longest_word,
                        [&rr, &ix](size_t k, uint32_t j) -> int64_t {
                            auto it = ix.find(rd64(rr[k].pairs + (size_t)j * ::PAIR + 2));
                            return it == ix.end() ? -1 : it->second;
                        },
                        out, &n_eow);
    if (rc) return rc;
    meta->n_eow = n_eow;
    if (any && payload_off && payload_len) {
        int64_t* o = (int64_t*)malloc((poff.size() + 1) * sizeof(int64_t));
        int64_t* l = (int64_t*)malloc((plen.size() + 1) * sizeof(int64_t));
        if (!o || !l) { free(o); free(l); acx_trie_free(*out); *out = nullptr; return acx_fail(ACX_E_NOMEM, "acx_trie_from_ref_savefile: out of memory"); }
        memcpy(o, poff.data(), poff.size() * sizeof(int64_t));
        memcpy(l, plen.data(), plen.size() * sizeof(int64_t));
        *payload_off = o; *payload_len = l;
    }
    return ACX_OK;
}

int acx_trie_to_ref_savefile(const acx_trie_t* t, int store, int key_type, int letter_bytes, int letters_multibyte,
                             const void* const* payloads, const size_t* payload_bytes, void** buf, size_t* nbytes) {
    if (!t || !buf || !nbytes) return acx_fail(ACX_E_INVAL, "acx_trie_to_ref_savefile: NULL argument");
    if (letter_bytes != 2 && letter_bytes != 4) return acx_fail(ACX_E_INVAL, "acx_trie_to_ref_savefile: letter_bytes must be 2 or 4");
    const size_t PAIR = letter_bytes == 4 ? PAIR_UCS4 : ::PAIR;
    const bool any = store == ACX_STORE_ANY;
    if (any && t->count > 0 && (!payloads || !payload_bytes)) return acx_fail(ACX_E_INVAL, "acx_trie_to_ref_savefile: STORE_ANY needs the serialized values");
    LView v;
    int rc = letter_view(t, letters_multibyte != 0, v);
    if (rc) return rc;
    try {
        std::vector<uint8_t> data(SAVE_HEADER, 0);
        memcpy(data.data(), MAGIC, 16);
        wr32(data.data() + 16, (uint32_t)t->kind); wr32(data.data() + 20, (uint32_t)store); wr32(data.data() + 24, (uint32_t)key_type);
        // longest_word counts letters in the reference; here it counts bytes of the trie (>= letters): harmless
        wr64(data.data() + 32, (uint64_t)t->count); wr32(data.data() + 40, (uint32_t)t->longest_word);
        // node "addresses": any distinct non-zero numbers do (the loader only uses them as keys)
        auto addr = [](int64_t pos) -> uint64_t { return 0x100000000ull + (uint64_t)pos * 32u; };
        int64_t eow_seen = 0;
        for (size_t k = 0; k < v.node.size(); k++) {
            const Node& nd = t->nodes[v.node[k]];
            const size_t nc = v.kids[k].size();
            const size_t pl = (any && nd.eow) ? payload_bytes[eow_seen] : 0;
            const size_t at = data.size();
            data.resize(at + 8 + REC + nc * PAIR + pl, 0);
            uint8_t* p = data.data() + at;
            wr64(p, addr((int64_t)k));
            p += 8;
            wr64(p, nd.eow ? (any ? (uint64_t)pl : (uint64_t)nd.value) : 0);
            const uint64_t f = fail_id(t, v, k);
            wr64(p + 8, f ? addr((int64_t)f - 1) : 0);
            wr32(p + 16, (uint32_t)nc);
            p[20] = nd.eow;
            p += REC;
            for (const auto& kc : v.kids[k]) { write_pair(p, letter_bytes, kc.first, addr(kc.second)); p += PAIR; }
            if (pl) memcpy(p, payloads[eow_seen], pl);
            if (nd.eow) eow_seen++;
        }
        const size_t at = data.size();
        data.resize(at + SAVE_FOOTER);
        wr64(data.data() + at, (uint64_t)v.node.size());
        memcpy(data.data() + at + 8, MAGIC, 16);
        uint8_t* ob = (uint8_t*)malloc(data.size());
        if (!ob) return acx_fail(ACX_E_NOMEM, "acx_trie_to_ref_savefile: out of memory");
        memcpy(ob, data.data(), data.size());
        *buf = ob; *nbytes = data.size();
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_trie_to_ref_savefile: out of memory");
    }
    return ACX_OK;
}

}  // extern "C"
