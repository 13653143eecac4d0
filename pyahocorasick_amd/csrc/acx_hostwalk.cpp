// acx_hostwalk.cpp — the search walks over the HOST trie (acx_trie_t), for what never pays a GPU launch:
// BASELINE.json's config 1 ("4-key automaton, Automaton.iter() over a 1 KB haystack on CPU: plumbing, no GPU"), a process
// without a device, and haystacks below the launch crossover (acx_set_host_walk_bytes, include/acx.h).  A 1 KB iter() is
// ~90 us as a GPU scan (launch + two copies) and 18-26 us in the reference; here it is a walk of the
// product's OWN arena trie, as the reference walks its own pointer trie:
//     iter / find_all   automaton_search_iter_next + automaton_build_output + ahocorasick_next
//                       (src/AutomatonSearchIter.c:243-300, :157-197; src/trie.c:177-194)
//     iter_long         automaton_search_iter_long_next (src/AutomatonSearchIterLong.c:89-153)
// It is not the oracle (oracle/ is test infrastructure and restates the reference on its own data structures) and no GPU
// parity result can come from it: tests/conftest.py switches it off for every `-m gpu` test and asserts
// acx_host_walk_calls() did not move.  Batches for the GPU never come here: the entry point refuses more than
// ACX_HOSTWALK_MAX_BYTES of haystack in one call.
#include "acx_trie_impl.h"

#include <cstring>
#include <new>
#include <vector>

namespace {

inline bool is_cspace(uint8_t b) { return b == ' ' || (b >= '\t' && b <= '\r'); }     // iswspace over the letters of a bytes build

// goto, else the fail links up to the root (src/trie.c:177-194)
inline int32_t ac_next(const acx_trie* t, int32_t state, uint8_t letter) {
    for (int32_t s = state; s >= 0; s = t->nodes[(size_t)s].fail) {
        const int32_t c = t->child(s, letter);
        if (c >= 0) return c;
    }
    return 0;
}

}  // namespace

// One haystack, every match (position ascending; at a position the state first, then its fail chain: longest key first —
// automaton_build_output).  ctx: the letters in front of the haystack; nothing that ends in there is reported.
// skip_ws: white space does not touch the state (src/AutomatonSearchIter.c:269-274); indices count the haystack's own bytes.
static void walk_all(const acx_trie* t, const uint8_t* ctx, int64_t n_ctx, const uint8_t* hay, int64_t n, int32_t base, bool skip_ws,
                     std::vector<acx_match_t>& out) {
    int32_t state = 0;
    for (int64_t i = 0; i < n_ctx; i++) state = ac_next(t, state, ctx[i]);
    for (int64_t i = 0; i < n; i++) {
        const uint8_t b = hay[i];
        if (skip_ws && is_cspace(b)) continue;
        state = ac_next(t, state, b);
        for (int32_t o = state; o > 0; o = t->nodes[(size_t)o].fail)
            if (t->nodes[(size_t)o].eow) out.push_back(acx_match_t{(int32_t)(base + i), (int32_t)t->nodes[(size_t)o].value});
    }
}

// iter_long: the reference's state machine, statement for statement (src/AutomatonSearchIterLong.c:101-150), on arena
// indices.  `state` in: where a stream stands (0 = root); out: where the walk stands when the haystack is exhausted.
static void walk_long(const acx_trie* t, const uint8_t* hay, int64_t n, int32_t base, int32_t* state_io, std::vector<acx_match_t>& out) {
    int32_t state = *state_io;
    int64_t index = -1;
    int32_t last_node = -1; int64_t last_index = -1;
    for (;;) {
        if (last_node >= 0) {                                           // return_output: report, start over behind the match (:101-110)
            out.push_back(acx_match_t{(int32_t)(base + last_index), (int32_t)t->nodes[(size_t)last_node].value});
            state = 0; index = last_index;
            last_node = -1; last_index = -1;
        }
        index += 1;
        bool report = false;
        while (index < n) {
            const int32_t next = t->child(state, hay[index]);
            if (next >= 0) {
                const Node& nx = t->nodes[(size_t)next];
                if (nx.eow) { last_node = next; last_index = index; }   // the last key on the path (:118-121)
                else if (nx.fail > 0 && t->nodes[(size_t)nx.fail].eow) {   // (:122-126)
                    last_node = nx.fail; last_index = index;
                    report = true;
                    break;
                }
                state = next;
                index += 1;
            } else if (last_node >= 0) { report = true; break; }        // (:131-132)
            else {
                for (;;) {                                               // (:134-144)
                    state = t->nodes[(size_t)state].fail;
                    if (state < 0) { state = 0; index += 1; break; }
                    if (t->child(state, hay[index]) >= 0) break;
                }
            }
        }
        if (report) continue;
        if (last_node >= 0) continue;                                    // (:148-150)
        break;
    }
    *state_io = state;
}

int acxi_hostwalk_batch(const acx_trie_t* t, int mode, const uint8_t* hay, const int64_t* off, int64_t n_hay,
                                  const uint8_t* ctx, const int64_t* ctx_off, const int32_t* init_node, const int32_t* index_base,
                                  int32_t flags, std::vector<int64_t>* moff, std::vector<acx_match_t>* m, std::vector<int32_t>* fin) {
    if (!t || !off || n_hay < 0 || !moff || !m || (n_hay && off[n_hay] > off[0] && !hay))
        return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: bad arguments");
    if (mode != ACX_SCAN_ALL && mode != ACX_SCAN_LONG) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: unknown mode");
    if (t->kind != ACX_KIND_AHOCORASICK)
        return acx_fail(ACX_E_STATE, "acx_trie_scan_host: not an Aho-Corasick automaton yet: call make_automaton first");
    if (mode == ACX_SCAN_LONG && (ctx || (flags & ACX_SCAN_SKIP_WS))) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: contexts and ACX_SCAN_SKIP_WS are for ACX_SCAN_ALL");
    if (mode == ACX_SCAN_ALL && init_node) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: ACX_SCAN_ALL continues from a context, not from a state");
    try {
        moff->assign((size_t)n_hay + 1, 0);
        m->clear();
        if (fin) fin->assign((size_t)n_hay, 0);
        const int64_t n_nodes = (int64_t)t->nodes.size();
        for (int64_t h = 0; h < n_hay; h++) {
            const int64_t lo = off[h], len = off[h + 1] - lo;
            if (len < 0) return acx_fail(ACX_E_INVAL, "acx_trie_scan_host: offsets are not monotone");
            const int32_t base = index_base ? index_base[h] : 0;
            if (mode == ACX_SCAN_ALL) {
                const int64_t clo = ctx ? ctx_off[h] : 0, clen = ctx ? ctx_off[h + 1] - clo : 0;
                walk_all(t, ctx ? ctx + clo : nullptr, clen, hay + lo, len, base, (flags & ACX_SCAN_SKIP_WS) != 0, *m);
            } else {
                // a carried state is -(node) - 1 (include/acx.h): never mistaken for a state id of a device image
                int32_t st = 0;
                if (init_node && init_node[h] < 0) { st = -(init_node[h] + 1); if (st >= n_nodes) st = 0; }
                walk_long(t, hay + lo, len, base, &st, *m);
                if (fin) (*fin)[(size_t)h] = st > 0 ? -st - 1 : 0;
            }
            (*moff)[(size_t)h + 1] = (int64_t)m->size();
        }
    } catch (const std::bad_alloc&) { return acx_fail(ACX_E_NOMEM, "acx_trie_scan_host: out of memory"); }
    return ACX_OK;
}
