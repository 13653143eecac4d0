// acx_ppm.cpp — builds the position-parallel scan image ("ppm", include/acx_blob.h) from the host
// trie: the trie of REVERSED keys, its top C levels direct-indexed by k-gram code (filter bitmap G
// for LDS, 32-byte cells, top_val), deeper nodes as dense child rows (kids/kval).
//
// What it replaces: nothing in the reference is built like this.  The reference reports at every
// position the current state's output chain (automaton_build_output,
// src/AutomatonSearchIter.c:157-197) = the keys that are suffixes of the text read so far, longest
// first.  A key is a suffix of text[..e] iff the reversed key is a path from the root of the
// reversed trie along text[e], text[e-1], ...  That walk needs no state from position e-1.
#include "acx_trie_impl.h"
#include "acx_internal.h"
#include "acx_ppm_layout.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

static_assert(sizeof(acx_ppm_header) == 256, "acx_ppm_header must be exactly 256 bytes");

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Threads that must not take the process down: what a job throws (std::bad_alloc, in practice) is caught on its thread
// and thrown again on the caller's, after every thread has been joined (emplace_back itself may throw, too).
struct Gang {
    std::vector<std::thread> th;
    std::mutex mu;
    std::exception_ptr first;
    template <class F> void guarded(F&& f) noexcept {
        try { f(); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    }
    template <class F> void spawn(F f) {
        try { th.emplace_back([this, f]() mutable { guarded(f); }); }
        catch (...) { std::lock_guard<std::mutex> g(mu); if (!first) first = std::current_exception(); }
    }
    void join() { for (auto& x : th) x.join(); th.clear(); if (first) { std::exception_ptr e = first; first = nullptr; std::rethrow_exception(e); } }
    ~Gang() { for (auto& x : th) if (x.joinable()) x.join(); }
};

// The trie of the REVERSED keys of `t`, built in one pass over the sorted reversed keys (a key shares a prefix
// with its predecessor: pop to that depth, append the rest).  Children come out in ascending letter order and are
// linked in O(1); acx_trie_add_word would walk sibling lists, which are 256 long at the top of a signature trie.
int build_reversed(const acx_trie* t, acx_trie* rev, std::vector<int32_t>* depth_out) {
    const bool timing_ = acx_tune_env("ACX_PPM_TIMING") != nullptr;
    auto t0_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char* what) { if (timing_) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[build_reversed] %s: %.3f s\n", what, std::chrono::duration<double>(t1 - t0_).count()); t0_ = t1; } };
    // 1. collect: DFS with an explicit stack; every key reversed into one buffer
    std::vector<uint8_t> buf;
    std::vector<uint64_t> koff;                 // key k = buf[koff[k] .. koff[k+1])
    std::vector<int64_t> kval;
    {
        // one depth-first walk per subtree of the root, on the host threads (the order of the keys does not matter:
        // they are sorted below)
        struct Part { std::vector<uint8_t> buf; std::vector<uint64_t> len; std::vector<int64_t> val; };
        std::vector<int32_t> tops;
        if (!t->nodes.empty())
            for (int32_t c = t->nodes[0].first_child; c >= 0; c = t->nodes[c].next_sibling) tops.push_back(c);
        std::vector<Part> parts(tops.size());
        std::atomic<size_t> next{0};
        auto worker = [&] {
            struct Frame { int32_t node; int32_t next_child; };
            std::vector<Frame> st;
            std::vector<uint8_t> path;
            for (size_t k = next.fetch_add(1); k < tops.size(); k = next.fetch_add(1)) {
                Part& P = parts[k];
                st.clear(); path.clear();
                const int32_t top = tops[k];
                path.push_back(t->nodes[top].letter);
                if (t->nodes[top].eow) { P.buf.insert(P.buf.end(), path.rbegin(), path.rend()); P.len.push_back(path.size()); P.val.push_back(t->nodes[top].value); }
                st.push_back({top, t->nodes[top].first_child});
                while (!st.empty()) {
                    Frame& f = st.back();
                    if (f.next_child < 0) { st.pop_back(); path.pop_back(); continue; }
                    const int32_t c = f.next_child;
                    f.next_child = t->nodes[c].next_sibling;
                    path.push_back(t->nodes[c].letter);
                    if (t->nodes[c].eow) { P.buf.insert(P.buf.end(), path.rbegin(), path.rend()); P.len.push_back(path.size()); P.val.push_back(t->nodes[c].value); }
                    st.push_back({c, t->nodes[c].first_child});
                }
            }
        };
        {
            size_t T = acx_host_threads();
            if (t->nodes.size() < 200000 || T > tops.size()) T = t->nodes.size() < 200000 ? 1 : tops.size();
            Gang gang;
            for (size_t k = 1; k < T; k++) gang.spawn([&] { worker(); });
            gang.guarded(worker);
            gang.join();
        }
        size_t total = 0, nkeys = 0;
        for (const Part& P : parts) { total += P.buf.size(); nkeys += P.val.size(); }
        buf.reserve(total); koff.reserve(nkeys + 1); kval.reserve(nkeys);
        koff.push_back(0);
        for (const Part& P : parts) {
            buf.insert(buf.end(), P.buf.begin(), P.buf.end());
            for (size_t i = 0; i < P.val.size(); i++) { koff.push_back(koff.back() + P.len[i]); kval.push_back(P.val[i]); }
        }
    }
    const size_t nk = kval.size();
    lap_("collect");
    // 2. sort the reversed keys
    std::vector<uint32_t> idx(nk);
    for (size_t i = 0; i < nk; i++) idx[i] = (uint32_t)i;
    const uint8_t* b = buf.data();
    auto less = [&](uint32_t x, uint32_t y) {
        const size_t lx = (size_t)(koff[x + 1] - koff[x]), ly = (size_t)(koff[y + 1] - koff[y]);
        const int c = memcmp(b + koff[x], b + koff[y], lx < ly ? lx : ly);
        return c != 0 ? c < 0 : lx < ly;
    };
    {   // sorted runs on the host threads, then pairwise merges (keys are unique: any stable or unstable order is the same)
        size_t T = acx_host_threads();
        while (T > 1 && nk / T < 20000) T >>= 1;
        size_t runs = 1; while (runs * 2 <= T) runs *= 2;          // a power of two
        std::vector<size_t> cut(runs + 1);
        for (size_t r = 0; r <= runs; r++) cut[r] = nk * r / runs;
        {
            Gang gang;
            for (size_t r = 1; r < runs; r++) gang.spawn([&, r] { std::sort(idx.begin() + cut[r], idx.begin() + cut[r + 1], less); });
            gang.guarded([&] { std::sort(idx.begin() + cut[0], idx.begin() + cut[1], less); });
            gang.join();
        }
        for (size_t w = 1; w < runs; w *= 2) {
            Gang gang;
            for (size_t r = 0; r + w < runs; r += 2 * w) {
                const size_t a = cut[r], m = cut[r + w], e = cut[r + 2 * w < runs ? r + 2 * w : runs];
                gang.spawn([&, a, m, e] { std::inplace_merge(idx.begin() + a, idx.begin() + m, idx.begin() + e, less); });
            }
            gang.join();
        }
    }
    lap_("sort");
    // 3. build.  The subtrie below every child of the root is built on its own (the sorted keys of one first letter are one
    // contiguous run), in an arena of its own, by the host threads; the arenas are then laid end to end in letter order,
    // which is exactly the arena a single pass over all keys would produce (nodes in pre-order, a parent before its
    // children): the same numbering, the same image, byte for byte.  (One pass took 0.8 s of the 6 s set-up of the
    // million-signature dictionary: 33 M nodes, first touched by one thread.)
    std::vector<int32_t>& depth = *depth_out;   // per node
    struct Sub { std::vector<Node> nodes; std::vector<int32_t> depth; size_t lo = 0, hi = 0; };
    std::vector<Sub> subs(256);
    {
        size_t q = 0;
        for (int c = 0; c < 256; c++) {            // (no key is empty: the trie stores none)
            subs[c].lo = q;
            while (q < nk && b[koff[idx[q]]] == (uint8_t)c) q++;
            subs[c].hi = q;
        }
        if (q != nk) return acx_fail(ACX_E_FORMAT, "acx_ppm_build: reversed keys out of order");
    }
    auto build_sub = [&](Sub& S, uint8_t letter) {
        size_t bytes = 0;
        for (size_t q = S.lo; q < S.hi; q++) bytes += (size_t)(koff[idx[q] + 1] - koff[idx[q]]);
        S.nodes.reserve(bytes); S.depth.reserve(bytes);      // (an upper bound: every byte of every key a node)
        std::vector<int32_t> last_child;        // per local node: its most recent child (siblings are appended in order)
        std::vector<int32_t> stack;             // stack[d - 1] = local node at depth d on the current key's path
        auto new_local = [&](uint8_t l, int32_t d) -> int32_t {
            Node n;
            n.value = 0; n.first_child = -1; n.next_sibling = -1; n.fail = -1; n.letter = l; n.eow = 0; n.wide = 0;
            S.nodes.push_back(n); S.depth.push_back(d); last_child.push_back(-1);
            return (int32_t)S.nodes.size() - 1;
        };
        stack.push_back(new_local(letter, 1));
        const uint8_t* prev = nullptr; size_t prev_len = 0;
        for (size_t q = S.lo; q < S.hi; q++) {
            const uint8_t* key = b + koff[idx[q]];
            const size_t len = (size_t)(koff[idx[q] + 1] - koff[idx[q]]);
            size_t lcp = 1;                         // (the first letter is the sub's own)
            const size_t m = len < prev_len ? len : prev_len;
            while (lcp < m && key[lcp] == prev[lcp]) lcp++;
            stack.resize(lcp);
            for (size_t d = lcp; d < len; d++) {
                const int32_t parent = stack[d - 1];
                const int32_t c = new_local(key[d], (int32_t)d + 1);
                if (last_child[parent] < 0) S.nodes[parent].first_child = c; else S.nodes[last_child[parent]].next_sibling = c;
                last_child[parent] = c;
                stack.push_back(c);
            }
            Node& nd = S.nodes[stack[len - 1]];
            nd.eow = 1; nd.value = kval[idx[q]];
            prev = key; prev_len = len;
        }
    };
    {
        std::atomic<int> next{0};
        auto worker = [&] { for (int c = next.fetch_add(1); c < 256; c = next.fetch_add(1)) if (subs[c].hi > subs[c].lo) build_sub(subs[c], (uint8_t)c); };
        size_t T = acx_host_threads();
        if (nk < 20000) T = 1;
        Gang gang;
        for (size_t k = 1; k < T; k++) gang.spawn([&] { worker(); });
        // Meanwhile this thread makes the final arena (an upper bound of it: every byte of every key a node): zero-filling
        // close to a gigabyte of fresh pages is the one serial piece left, so it runs beside the builders.
        gang.guarded([&] {
            if (T > 1) { rev->nodes.clear(); rev->nodes.resize(buf.size() + 1); depth.clear(); depth.resize(buf.size() + 1); }
            worker();
        });
        gang.join();
    }
    lap_("subtries (and the arena)");
    // the arenas end to end: local index i of sub c becomes first[c] + i
    std::vector<int64_t> first(257);
    first[0] = 1;
    for (int c = 0; c < 256; c++) first[c + 1] = first[c] + (int64_t)subs[c].nodes.size();
    if (first[256] > (int64_t)INT32_MAX) return acx_fail(ACX_E_UNSUPPORTED, "acx_ppm_build: more than 2^31 nodes");
    if (rev->nodes.size() < (size_t)first[256]) rev->nodes.clear();     // (not pre-sized above: one thread)
    rev->nodes.resize((size_t)first[256]);
    if (depth.size() < (size_t)first[256]) depth.clear();
    depth.resize((size_t)first[256]);
    {
        Node root;
        root.value = 0; root.first_child = -1; root.next_sibling = -1; root.fail = -1; root.letter = 0; root.eow = 0; root.wide = 0;
        int prev_c = -1;
        for (int c = 0; c < 256; c++) {
            if (subs[c].nodes.empty()) continue;
            rev->root_child[c] = (int32_t)first[c];
            if (prev_c < 0) root.first_child = (int32_t)first[c];
            else subs[prev_c].nodes[0].next_sibling = (int32_t)(first[c] - first[prev_c]);      // (local to prev_c's arena: rebased with the rest below)
            prev_c = c;
        }
        rev->nodes[0] = root; depth[0] = 0;
        std::atomic<int> next{0};
        auto worker = [&] {
            for (int c = next.fetch_add(1); c < 256; c = next.fetch_add(1)) {
                Sub& S = subs[c];
                const int32_t off = (int32_t)first[c];
                Node* dst = rev->nodes.data() + off;
                for (size_t i = 0; i < S.nodes.size(); i++) {
                    Node n = S.nodes[i];
                    if (n.first_child >= 0) n.first_child += off;
                    if (n.next_sibling >= 0) n.next_sibling += off;
                    dst[i] = n;
                }
                if (!S.depth.empty()) memcpy(depth.data() + off, S.depth.data(), S.depth.size() * sizeof(int32_t));
                std::vector<Node>().swap(S.nodes); std::vector<int32_t>().swap(S.depth);
            }
        };
        size_t T = acx_host_threads();
        if (nk < 20000) T = 1;
        Gang gang;
        for (size_t k = 1; k < T; k++) gang.spawn([&] { worker(); });
        gang.guarded(worker);
        gang.join();
    }
    rev->live_nodes = first[256];
    lap_("arenas end to end");
    rev->kind = ACX_KIND_TRIE;
    rev->count = (int64_t)nk;
    rev->longest_word = t->longest_word;
    return ACX_OK;
}

}  // namespace

// Returns ACX_OK with *out = nullptr when the automaton gets no ppm image (keys too long, deep rows
// too large): the scan then uses the serial walk kernels.  *out is malloc'd.
int acx_ppm_build(const acx_trie* t, const uint8_t* cls, uint32_t n_classes, bool has_other, uint8_t** out, size_t* nbytes, bool hot12) {
    *out = nullptr; *nbytes = 0;
    if (t->count <= 0 || t->longest_word <= 0 || t->longest_word > (int64_t)ACX_PPM_MAX_LONGEST) return ACX_OK;
    try {
        const uint32_t sigma = has_other ? n_classes - 1 : 256u;        // symbols = bytes that occur in keys
        if (sigma == 0) return ACX_OK;
        const uint32_t ho = has_other ? 1u : 0u;
        // byte -> symbol: byte order (class - 1), or (byte >> s) & 3 where a shift tells exactly four key bytes apart
        uint8_t symof[256];
        uint32_t sym_arith = 0, sym_lut = 0;
        for (int b = 0; b < 256; b++) symof[b] = (has_other && cls[b] == 0) ? 0xFFu : (uint8_t)(cls[b] - ho);
        if (sigma == 4 && !acx_tune_env("ACX_PPM_NO_ARITH")) {
            for (uint32_t sh = 0; sh <= 6 && !sym_arith; sh++) {
                uint32_t seen = 0, lut = 0;
                for (int b = 0; b < 256; b++) if (symof[b] != 0xFFu) { seen |= 1u << ((b >> sh) & 3); lut |= (uint32_t)b << (8 * ((b >> sh) & 3)); }
                if (seen == 0xFu) { sym_arith = 1 + sh; sym_lut = lut; }
            }
            if (sym_arith) for (int b = 0; b < 256; b++) if (symof[b] != 0xFFu) symof[b] = (uint8_t)((b >> (sym_arith - 1)) & 3);
        }

        const bool timing = acx_tune_env("ACX_PPM_TIMING") != nullptr;
        auto t0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) { if (timing) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "[acx_ppm_build] %s: %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count()); t0 = t1; } };
        acx_trie rev;
        std::vector<int32_t> depth;
        int rc = build_reversed(t, &rev, &depth);
        lap("reversed trie");
        if (rc) return rc;
        const size_t n = rev.nodes.size();
        if (n < 2) return ACX_OK;

        // Parent-first order: the arena itself (build_reversed creates the nodes in pre-order from the sorted keys), which
        // is also the order they lie in memory — the passes below stream through it instead of hopping level by level.
        std::vector<int32_t> order(n);
        int32_t max_depth = 0;
        for (size_t i = 0; i < n; i++) { order[i] = (int32_t)i; if (depth[i] > max_depth) max_depth = depth[i]; }
        lap("order");
        // parameters
        acx_ppm_header h;
        memset(&h, 0, sizeof h);
        h.magic = ACX_PPM_MAGIC; h.K = sigma; h.has_other = ho; h.longest = (uint32_t)t->longest_word;
        h.sym_bits = sigma <= 4 ? 2u : (sigma <= 16 ? 4u : 8u);
        h.pow2 = sigma == (1u << h.sym_bits) ? 1u : 0u;
        const uint32_t max_syms = 32u / h.sym_bits;                     // a window is 32 bits
        auto ipow = [&](uint32_t e) -> uint64_t { uint64_t p = 1; for (uint32_t i = 0; i < e; i++) { p *= sigma; if (p > ((uint64_t)1 << 40)) break; } return p; };
        uint32_t C = 0;
        uint32_t cell_bits = 18;                                        // cells: 32 bytes each, at most 2^cell_bits of them
        if (const char* e = acx_tune_env("ACX_PPM_CELL_BITS")) { const int v = atoi(e); if (v >= 4 && v <= 22) cell_bits = (uint32_t)v; }   // tuning hook
        const uint32_t c_cap = h.sym_bits == 2 ? 12u : 16u;            // (the hot cell: eowmask beside the child bits)
        while (C + 1 <= (uint32_t)max_depth && C + 1 <= max_syms && C + 1 <= c_cap && ipow(C + 1) <= ((uint64_t)1 << cell_bits)) C++;
        if (C == 0) return ACX_OK;
        const acx_ppm_lds base_layout = acx_ppm_lds_layout(0, h.sym_bits, h.longest);
        if (base_layout.total_words + 64 > ACX_PPM_LDS_BYTES / 4) return ACX_OK;
        const uint64_t gbits_cap = (uint64_t)(ACX_PPM_LDS_BYTES / 4 - base_layout.total_words - 8) * 32;
        uint32_t F = C;
        if (C + 1 <= (uint32_t)max_depth && C + 1 <= max_syms && ipow(C + 1) <= gbits_cap) F = C + 1;
        else if (h.sym_bits == 8 && C + 1 <= (uint32_t)max_depth && C + 1 <= max_syms && ipow(C + 1) <= ((uint64_t)1 << 27)) {
            // one level below the cells does not fit LDS (wide alphabets: 256^3 bits = 2 MB) but it is what tells most
            // positions apart from key ends: keep it in global memory (L2 resident), the stream kernel reads it there
            const char* gg = acx_tune_env("ACX_PPM_GLOBAL_FILTER");
            if (!(gg && gg[0] == '0')) { F = C + 1; h.g_global = 1; }
        }
        if (const char* e = acx_tune_env("ACX_PPM_MAX_F")) { const uint32_t v = (uint32_t)atoi(e); if (v >= 1 && v < F) { F = v; if (C > F) C = F; } }   // tuning hook
        if (!h.g_global && ipow(F) > gbits_cap) return ACX_OK;           // (F == C and even that does not fit: no image)
        h.C = C; h.F = F;
        // Second-level filter: the same question with more symbols, in global memory (L2 resident), asked only for the
        // positions that pass G.  It pays where G passes many positions that end no key and every candidate costs
        // dependent gathers: alphabets whose F is capped by the 32-bit window (text: four 8-bit symbols; config 3:
        // 219 -> 328 GB/s).  With 2-bit symbols G is not capped; a 12-symbol second level halved the candidates there
        // and gained nothing.
        uint32_t F2 = 0;
        {
            uint32_t g2_bits = 25;
            const char* e = acx_tune_env("ACX_PPM_G2_BITS");                   // tuning hook: 0 none, 10..27 the size cap
            if (e) { const int v = atoi(e); g2_bits = (v >= 10 && v <= 27) ? (uint32_t)v : 0u; }
            // (the kernel extends the code of F = max_syms symbols by older ones: only where the window caps F)
            if (g2_bits && !h.g_global && F == max_syms) {
                uint32_t f = F;
                while (f + 1 <= (uint32_t)max_depth && f + 1 < 2 * max_syms && ipow(f + 1) <= ((uint64_t)1 << g2_bits)) f++;
                if (f > F) F2 = f;
            }
        }
        h.F2 = F2;
        const uint64_t nC = ipow(C), nF = ipow(F), nF2 = F2 ? ipow(F2) : 0;
        h.g2_words = (uint32_t)((nF2 + 31) / 32);
        h.g_words = (uint32_t)((nF + 31) / 32);
        h.top_base[0] = 0;
        for (uint32_t d = 1; d <= C + 1; d++) h.top_base[d] = h.top_base[d - 1] + (uint32_t)ipow(d - 1);
        h.n_top = h.top_base[C + 1];

        // codes of the shallow nodes (depth <= F)
        std::vector<uint32_t> code(n, 0);
        std::vector<uint32_t> nkids(n, 0);
        uint32_t min_len = 0xFFFFFFFFu;
        {   // children per node and the shortest key: every node on its own, on the host threads
            std::mutex mu;
            parallel_range(0, n, [&](size_t a, size_t b2) {
                uint32_t ml = 0xFFFFFFFFu;
                for (size_t u = a; u < b2; u++) {
                    const Node& nd = rev.nodes[u];
                    if (nd.eow && (uint32_t)depth[u] < ml) ml = (uint32_t)depth[u];
                    uint32_t k = 0;
                    for (int32_t c = nd.first_child; c >= 0; c = rev.nodes[c].next_sibling) k++;
                    nkids[u] = k;
                }
                std::lock_guard<std::mutex> g(mu);
                if (ml < min_len) min_len = ml;
            });
        }
        {   // codes: a child's from its parent's, parents first (the arena's order); only the top levels have any
            const uint32_t lim = F2 > F ? F2 : F;
            for (size_t u = 0; u < n; u++) {
                if ((uint32_t)depth[u] >= lim) continue;
                for (int32_t c = rev.nodes[u].first_child; c >= 0; c = rev.nodes[c].next_sibling) code[c] = code[u] * sigma + (uint32_t)symof[rev.nodes[c].letter];
            }
        }
        h.min_len = min_len;

        // Deep structure: what the walk needs below the cells.  The walk STANDS on a node that has
        // children and takes one step per gather: a 16-byte record {label, len | flags, value, next}
        // that consumes 1 + len symbols (rows) or len symbols (singles):
        //   a node with two or more children owns a ROW of K records, indexed by the next symbol;
        //   a node with exactly one child owns a SINGLE record (its id carries bit 31).
        // A record follows the unbranched path below its first edge for up to 48 / sym_bits symbols,
        // stopping at the first node that is a key, branches or is a leaf; `next` is the id of that
        // node if it has children.  Random text leaves the trie within a level or two of the cells;
        // the long walks are occurrences of long keys, whose tails are unbranched.
        std::vector<uint32_t> deep(n, 0);           // id of a node the walk can stand on (0: none)
        std::vector<int32_t> row_nodes, single_nodes;
        // (a record's label: max_syms symbols in its first word, 16 / sym_bits more in the top half of its second — 24 symbols of a
        //  four-letter alphabet: with the C = 9 of the cells and a row's own symbol every key of up to 34 letters that shares no
        //  tail with another ends ONE gather below its cell.  A walk's steps are dependent round trips to the L2, and a wave
        //  takes a step when any of its walkers does.)
        const uint32_t more_syms = 16u / h.sym_bits, rec_syms = max_syms + more_syms;
        auto put_sym = [&](uint64_t& label, uint32_t len, int32_t v) {   // symbol number len (from 1) of a label: bits 63..32 the first word, 31..16 the second's top half
            const uint64_t sy = (uint64_t)symof[rev.nodes[v].letter];
            if (len <= max_syms) label |= sy << (64 - h.sym_bits * len);
            else label |= sy << (32 - h.sym_bits * (len - max_syms));
        };
        auto path_end = [&](int32_t c, uint32_t& len, uint64_t& label) -> int32_t {
            // follow the unbranched, key-free path below node c
            int32_t v = c;
            while (len < rec_syms && !rev.nodes[v].eow && nkids[v] == 1) {
                v = rev.nodes[v].first_child; len++;
                put_sym(label, len, v);
            }
            return v;
        };
        {
            std::vector<int32_t> work;
            auto want = [&](int32_t u) {           // node u (has children) gets an id
                if (deep[u]) return;
                if (nkids[u] >= 2) { row_nodes.push_back(u); deep[u] = (uint32_t)row_nodes.size(); }
                else { single_nodes.push_back(u); deep[u] = 0x80000000u | (uint32_t)single_nodes.size(); }
                work.push_back(u);
            };
            for (size_t i = 0; i < order.size(); i++) { const int32_t u = order[i]; if ((uint32_t)depth[u] == C && nkids[u]) want(u); }
            while (!work.empty()) {
                const int32_t u = work.back(); work.pop_back();
                if (nkids[u] >= 2) {
                    for (int32_t c = rev.nodes[u].first_child; c >= 0; c = rev.nodes[c].next_sibling) {
                        uint32_t len = 0; uint64_t label = 0;
                        const int32_t v = path_end(c, len, label);
                        if (nkids[v]) want(v);
                    }
                } else {
                    const int32_t c = rev.nodes[u].first_child;
                    uint32_t len = 1; uint64_t label = 0;
                    put_sym(label, 1, c);
                    const int32_t v = path_end(c, len, label);
                    if (nkids[v]) want(v);
                }
            }
        }
        lap("codes + deep ids");
        const uint32_t n_rows = (uint32_t)row_nodes.size(), n_single = (uint32_t)single_nodes.size();
        h.n_deep = n_rows; h.n_chain = n_single;
        const uint64_t row_bytes = (uint64_t)(n_rows + 1) * sigma * 16;
        uint64_t cap = (uint64_t)6 << 30;
        if (const char* e = acx_tune_env("ACX_PPM_MAX_DEEP_BYTES")) { const long long v = atoll(e); if (v > 0) cap = (uint64_t)v; }
        if (row_bytes > cap || n_rows >= 0x7FFFFFFFu || n_single >= 0x7FFFFFFFu) return ACX_OK;

        // layout
        size_t off = sizeof h;
        h.off_g = off;        off = align256(off + (size_t)h.g_words * 4);
        if (F2) { h.off_g2 = off; off = align256(off + (size_t)h.g2_words * 4); }
        h.sym_arith = sym_arith; h.sym_lut = sym_lut;
        h.off_symtab = off;   off = align256(off + 256);
        h.off_cells = off;    off = align256(off + (size_t)nC * 32);
        h.off_hot = off;      off = align256(off + (size_t)nC * 8);
        if (h.sym_bits == 2) {                                          // four-letter alphabets: the cells k_ppm_stream4 reads (one spare cell behind them, all zero)
            // (hot12 — ACX_FLATTEN_HOT12, the dictionaries of iter_long —: 12-byte cells with the id in the third word, no cid section)
            h.off_hot4 = off; off = align256(off + ((size_t)nC + 1) * (hot12 ? 12 : 8));
            if (!hot12) { h.off_cid = off;  off = align256(off + ((size_t)nC + 1) * 4); }
        }
        if (h.g_global) { h.off_gh = off; off = align256(off + (size_t)ACX_PPM_GH_WORDS * 4); }       // (dropped again below when it rejects too little)
        h.off_top_val = off;  off = align256(off + (size_t)h.n_top * 4);
        h.off_kids = off;     off = align256(off + (size_t)row_bytes);
        h.off_chains = off;   off = align256(off + ((size_t)n_single + 1) * 16);
        h.total_bytes = off;
        uint8_t* sec = (uint8_t*)calloc(1, off);
        if (!sec) return acx_fail(ACX_E_NOMEM, "acx_ppm_build: cannot allocate %zu bytes", off);
        uint32_t* G = (uint32_t*)(sec + h.off_g);
        uint32_t* cells = (uint32_t*)(sec + h.off_cells);
        uint32_t* hot = (uint32_t*)(sec + h.off_hot);
        uint32_t* hot4 = h.off_hot4 ? (uint32_t*)(sec + h.off_hot4) : nullptr;
        uint32_t* cid = h.off_cid ? (uint32_t*)(sec + h.off_cid) : nullptr;
        memcpy(sec + h.off_symtab, symof, 256);
        int32_t* top_val = (int32_t*)(sec + h.off_top_val);
        uint32_t* rows = (uint32_t*)(sec + h.off_kids);
        uint32_t* singles = (uint32_t*)(sec + h.off_chains);

        auto val32 = [&](const Node& nd) -> int32_t { return (int32_t)(uint32_t)(uint64_t)nd.value; };   // "ii" truncation, src/AutomatonSearchIter.c:180-184
        std::vector<uint8_t> top_eow(h.n_top, 0);
        std::vector<int32_t> topC_node(nC, -1);                         // arena index of node (C, code)
        for (size_t i = 1; i < order.size(); i++) {
            const int32_t u = order[i];
            const Node& nd = rev.nodes[u];
            const uint32_t d = (uint32_t)depth[u];
            if (d <= C) {
                if (nd.eow) { top_eow[h.top_base[d] + code[u]] = 1; top_val[h.top_base[d] + code[u]] = val32(nd); }
                if (d == C) topC_node[code[u]] = u;
            }
            if (d == F && F == C + 1) G[code[u] >> 5] |= 1u << (code[u] & 31);
            if (F2) {
                uint32_t* G2 = (uint32_t*)(sec + h.off_g2);
                if (d == F2) G2[code[u] >> 5] |= 1u << (code[u] & 31);
                else if (d < F2 && nd.eow) {                              // a shorter key: every filling of the older symbols
                    const uint64_t span = ipow(F2 - d), lo = (uint64_t)code[u] * span, hi = lo + span;
                    for (uint64_t x = lo; x < hi;) {
                        if ((x & 31) == 0 && x + 32 <= hi) { G2[x >> 5] = 0xFFFFFFFFu; x += 32; }
                        else { G2[x >> 5] |= 1u << (x & 31); x++; }
                    }
                }
            }
        }
        // ids as the kernels read them: a row's id is its first record's index (row number x K: no multiply on the
        // device), a single's id is its number with bit 31 set
        auto stored_id = [&](int32_t v) -> uint32_t { const uint32_t d = deep[v]; return (d >> 31) ? d : d * sigma; };
        auto fill = [&](uint32_t* rec, int32_t v, uint32_t len, uint64_t label) {
            rec[0] = (uint32_t)(label >> 32);
            rec[1] = len | (rev.nodes[v].eow ? 0x100u : 0u) | 0x200u | ((uint32_t)label & 0xFFFF0000u);   // 0x200: the record exists; top half: the symbols beyond the first word's
            rec[2] = rev.nodes[v].eow ? (uint32_t)val32(rev.nodes[v]) : 0u;
            rec[3] = stored_id(v);
        };
        parallel_range(0, n_rows, [&](size_t lo, size_t hi) {
            for (size_t b = lo; b < hi; b++) {
                const int32_t u = row_nodes[b];
                for (int32_t c = rev.nodes[u].first_child; c >= 0; c = rev.nodes[c].next_sibling) {
                    const uint32_t s1 = (uint32_t)symof[rev.nodes[c].letter];
                    uint32_t len = 0; uint64_t label = 0;
                    const int32_t v = path_end(c, len, label);
                    fill(rows + ((size_t)(b + 1) * sigma + s1) * 4, v, len, label);
                }
            }
        });
        parallel_range(0, n_single, [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; k++) {
                const int32_t u = single_nodes[k];
                const int32_t c = rev.nodes[u].first_child;
                uint32_t len = 1; uint64_t label = 0;
                put_sym(label, 1, c);
                const int32_t v = path_end(c, len, label);
                fill(singles + (size_t)(k + 1) * 4, v, len, label);
            }
        });
        lap("rows + singles");
        // cells: everything the d <= C newest symbols say
        for (uint64_t cc = 0; cc < nC; cc++) {
            uint32_t* cell = cells + cc * 8;
            uint64_t div = nC;
            uint32_t mask = 0, nv = 0;
            for (uint32_t d = 0; d <= C; d++) {                          // prefix code of length d = cc / sigma^(C-d)
                if (d > 0) {
                    const uint32_t pc = (uint32_t)(cc / div);
                    if (top_eow[h.top_base[d] + pc]) {
                        mask |= 1u << (d - 1);
                        if (nv < 5) cell[3 + nv++] = (uint32_t)top_val[h.top_base[d] + pc];   // values in match order (shortest first)
                    }
                }
                div /= sigma;
            }
            cell[0] = mask;
            const int32_t u = topC_node[cc];
            if (u >= 0 && deep[u]) {
                cell[1] = stored_id(u);
                if (sigma <= 4) {
                    uint32_t w = 0;
                    for (int32_t c = rev.nodes[u].first_child; c >= 0; c = rev.nodes[c].next_sibling) {
                        const uint32_t s1 = (uint32_t)symof[rev.nodes[c].letter];
                        w |= 1u << s1;
                        if (rev.nodes[c].eow) w |= 1u << (4 + s1);
                        for (int32_t g = rev.nodes[c].first_child; g >= 0; g = rev.nodes[g].next_sibling)
                            w |= 1u << (8 + s1 * 4 + (uint32_t)symof[rev.nodes[g].letter]);
                    }
                    cell[2] = w;
                }
            }
            // the hot cell: the same facts in 8 bytes
            {
                uint32_t hw = mask, hx = mask ? cell[3] : 0u;
                if (cell[1]) {
                    hx = cell[1];
                    if (h.sym_bits == 2) hw |= ((cell[2] >> 4) & 0xFu) << 12 | ((cell[2] >> 8) & 0xFFFFu) << 16;
                    else hw |= 0x80000000u;
                }
                hot[cc * 2] = hw; hot[cc * 2 + 1] = hx;
            }
            if (hot4) {                                                  // include/acx_blob.h "hot4": the value where a key ends, the id elsewhere
                uint32_t go = 0;
                if (cell[1]) for (uint32_t t4 = 0; t4 < 16; t4++) if (((cell[2] >> (4 + (t4 >> 2))) | (cell[2] >> (8 + t4))) & 1u) go |= 1u << t4;
                uint32_t xw = mask | go << 16;
                if ((cell[1] >> 31) && C <= 9) {                         // (bits 9 .. 12 are free of the mask: what k_ppm_stream4 reads has C = 9)
                    // the depth-C node has ONE child (its id is a single's): everything below it begins with the unbranched, key-free path of its
                    // record — bit 13, the path's symbols (at most seven: what the entry's window still holds; bits 16 .. 29, the first on top) instead of the
                    // 16 bits, and in bits 9 .. 12 how far to shift the 14 bits right to keep the path's own; the walk goes deeper iff the text agrees with
                    // all of them (include/acx_blob.h "hot4")
                    const uint32_t* rec = singles + (size_t)(cell[1] & 0x7FFFFFFFu) * 4;
                    const uint32_t len = rec[1] & 0xFFu, np = len < 7u ? len : 7u;
                    xw = mask | (14u - 2u * np) << 9 | 1u << 13 | (rec[0] >> 18) << 16;
                }
                if (cid) {
                    hot4[cc * 2] = xw;
                    hot4[cc * 2 + 1] = mask ? cell[3] : cell[1];
                    cid[cc] = cell[1];
                } else {                                                 // 12-byte cells: the value AND the id
                    hot4[cc * 3] = xw;
                    hot4[cc * 3 + 1] = mask ? cell[3] : 0u;
                    hot4[cc * 3 + 2] = cell[1];
                }
            }
            if (F == C) { if (cell[0] | cell[1]) G[cc >> 5] |= 1u << (cc & 31); }
            else if (cell[0]) for (uint32_t s = 0; s < sigma; s++) { const uint64_t x = cc * sigma + s; G[x >> 5] |= 1u << (x & 31); }
        }
        lap("cells + filter");
        if (h.off_gh) {                                                 // include/acx_blob.h "gh": the hashed copy of a global filter
            uint32_t* gh = (uint32_t*)(sec + h.off_gh);
            uint64_t n_set = 0;
            for (uint32_t w = 0; w < h.g_words; w++) {
                uint32_t m = G[w];
                while (m) {
                    const uint32_t code = w * 32u + (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    const uint32_t ix = ACX_PPM_GH_INDEX(code);
                    gh[ix >> 5] |= 1u << (ix & 31);
                    n_set++;
                }
            }
            uint64_t occ = 0;
            for (uint32_t w = 0; w < ACX_PPM_GH_WORDS; w++) occ += (uint64_t)__builtin_popcount(gh[w]);
            if (occ * 4 > (uint64_t)ACX_PPM_GH_BITS * 3) {              // more than 3 of 4 random codes would pass: not worth a probe
                memset(gh, 0, (size_t)ACX_PPM_GH_WORDS * 4);
                h.off_gh = 0;                                            // (the section stays in the blob, unused: offsets behind it are set)
            }
            (void)n_set;
            lap("hashed filter for LDS");
        }
        memcpy(sec, &h, sizeof h);
        *out = sec; *nbytes = off;
        return ACX_OK;
    } catch (const std::bad_alloc&) {
        return acx_fail(ACX_E_NOMEM, "acx_ppm_build: out of memory");
    } catch (const std::exception& e) {                             // (a thread that could not be started, ...)
        return acx_fail(ACX_E_NOMEM, "acx_ppm_build: %s", e.what());
    }
}
