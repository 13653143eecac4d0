"""The position-parallel form of iter_long that DESIGN.md §8b plans, pinned against the oracle BEFORE any kernel exists.

The serial walk (oracle/ac_oracle.c orc_iter_long, after /root/reference/src/AutomatonSearchIterLong.c:89-153) carries a
state; the plan replaces it by two things that depend on the text only:
    m(p)     how many symbols of the text from p follow the trie from the root (per start position),
    records  (end, start, kind, value) for every node on such a path that ends a key (E) or that does not while its fail
             node (not the root) does (FE: the value is the fail node's),
and a per-haystack sweep without any table: after a restart at r the walk's node at i is node(q, i - q + 1) with
q = min { p >= r : p + m(p) - 1 >= i }.  This file is that sweep in plain Python, compared with the oracle on dictionaries
built to hit the odd corners (nested keys, keys that are suffixes of prefixes of other keys, restarts inside a longer path).
Pure-Python loops: small cases only."""
import numpy as np
import pytest

from oracle import orc


class _Trie:
    def __init__(self, keys, values):
        self.child = [{}]
        self.value = [None]
        self.fail = [0]
        for k, v in zip(keys, values):
            n = 0
            for c in k:
                nx = self.child[n].get(c)
                if nx is None:
                    nx = len(self.child)
                    self.child.append({}); self.value.append(None); self.fail.append(0)
                    self.child[n][c] = nx
                n = nx
            self.value[n] = v
        order = list(self.child[0].values())
        for n in order:
            self.fail[n] = 0
        i = 0
        while i < len(order):
            n = order[i]; i += 1
            for c, nx in self.child[n].items():
                f = self.fail[n]
                while f and c not in self.child[f]:
                    f = self.fail[f]
                g = self.child[f].get(c, 0)
                self.fail[nx] = g if g != nx else 0
                order.append(nx)


def plan_iter_long(T, hay):
    """[(end, value)] by the decomposition: m(p), records along every path, one sweep."""
    n = len(hay)
    m = [0] * n
    by_start = [[] for _ in range(n)]           # records of the path from p, by depth: (depth, kind, value)
    for p in range(n):
        node, l = 0, 0
        while p + l < n and hay[p + l] in T.child[node]:
            node = T.child[node][hay[p + l]]; l += 1
            if T.value[node] is not None:
                by_start[p].append((l, "E", T.value[node]))
            else:
                f = T.fail[node]
                if f != 0 and T.value[f] is not None:
                    by_start[p].append((l, "FE", T.value[f]))
        m[p] = l
    out = []
    r = 0
    while r < n:
        # phase 1: the first position i >= r whose node (q, i - q + 1) carries a record
        q, i, hit = r, r, None
        while i < n:
            while q <= i and q + m[q] - 1 < i:
                q += 1
            if q <= i:
                for (l, kind, v) in by_start[q]:
                    if l == i - q + 1:
                        hit = (l, kind, v)
                if hit:
                    break
            i += 1
        if not hit:
            break
        l0, kind, v = hit
        if kind == "FE":
            out.append((i, v)); r = i + 1
            continue
        # phase 2: the later records of the same start; the first FE wins, else the deepest E within m(q)
        last = (i, v)
        done = False
        for (l, kind2, v2) in by_start[q]:
            if l <= l0:
                continue
            if kind2 == "FE":
                out.append((q + l - 1, v2)); r = q + l; done = True
                break
            last = (q + l - 1, v2)
        if not done:
            out.append(last); r = last[0] + 1
    return out


def plan_iter_long_records_only(T, hay, longest):
    """The same with NO per-position array in the sweep: every record carries its depth l and l_up, the length of the next
    longer trie path that ends at the same position (0: none).  After a restart at r the record (end i, l, l_up) is where
    phase 1 stops iff its start is at or behind r and the longer path's start is not: i - l_up + 1 < r <= i - l + 1.  Phase 2
    reads on through the records that start where it did (they end within `longest` positions)."""
    n = len(hay)
    m = [0] * n
    recs = []                                    # (end, depth, kind, value), then sorted by end
    for p in range(n):
        node, l = 0, 0
        while p + l < n and hay[p + l] in T.child[node]:
            node = T.child[node][hay[p + l]]; l += 1
            if T.value[node] is not None:
                recs.append((p + l - 1, l, "E", T.value[node]))
            else:
                f = T.fail[node]
                if f != 0 and T.value[f] is not None:
                    recs.append((p + l - 1, l, "FE", T.value[f]))
        m[p] = l
    recs.sort(key=lambda t: (t[0], t[1]))
    full = []
    for (i, l, kind, v) in recs:                 # l_up: what the position-parallel kernel finds in its tile's m[] (a short scan back)
        l_up = 0
        for p2 in range(i - l, max(-1, i - longest), -1):
            if p2 + m[p2] - 1 >= i:
                l_up = i - p2 + 1
                break
        full.append((i, l, l_up, kind, v))
    out, r, k = [], 0, 0
    while k < len(full):
        i, l, l_up, kind, v = full[k]
        fires = (i - l + 1 >= r) and (l_up == 0 or i - l_up + 1 < r)
        if not fires:
            k += 1
            continue
        if kind == "FE":
            out.append((i, v)); r = i + 1
        else:
            p, last, done = i - l + 1, (i, v), False
            for (i2, l2, _, kind2, v2) in full[k + 1:]:
                if i2 > p + longest - 1:
                    break
                if i2 - l2 + 1 != p:
                    continue
                if kind2 == "FE":
                    out.append((i2, v2)); r = i2 + 1; done = True
                    break
                last = (i2, v2)
            if not done:
                out.append(last); r = last[0] + 1
        while k < len(full) and full[k][0] < r:   # the walk goes on behind the match: records that end in front of r are over
            k += 1
    return out


def _check(keys, hays):
    keys = list(dict.fromkeys(keys))
    vals = list(range(100, 100 + len(keys)))
    O = orc.Oracle()
    for k, v in zip(keys, vals):
        O.add_word(k, v)
    O.make_automaton()
    T = _Trie(keys, vals)
    longest = max(len(k) for k in keys)
    for h in hays:
        want = O.iter_long(h)
        assert plan_iter_long(T, h) == want, (keys, h)
        assert plan_iter_long_records_only(T, h, longest) == want, (keys, h)


def test_plan_on_the_reference_examples():
    _check([b"he", b"her", b"hers", b"she"], [b"_sherhershe_", b"shers", b"hehehers", b"", b"h", b"sh"])
    _check([b"abcd", b"bc"], [b"abc", b"xbc", b"abcd", b"abcabcd", b"bcbc"])          # FE: a key that is the fail node of a longer path
    _check([b"abcde", b"bcd", b"c"], [b"abc", b"abcd", b"abcde", b"abcdx", b"ccc"])     # nothing at "abc" though "c" ends there
    _check([b"a", b"ab", b"bab", b"ba"], [b"abab", b"babab", b"bbaabb", b"aaaa"])


@pytest.mark.parametrize("seed", range(9))
def test_plan_on_random_dictionaries(seed):
    rng = np.random.default_rng(seed)
    alpha = [b"ab", b"abc", b"ACGT"][seed % 3]
    n_keys = int(rng.integers(2, 40))
    keys = [bytes(rng.choice(list(alpha), size=int(rng.integers(1, 9))).astype(np.uint8)) for _ in range(n_keys)]
    # nested families: prefixes and suffixes of a long key
    long_key = bytes(rng.choice(list(alpha), size=10).astype(np.uint8))
    keys += [long_key, long_key[:6], long_key[2:7], long_key[3:], long_key[4:5]]
    hays = [bytes(rng.choice(list(alpha), size=int(rng.integers(0, 80))).astype(np.uint8)) for _ in range(30)]
    hays += [long_key * 3, long_key[:9] + long_key, b""]
    _check(keys, hays)


# ---------------------------------------------------------------------------------------------------------------------
# The same through the product's host code: acx_blob_long_trie (acx_long.cpp) builds the dictionary D = E + FE + U from a flat
# image; the ORACLE's iter over D stands in for the position-parallel scan (records per haystack, position ascending,
# longest first); sweep_records below is the sweep with its loops spelled out, sweep_records_lockstep the form k_long_sweep
# (acx_long.hip) runs.
def long_dictionary(A):
    """-> (keys, packed values, real values, longest) of the dictionary the device scans for iter_long; None when the form does not apply"""
    import ctypes as C
    from pyahocorasick_amd._lib import check, lib
    blob = A.flat_image_bytes()
    trie, real, n, longest = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int32()
    check(lib().acx_blob_long_trie(blob, len(blob), C.byref(trie), C.byref(real), C.byref(n), C.byref(longest)))
    if n.value == 0:
        return None
    try:
        kb, ko, kv, kn = C.c_void_p(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)(), C.c_int64()
        check(lib().acx_trie_items(trie, None, 0, None, 0, 2, 0, C.byref(kb), C.byref(ko), C.byref(kv), C.byref(kn)))
        assert kn.value == n.value
        off = [ko[i] for i in range(kn.value + 1)]
        raw = C.string_at(kb.value, off[-1]) if off[-1] else b""
        keys = [raw[off[i]:off[i + 1]] for i in range(kn.value)]
        vals = [kv[i] for i in range(kn.value)]
        reals = list((C.c_int32 * n.value).from_address(real.value))
        for p in (kb, C.cast(ko, C.c_void_p), C.cast(kv, C.c_void_p)):
            lib().acx_blob_free(p)
        return keys, vals, reals, longest.value
    finally:
        lib().acx_trie_free(trie)
        lib().acx_blob_free(real)


def sweep_records(ends, vals, reals, longest, base=0):
    """k_long_sweep for one haystack: records (end, packed value) in scan order -> [(end, value)]"""
    out, r, k, n = [], base, 0, len(ends)
    while k < n:
        e, v = ends[k], vals[k] & 0xFFFFFFFF
        kind, ln = v >> 30, (v >> 24) & 63
        fires = False
        if kind:
            lup = ((vals[k - 1] & 0xFFFFFFFF) >> 24) & 63 if k > 0 and ends[k - 1] == e else 0
            fires = e - ln + 1 >= r and (lup == 0 or e - lup + 1 < r)
        if not fires:
            k += 1
            continue
        if kind == 2:
            out.append((e, reals[v & 0x3FFFF])); r = e + 1
        else:
            p, last, done = e - ln + 1, (e, reals[v & 0x3FFFF]), False
            j = k + 1
            while j < n and ends[j] <= p + longest - 1:
                v2 = vals[j] & 0xFFFFFFFF
                k2, l2 = v2 >> 30, (v2 >> 24) & 63
                if k2 and ends[j] - l2 + 1 == p:
                    if k2 == 2:
                        out.append((ends[j], reals[v2 & 0x3FFFF])); r = ends[j] + 1; done = True
                        break
                    last = (ends[j], reals[v2 & 0x3FFFF])
                j += 1
            if not done:
                out.append(last); r = last[0] + 1
        while k < n and ends[k] < r:
            k += 1
    return out


def sweep_records_lockstep(ends, vals, reals, longest, base=0):
    """the same sweep in the form k_long_sweep runs it (acx_long.hip sweep_one, statement by statement): ONE record per trip
    of the loop, every condition a 0 / 1 word — looking for a stop, or following a path; after a report at the end of a path
    the records behind the reported one are read again"""
    INT_MIN = -(1 << 31)
    out, r, k, w, n = [], base, 0, 0, len(ends)
    prev_e, prev_len = INT_MIN, 0
    path, p, last_e, last_i, k_last = 0, 0, 0, 0, 0
    reach = longest - 1
    if n == 0:
        return out
    while True:
        eor = 1 if k >= n else 0
        if eor & (path ^ 1):
            break
        kk = n - 1 if eor else k
        e, v = ends[kk], vals[kk] & 0xFFFFFFFF
        kind, ln, idx = v >> 30, (v >> 24) & 63, v & 0x3FFFF                  # (dictionaries of fewer than 2^18 entries: bits 18-23 are `below`)
        start = e - ln + 1
        is_fe, is_ev = int(kind == 2), int(kind != 0)
        p_end = path & (eor | int(e > p + reach))
        p_hit = path & (p_end ^ 1) & is_ev & int(start == p)
        p_fe, p_e = p_hit & is_fe, p_hit & (is_fe ^ 1)
        s_act = (path ^ 1) & (eor ^ 1)
        longer_in = int(prev_e == e) & int(e - prev_len + 1 >= r)
        fires = s_act & is_ev & int(start >= r) & (longer_in ^ 1)
        s_fe, s_e = fires & is_fe, fires & (is_fe ^ 1)
        emit = p_end | p_fe | s_fe
        ox, oy = (last_e, last_i) if p_end else (e, idx)
        if emit:
            out.append((ox, reals[oy])); assert w <= kk
        w += emit
        r = ox + 1 if emit else r
        keep = p_e | s_e
        if keep:
            last_e, last_i = e, idx
        k_next = k_last + 1 if p_end else k + 1
        if keep:
            k_last = k
        if s_e:
            p = start
        seen = s_act | p_fe
        prev_e = INT_MIN if p_end else (e if seen else prev_e)
        prev_len = ln if seen else prev_len
        path = (path & (p_end ^ 1) & (p_fe ^ 1)) | s_e
        k = k_next
    return out


def sweep_records_lockstep_now(ends, vals, reals, longest, base=0, small=True):
    """round 6: the RAW lock-step sweep (sweep_one of acx_long.hip, statement by statement) with what round 5 found for the compact
    form: kinds 2 (FE) and 3 (an E node with nothing below it) are reported the moment they fire (`now`), a remembered E's path is
    followed for `below` letters (bits 18-23 of the value) when the dictionary is small, else for longest - 1.  No staging pass beyond a
    copy: the record in front travels in registers, U records cost a trip each"""
    INT_MIN = -(1 << 31)
    imask = (1 << 18) - 1 if small else 0xFFFFFF
    if not small:
        vals = [v & ~(63 << 18) for v in vals]
    out, r, k, w, n = [], base, 0, 0, len(ends)
    prev_e, prev_len = INT_MIN, 0
    path, p, last_e, last_i, k_last, limit = 0, 0, 0, 0, 0, 0
    reach = longest - 1
    if n == 0:
        return out
    while True:
        eor = 1 if k >= n else 0
        if eor & (path ^ 1):
            break
        kk = n - 1 if eor else k
        e, v = ends[kk], vals[kk] & 0xFFFFFFFF
        kind, ln, idx = v >> 30, (v >> 24) & 63, v & imask
        start = e - ln + 1
        now, is_ev = kind >> 1, int(kind != 0)
        p_end = path & (eor | int(e > limit))
        p_hit = path & (p_end ^ 1) & is_ev & int(start == p)
        p_now, p_e = p_hit & now, p_hit & (now ^ 1)
        s_act = (path ^ 1) & (eor ^ 1)
        longer_in = int(prev_e == e) & int(e - prev_len + 1 >= r)
        fires = s_act & is_ev & int(start >= r) & (longer_in ^ 1)
        s_now, s_e = fires & now, fires & (now ^ 1)
        emit = p_end | p_now | s_now
        ox, oy = (last_e, last_i) if p_end else (e, idx)
        if emit:
            out.append((ox, reals[oy])); assert w <= kk
        w += emit
        r = ox + 1 if emit else r
        keep = p_e | s_e
        if keep:
            last_e, last_i = e, idx
            limit = e + ((v >> 18) & 63) if small else start + reach
        k_next = k_last + 1 if p_end else k + 1
        if keep:
            k_last = k
        if s_e:
            p = start
        seen = s_act | p_now
        prev_e = INT_MIN if p_end else (e if seen else prev_e)
        prev_len = ln if seen else prev_len
        path = (path & (p_end ^ 1) & (p_now ^ 1)) | s_e
        k = k_next
    return out


def sweep_records_compact(ends, vals, reals, longest, base=0, small=True):
    """the sweep as k_long_sweep runs it since round 5 (acx_long.hip): the staging pass turns the raw records into compact ones —
    U records dropped, {start, up_start, value} kept for the E / FE records — and sweep_compact, statement by statement: a record
    fires iff up_start < r <= start; kinds 2 (FE) and 3 (an E node with nothing below it) are reported at once, kind 1 is
    remembered and its path followed — for `below` letters (bits 18-23 of the value: how far below the node the deepest E / FE
    node lies) when the dictionary is small (fewer than 2^18 entries), else for longest - 1 letters"""
    INT_MIN = -(1 << 31)
    imask = (1 << 18) - 1 if small else 0xFFFFFF
    if not small:
        vals = [v & ~(63 << 18) for v in vals]                  # (a large dictionary's values carry no `below`)
    comp = []
    for i, (e, v) in enumerate(zip(ends, vals)):
        v &= 0xFFFFFFFF
        if v >> 30 == 0:
            continue
        st = e - ((v >> 24) & 63) + 1
        up = INT_MIN
        if i > 0 and ends[i - 1] == e:
            up = e - (((vals[i - 1] & 0xFFFFFFFF) >> 24) & 63) + 1
        comp.append((st, up, v))
    out, r, k, w, n = [], base, 0, 0, len(comp)
    path, p, last_e, last_i, k_last, limit = 0, 0, 0, 0, 0, 0
    reach = longest - 1
    if n == 0:
        return out
    while True:
        eor = 1 if k >= n else 0
        if eor & (path ^ 1):
            break
        kk = n - 1 if eor else k
        start, up, v = comp[kk]
        kind, idx = v >> 30, v & imask
        e = start + ((v >> 24) & 63) - 1
        now = kind >> 1
        p_end = path & (eor | int(e > limit))
        p_hit = path & (p_end ^ 1) & int(start == p)
        fires = (path ^ 1) & (eor ^ 1) & int(start >= r) & int(up < r)
        hit = p_hit | fires
        emit = p_end | (hit & now)
        keep = hit & (now ^ 1)
        ox, oy = (last_e, last_i) if p_end else (e, idx)
        if emit:
            out.append((ox, reals[oy])); assert w <= kk
        w += emit
        r = ox + 1 if emit else r
        if keep:
            last_e, last_i = e, idx
            limit = e + ((v >> 18) & 63) if small else start + reach
            assert limit <= (start if fires else p) + reach
        k_next = k_last + 1 if p_end else k + 1
        if keep:
            k_last = k
        if fires & keep:
            p = start
        path = (path & (emit ^ 1)) | (fires & keep)
        k = k_next
    return out


def _check_product_dictionary(keys, hays, values=None):
    from helpers import build_pair
    keys = list(dict.fromkeys(keys))
    values = list(range(1000, 1000 + len(keys))) if values is None else values
    A, O = build_pair(keys, values)
    D = long_dictionary(A)
    assert D is not None
    dkeys, dvals, reals, longest = D
    assert longest == max(len(k) for k in dkeys) and len(set(dkeys)) == len(dkeys)
    # kind 3 = an E entry that is a proper prefix of no other E / FE entry (nothing below it in the trie can replace it), kind 1 = one that is
    ev = [k for k, v in zip(dkeys, dvals) if ((v & 0xFFFFFFFF) >> 30) != 0]
    for k, v in zip(dkeys, dvals):
        kind = (v & 0xFFFFFFFF) >> 30
        if kind in (1, 3):
            below = any(len(x) > len(k) and x.startswith(k) for x in ev)
            assert kind == (1 if below else 3), (k, kind)
    OD = orc.Oracle()
    for k, v in zip(dkeys, dvals):
        OD.add_word(k, v)
    OD.make_automaton()
    for h in hays:
        recs = OD.iter(h)
        got = sweep_records([e for e, _ in recs], [v for _, v in recs], reals, longest)
        assert got == O.iter_long(h), (keys, h)
        assert sweep_records_lockstep([e for e, _ in recs], [v for _, v in recs], reals, longest) == got, (keys, h)
        assert sweep_records_compact([e for e, _ in recs], [v for _, v in recs], reals, longest) == got, (keys, h)
        assert sweep_records_compact([e + 77 for e, _ in recs], [v for _, v in recs], reals, longest, base=77) == [(e + 77, v) for e, v in got], (keys, h)
        assert sweep_records_compact([e for e, _ in recs], [v for _, v in recs], reals, longest, small=False) == got, (keys, h)
        assert sweep_records_lockstep_now([e for e, _ in recs], [v for _, v in recs], reals, longest) == got, (keys, h)
        assert sweep_records_lockstep_now([e for e, _ in recs], [v for _, v in recs], reals, longest, small=False) == got, (keys, h)
        assert sweep_records_lockstep_now([e + 77 for e, _ in recs], [v for _, v in recs], reals, longest, base=77) == [(e + 77, v) for e, v in got], (keys, h)
        got = sweep_records([e + 77 for e, _ in recs], [v for _, v in recs], reals, longest, base=77)      # index_base
        assert got == [(e + 77, v) for e, v in O.iter_long(h)], (keys, h)


def test_product_dictionary_on_the_reference_examples():
    _check_product_dictionary([b"he", b"her", b"hers", b"she"], [b"_sherhershe_", b"shers", b"hehehers", b"", b"h", b"sh"])
    _check_product_dictionary([b"abcd", b"bc"], [b"abc", b"xbc", b"abcd", b"abcabcd", b"bcbc"])
    _check_product_dictionary([b"abcde", b"bcd", b"c"], [b"abc", b"abcd", b"abcde", b"abcdx", b"ccc"])
    _check_product_dictionary([b"a", b"ab", b"bab", b"ba"], [b"abab", b"babab", b"bbaabb", b"aaaa"])


@pytest.mark.parametrize("seed", range(8))
def test_product_dictionary_on_random_dictionaries(seed):
    rng = np.random.default_rng(100 + seed)
    alpha = [b"ab", b"abc", b"ACGT", bytes(range(60, 90))][seed % 4]
    n_keys = int(rng.integers(2, 60))
    keys = [bytes(rng.choice(list(alpha), size=int(rng.integers(1, 12))).astype(np.uint8)) for _ in range(n_keys)]
    long_key = bytes(rng.choice(list(alpha), size=14).astype(np.uint8))
    keys += [long_key, long_key[:6], long_key[2:7], long_key[3:], long_key[4:5]]
    hays = [bytes(rng.choice(list(alpha) + [33], size=int(rng.integers(0, 120))).astype(np.uint8)) for _ in range(40)]
    hays += [long_key * 3, long_key[:9] + long_key, b""]
    _check_product_dictionary(keys, hays, values=[int(x) for x in rng.integers(-2**31, 2**31, size=len(dict.fromkeys(keys)))])


def test_dictionary_does_not_apply_to_keys_longer_than_63():
    from helpers import build_pair
    A, _ = build_pair([b"a" * 70, b"b"])
    assert long_dictionary(A) is None


def test_dictionary_of_config5_has_the_size_the_design_counts_on():
    """100 k ACGT keys of 8-32 letters (BASELINE config 2 / 5): D = 100 000 E + 59 612 FE + 20 921 U-only = 180 533 entries, none longer
    than 32 — the numbers DESIGN.md §4.3b prices the position-parallel iter_long with (1.8 x the scan of the keys alone)."""
    from helpers import build_pair
    from pyahocorasick_amd.workloads import dna_keys
    keys = dna_keys(100_000, seed=0)
    A, _ = build_pair(keys)
    dkeys, dvals, reals, longest = long_dictionary(A)
    kinds = np.array([(v & 0xFFFFFFFF) >> 30 for v in dvals])
    assert len(dkeys) == 180_533 and int((kinds == 2).sum()) == 59_612 and int((kinds == 0).sum()) == 20_921
    # the 100 000 E entries: 95 440 have no E / FE node below them (kind 3: reported the moment they fire), 4 560 do (kind 1: 2 984 of
    # the 3 840 eight-letter keys among them — the keys most matches are of)
    assert int((kinds == 3).sum()) == 95_440 and int((kinds == 1).sum()) == 4_560
    assert longest == 32 and all(((v & 0xFFFFFFFF) >> 24) & 63 == len(k) for k, v in zip(dkeys[:2000], dvals[:2000]))
