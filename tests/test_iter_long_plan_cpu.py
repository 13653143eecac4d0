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
