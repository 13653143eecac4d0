"""SURVEY §8f N4, dict-like methods: keys() / values() / items() / __iter__ with prefix, wildcard and
`how`, and get_stats(), on both hosts, against fixtures produced by running the reference
(tests/golden/make_items_golden.py).  The enumeration itself is acx_trie_items in libacx."""
import json
import os
import sys

import pytest

import pyahocorasick_amd as acx

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ref_items.json")))["cases"]


def dropin():
    from pyahocorasick_amd.build import build_dropin, DROPIN_DIR
    build_dropin(verbose=False)
    cached = sys.modules.get("ahocorasick")
    if cached is not None and getattr(cached, "__file__", "").startswith(DROPIN_DIR):
        return cached
    sys.modules.pop("ahocorasick", None)
    sys.path.insert(0, DROPIN_DIR)
    try:
        import ahocorasick
    finally:
        sys.path.remove(DROPIN_DIR)
    return ahocorasick


HOSTS = {"mirror": lambda: acx, "dropin": dropin}


def build(mod, case):
    A = mod.Automaton(case["store"])
    for k, v in zip(case["keys"], case["values"]):
        k = bytes.fromhex(k)
        if case["store"] == acx.STORE_LENGTH:
            A.add_word(k)
        else:
            A.add_word(k, v)
    if case["finalised"]:
        A.make_automaton()
    return A


@pytest.mark.parametrize("host", list(HOSTS))
def test_keys_values_items_match_the_reference(host):
    mod = HOSTS[host]()
    for case in CASES:
        A = build(mod, case)
        assert [k.hex() for k in A] == case["iter"]
        assert A.get_stats() == case["stats"]
        for q, want in zip(case["queries"], case["results"]):
            args = [bytes.fromhex(x) if isinstance(x, str) else x for x in q]
            keys = list(A.keys(*args))
            values = list(A.values(*args))
            assert [k.hex() for k in keys] == want["keys"]
            assert values == want["values"]
            assert list(A.items(*args)) == list(zip(keys, values))


@pytest.mark.parametrize("host", list(HOSTS))
def test_items_iterator_rules(host):
    mod = HOSTS[host]()
    A = mod.Automaton(mod.STORE_INTS)
    for i, k in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(k, i)
    assert sorted(A.keys()) == [b"he", b"her", b"hers", b"she"] and len(list(A)) == 4
    assert sorted(A.keys(b"he")) == [b"he", b"her", b"hers"]
    assert sorted(A.keys(b"h?r", b"?")) == [b"her"]                              # wildcard: exact length by default
    assert sorted(A.keys(b"h?", b"?", mod.MATCH_AT_LEAST_PREFIX)) == [b"he", b"her", b"hers"]
    assert sorted(A.keys(b"h?r?", b"?", mod.MATCH_AT_MOST_PREFIX)) == [b"he", b"her", b"hers"]
    with pytest.raises(ValueError, match="single character"):
        A.keys(b"he", b"??")
    with pytest.raises(ValueError, match="third argument"):
        A.keys(b"he", b"?", 7)
    with pytest.raises(TypeError):
        A.keys("text")
    it = A.items()
    assert next(it)[0] in (b"he", b"she")
    A.add_word(b"zz", 9)
    with pytest.raises(ValueError, match="no longer valid"):             # src/AutomatonItemsIter.c:133-136
        next(it)
    E = mod.Automaton()
    assert list(E.keys()) == [] and list(E) == [] and E.get_stats()["nodes_count"] == 0


def _canon(dump):
    """name every node by the key prefix that leads to it (the reference prints truncated addresses and
    dumps in pre-order, this engine numbers nodes in creation order): the root is the first node"""
    nodes, edges, fail = dump
    kids = {}
    for a, l, b in edges:
        kids.setdefault(a, []).append((l, b))
    path = {nodes[0][0]: b""}
    todo = [nodes[0][0]]
    while todo:
        a = todo.pop()
        for l, b in kids.get(a, []):
            path[b] = path[a] + l
            todo.append(b)
    return (sorted((path[n], e) for n, e in nodes), sorted((path[a], l, path[b]) for a, l, b in edges),
            sorted((path[a], path[b]) for a, b in fail))


@pytest.mark.parametrize("host", list(HOSTS))
def test_match_dump_sizeof(host):
    mod = HOSTS[host]()
    A = mod.Automaton(mod.STORE_INTS)
    assert A.dump() is None                                            # src/Automaton.c:1153-1154
    for i, k in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(k, i)
    assert A.match(b"h") and A.match(b"sh") and A.match(b"hers") and not A.match(b"hex") and A.match(b"")
    nodes, edges, fail = A.dump()
    assert len(nodes) == 8 and len(edges) == 7 and fail == []         # a plain trie has no fail links
    assert sum(e for _, e in nodes) == 4 and sorted(l for _, l, _ in edges) == [b"e", b"e", b"h", b"h", b"r", b"s", b"s"]
    A.make_automaton()
    nodes, edges, fail = A.dump()
    assert len(fail) == 7                                              # every node but the root
    # she -> he, sh -> h (the classic)
    canon = _canon((nodes, edges, fail))
    assert (b"sh", b"h") in canon[2] and (b"she", b"he") in canon[2] and (b"h", b"") in canon[2]
    assert A.__sizeof__() >= A.get_stats()["total_size"] > 0


def test_dump_equals_the_live_reference():
    from oracle import orc
    ref = orc.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref (the reference itself) is not built")
    import random
    rng = random.Random(9)
    keys = list({bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 7))) for _ in range(60)})
    R = ref.Automaton(ref.STORE_INTS)
    for i, k in enumerate(keys):
        R.add_word(k, i)
    R.make_automaton()
    want = _canon(R.dump())
    for name in HOSTS:
        mod = HOSTS[name]()
        A = mod.Automaton(mod.STORE_INTS)
        for i, k in enumerate(keys):
            A.add_word(k, i)
        A.make_automaton()
        assert _canon(A.dump()) == want
        assert all(A.match(k[:j]) for k in keys for j in range(len(k) + 1))
