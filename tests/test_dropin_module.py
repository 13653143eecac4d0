"""The CPython extension `ahocorasick` (dropin/): the drop-in host side of SURVEY.md §8b.
CPU part: module surface, trie API, error conventions (modelled on reference tests/test_unit.py).
GPU part (-m gpu): the reference's known-answer vectors and generated fixtures through
`import ahocorasick` exactly as a user of the reference would call it."""
import os
import sys

import pytest

from helpers import expected_pairs, load_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ahocorasick():
    from pyahocorasick_amd.build import build_dropin, DROPIN_DIR
    build_dropin(verbose=False)
    sys.path.insert(0, DROPIN_DIR)
    try:
        sys.modules.pop("ahocorasick", None)
        import ahocorasick as mod
        assert mod.__file__.startswith(DROPIN_DIR)
        yield mod
    finally:
        sys.path.remove(DROPIN_DIR)
        sys.modules.pop("ahocorasick", None)


def test_module_surface(ahocorasick):
    import ctypes
    dll = ctypes.CDLL(ahocorasick.__file__)
    assert hasattr(dll, "PyInit_ahocorasick")
    for name, val in dict(EMPTY=0, TRIE=1, AHOCORASICK=2, STORE_INTS=10, STORE_LENGTH=20, STORE_ANY=30,
                          KEY_STRING=100, KEY_SEQUENCE=200, MATCH_EXACT_LENGTH=0, MATCH_AT_MOST_PREFIX=1,
                          MATCH_AT_LEAST_PREFIX=2, unicode=0).items():
        assert getattr(ahocorasick, name) == val          # reference src/pyahocorasick.c:113-134
    for m in ("add_word", "exists", "get", "longest_prefix", "remove_word", "pop", "clear", "make_automaton",
              "iter", "iter_long", "find_all", "iter_batch"):
        assert callable(getattr(ahocorasick.Automaton, m))


def test_trie_api_and_errors(ahocorasick):
    with pytest.raises(ValueError):
        ahocorasick.Automaton(1234)
    with pytest.raises(ValueError):
        ahocorasick.Automaton(ahocorasick.STORE_ANY, 5)
    A = ahocorasick.Automaton()
    assert A.kind == ahocorasick.EMPTY and len(A) == 0 and A.store == ahocorasick.STORE_ANY
    with pytest.raises(ValueError):
        A.add_word(b"x")                                  # STORE_ANY needs a value
    with pytest.raises(TypeError):
        A.add_word("text", 1)
    assert A.add_word(b"he", "x") is True and A.add_word(b"he", "y") is False
    assert A.get(b"he") == "y" and b"he" in A and A.exists(b"he") and not A.exists(b"h")
    assert A.kind == ahocorasick.TRIE and len(A) == 1
    with pytest.raises(AttributeError):
        A.iter(b"she")
    with pytest.raises(AttributeError):
        A.iter_long(b"she")
    assert A.find_all(b"she", b"not even callable") is None      # silently None before make_automaton
    assert A.make_automaton() is None and A.kind == ahocorasick.AHOCORASICK and A.make_automaton() is False
    with pytest.raises(TypeError, match="bytes required"):
        A.iter("text")
    with pytest.raises(TypeError, match="callable"):
        A.find_all(b"she", None)
    with pytest.raises(TypeError):
        A.iter(b"x", ignore_white_space2=True)
    assert A.get(b"nope", 42) == 42
    with pytest.raises(KeyError):
        A.get(b"nope")
    with pytest.raises(KeyError):
        A.pop(b"nope")
    assert A.longest_prefix(b"hex") == 2
    assert A.pop(b"he") == "y" and len(A) == 0 and A.kind == ahocorasick.TRIE
    B = ahocorasick.Automaton(ahocorasick.STORE_INTS)
    B.add_word(b"a"); B.add_word(b"b"); B.add_word(b"c", 77)
    assert [B.get(k) for k in (b"a", b"b", b"c")] == [1, 2, 77]
    with pytest.raises(TypeError):
        B.add_word(b"d", "not an int")
    C = ahocorasick.Automaton(ahocorasick.STORE_LENGTH)
    C.add_word(b"hello")
    assert C.get(b"hello") == 5
    C.clear()
    assert C.kind == ahocorasick.EMPTY and len(C) == 0


def _build(mod, keys):
    A = mod.Automaton(mod.STORE_INTS)
    for i, k in enumerate(keys):
        A.add_word(k, i)
    A.make_automaton()
    return A


VECTORS = load_json("ref_vectors.json")["vectors"]
RANDOM = load_json("ref_random.json")


@pytest.mark.gpu
@pytest.mark.parametrize("v", VECTORS, ids=[v["id"] for v in VECTORS])
def test_dropin_reference_vectors(ahocorasick, v):
    keys = [bytes.fromhex(k) for k in v["keys_hex"]]
    hay = bytes.fromhex(v["hay_hex"])
    A = _build(ahocorasick, keys)
    rng = [] if v["start"] is None else ([v["start"]] if v["end"] is None else [v["start"], v["end"]])
    if v["mode"] == "iter":
        got = list(A.iter(hay, *rng))
    elif v["mode"] == "iter_long":
        got = list(A.iter_long(hay, *rng))
    else:
        got = []
        A.find_all(hay, lambda i, val: got.append((i, val)), *rng)
    assert got == expected_pairs(v["expected"]), v["source"]


@pytest.mark.gpu
def test_dropin_generated_fixtures_and_objects(ahocorasick):
    for c in RANDOM["cases"]:
        if c["store"] != "ints":
            continue
        keys = [bytes.fromhex(k) for k in c["keys_hex"]]
        A = ahocorasick.Automaton(ahocorasick.STORE_INTS)
        for k, val in zip(keys, c["values"]):
            A.add_word(k, val)
        A.make_automaton()
        hays = [bytes.fromhex(h["hay_hex"]) for h in c["hays"]]
        assert A.iter_batch(hays) == [expected_pairs(h["iter"]) for h in c["hays"]]
        assert A.iter_batch(hays, long=True) == [expected_pairs(h["iter_long"]) for h in c["hays"]]
        it = A.iter(b"")
        for part_hex, exp in zip(c["chunks"]["parts_hex"], c["chunks"]["iter_set"]):
            it.set(bytes.fromhex(part_hex))
            assert list(it) == expected_pairs(exp)
    # STORE_ANY objects, order within a position (reference tests/test_basic.py:18-50)
    A = ahocorasick.Automaton()
    for i, w in enumerate(b"he e hers his she hi him man he".split()):
        A.add_word(w, (i, w))
    A.make_automaton()
    q = b"he rshershidamanza "
    assert list(A.iter(string=q, start=2, end=8)) == [(6, (4, b"she")), (6, (8, b"he")), (6, (1, b"e"))]
    res = []
    A.find_all(q, lambda index, item: res.append((index, item)), 2, 11)
    assert res == [(6, (4, b"she")), (6, (8, b"he")), (6, (1, b"e")), (8, (2, b"hers")), (10, (5, b"hi"))]
    with pytest.raises(IndexError, match="end index not in range 0..19"):
        A.find_all(q, lambda i, v: None, 0, len(q) + 5)
    # white space skipping (reference tests/test_unit.py:813-849) and iterator invalidation (860-879)
    B = ahocorasick.Automaton()
    for w in "he her hers she".split():
        B.add_word(w.encode(), w)
    B.make_automaton()
    s = b"_sh e rher she_"
    assert list(B.iter(s, ignore_white_space=True)) == [(4, "she"), (4, "he"), (6, "her"), (8, "he"), (9, "her"),
                                                        (11, "hers"), (13, "she"), (13, "he")]
    it = B.iter(b"hehe")
    assert next(it) == (1, "he")
    B.add_word(b"zzz", "zzz")
    with pytest.raises(ValueError):
        next(it)


def test_longest_prefix_and_match_with_high_bytes_vs_reference(ahocorasick):
    """bytes 0x80..0xBF are letters like any other in the bytes build (a key's byte is one letter): they must
    not be mistaken for UTF-8 continuation bytes.  Cross-checked with the reference itself when it is present."""
    import random
    import pyahocorasick_amd as acx
    from oracle import orc
    ref = orc.load_reference()
    rng = random.Random(5)
    A, M = ahocorasick.Automaton(), acx.Automaton()
    R = ref.Automaton() if ref is not None else None
    keys = [b"\x88\x88\x90", b"3", b"\x80", b"a\xbf\xbfb", b"\xc3\xa9t\xc3\xa9"] + \
           [bytes(rng.choice([0x80, 0x90, 0xBF, 0x41, 0xC3]) for _ in range(rng.randint(1, 6))) for _ in range(40)]
    for k in keys:
        A.add_word(k, k)
        M.add_word(k, k)
        if R is not None:
            R.add_word(k, k)
    assert A.longest_prefix(b"\x88\x88\x90") == 3 and A.longest_prefix(b"3\x88") == 1      # the advisor's two cases
    probes = keys + [k[:-1] + b"\x80" for k in keys] + [k + b"\x90\x90" for k in keys] + [b"", b"\x90"]
    for p in probes:
        want = R.longest_prefix(p) if R is not None else M.longest_prefix(p)
        assert A.longest_prefix(p) == want == M.longest_prefix(p), p
        if R is not None:
            assert A.match(p) == R.match(p) == M.match(p), p
    E = ahocorasick.Automaton()
    assert E.match(b"") is False and acx.Automaton().match(b"") is False                   # trie_find on an empty trie
    if ref is not None:
        assert ref.Automaton().match(b"") is False


def test_store_any_value_slots_are_reused(ahocorasick):
    A = ahocorasick.Automaton()
    for round_ in range(50):
        for i in range(20):
            A.add_word(b"k%d" % i, ("v", round_, i))
        for i in range(20):
            assert A.pop(b"k%d" % i) == ("v", round_, i)
    A.add_word(b"x", 1)
    cls, args = A.__reduce__()                      # (what pickle calls)
    B = cls(*args)
    assert list(B.items()) == [(b"x", 1)]
    import pyahocorasick_amd as acx
    M = acx.Automaton()
    for round_ in range(50):
        for i in range(20):
            M.add_word(b"k%d" % i, (round_, i))
        for i in range(20):
            assert M.pop(b"k%d" % i) == (round_, i)
    assert len(M._values) <= 20


def test_dropin_add_words(ahocorasick):
    m = ahocorasick
    A, B = m.Automaton(m.STORE_INTS), m.Automaton(m.STORE_INTS)
    keys = [b"he", b"her", b"hers", b"she", b"he", b""]
    n_new = sum(bool(A.add_word(k, i)) for i, k in enumerate(keys))
    assert B.add_words(keys, list(range(len(keys)))) == n_new == 4
    assert sorted(A.items()) == sorted(B.items())
    D = m.Automaton()
    assert D.add_words([b"a", b"b"], [("x", 1), None]) == 2 and D.get(b"a") == ("x", 1)
    with pytest.raises(ValueError):
        B.add_words([b"q"], [1, 2])


@pytest.mark.gpu
def test_dropin_scans_release_the_gil_and_threads_share_an_automaton(ahocorasick):
    """scans run without the GIL: several threads scanning ONE automaton at once — each with a result
    object of its own, all on the same immutable device image — get the reference's answers, and a thread that
    changes the automaton meanwhile (new image) makes the old iterators stale, not wrong"""
    import threading
    m = ahocorasick
    A = m.Automaton(m.STORE_INTS)
    for i, w in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(w, i)
    A.make_automaton()
    hay = b"_sherhershe_" * 2000
    want = list(A.iter(hay))
    assert len(want) == 8 * 2000
    errs, counts = [], []

    def worker():
        try:
            for _ in range(30):
                got = list(A.iter(hay))
                assert got == want
                B_ = A.iter_batch([hay[:1200], hay[:24]])
                assert len(B_[1]) == 16
            counts.append(1)
        except Exception as ex:                        # noqa: BLE001
            errs.append(repr(ex))

    ts = [threading.Thread(target=worker) for _ in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs and len(counts) == 6, errs
    it = A.iter(hay)
    first = next(it)
    A.add_word(b"rhe", 9)                             # the automaton changes under a live iterator
    with pytest.raises(ValueError):
        next(it)
    assert first == want[0]
    A.make_automaton()
    assert len(list(A.iter(hay))) > len(want)


@pytest.mark.gpu
def test_dropin_iterator_refuses_a_second_thread_while_its_scan_runs(ahocorasick):
    """the scan behind next() runs without the GIL and holds pointers into the iterator's source: a second thread that calls
    next() on the SAME iterator meanwhile gets ValueError("iterator already executing"), as a generator would give (the
    reference holds the GIL throughout, src/AutomatonSearchIter.c:243-300)"""
    import threading
    m = ahocorasick
    A = m.Automaton(m.STORE_INTS)
    for i, w in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(w, i)
    A.make_automaton()
    hay = b"_sherhershe_" * 4_000_000                       # 48 MB: the first next() uploads and scans for tens of milliseconds
    want = list(A.iter(b"_sherhershe_"))
    refused = 0
    for attempt in range(5):
        it = A.iter(hay)
        start = threading.Barrier(2)
        got, errs = [], []

        def one():
            start.wait()
            try:
                got.append(next(it))
            except ValueError as ex:
                errs.append(str(ex))

        ts = [threading.Thread(target=one) for _ in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert all("already executing" in e for e in errs), errs
        assert sorted(got) == sorted(want[:len(got)]) and len(got) + len(errs) == 2      # whoever got through got the stream's next items
        refused += len(errs)
        if refused:
            break
    assert refused >= 1
