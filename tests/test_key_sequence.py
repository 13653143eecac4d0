"""SURVEY §8f N4 — KEY_SEQUENCE automata (keys and haystacks are tuples of integers) in the drop-in
extension.  Each letter is stored in the byte trie as one self-synchronising byte sequence (UTF-8
continued to 31 bits), so the same machinery as the str flavour applies.  Fixtures were produced
by running the reference (tests/golden/make_sequence_golden.py)."""
import json
import os
import pickle
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ref_sequence.json")))["cases"]


@pytest.fixture(scope="module")
def D():
    from pyahocorasick_amd.build import build_dropin, DROPIN_DIR
    build_dropin(verbose=False)
    saved = sys.modules.pop("ahocorasick", None)
    sys.path.insert(0, DROPIN_DIR)
    try:
        import ahocorasick as mod
        assert mod.__file__.startswith(DROPIN_DIR) and mod.unicode == 0
        yield mod
    finally:
        sys.path.remove(DROPIN_DIR)
        sys.modules.pop("ahocorasick", None)
        if saved is not None:
            sys.modules["ahocorasick"] = saved


def build(D, case, finalise=False):
    A = D.Automaton(case["store"], D.KEY_SEQUENCE)
    for i, k in enumerate(case["keys"]):
        k = tuple(k)
        if case["store"] == D.STORE_LENGTH:
            A.add_word(k)
        elif case["store"] == D.STORE_INTS:
            A.add_word(k, i - 3)
        else:
            A.add_word(k, [i, list(k)])
    if finalise:
        A.make_automaton()
    return A


def test_sequence_trie_api_matches_the_reference(D):
    for case in CASES:
        A = build(D, case)
        assert len(A) == len(case["keys"])
        assert [x.hex() for x in A.keys()] == case["enum_keys"] and list(A.values()) == case["enum_values"]
        for p, (keys, values) in zip(case["pats"], case["pat_results"]):
            args = [bytes.fromhex(q) for q in p]
            assert [x.hex() for x in A.keys(*args)] == keys and list(A.values(*args)) == values
        for p, want in zip(case["probes"], case["probe_results"]):
            p = tuple(p)
            assert [A.exists(p), A.match(p), A.longest_prefix(p), A.get(p, None), p in A] == want
        B = pickle.loads(pickle.dumps(A))                             # dumps are written in creation order: order kept
        assert [x.hex() for x in B.keys()] == case["enum_keys"] and len(B) == len(A)


def test_reference_sequence_dumps_load_and_own_files_round_trip(D, tmp_path):
    """the reference's pickles / save files of KEY_SEQUENCE automata (letters = the integers) are
    re-encoded on load; files written here are marked and read back node for node"""
    for case in CASES:
        r = case["reduce"]
        values = None if r["values_pickle"] is None else pickle.loads(bytes.fromhex(r["values_pickle"]))
        A = D.Automaton([bytes.fromhex(c) for c in r["chunks"]], *r["rest"], values)
        path = str(tmp_path / "ref.sav")
        open(path, "wb").write(bytes.fromhex(case["savefile"]))
        B = D.load(path, pickle.loads)
        own = str(tmp_path / "own.sav")
        if case["store"] == D.STORE_ANY:
            B.save(own, lambda v: pickle.dumps(v, protocol=2))
        else:
            B.save(own)
        C = D.load(own, pickle.loads)
        for X in (A, B, C):
            assert X.kind == D.AHOCORASICK and len(X) == len(case["keys"])
            assert [x.hex() for x in X.keys()] == case["enum_keys"] and list(X.values()) == case["enum_values"]
            for p, want in zip(case["probes"], case["probe_results"]):
                p = tuple(p)
                assert [X.exists(p), X.match(p), X.longest_prefix(p), X.get(p, None), p in X] == want


def test_sequence_argument_rules(D):                                     # src/utils.c:238-289
    A = D.Automaton(D.STORE_INTS, D.KEY_SEQUENCE)
    with pytest.raises(TypeError, match="not a supported sequence type"):
        A.add_word(b"ab", 1)
    with pytest.raises(ValueError, match=r"item #1: value 70000 outside range \[0..65535\]"):
        A.add_word((1, 70000), 1)
    with pytest.raises(ValueError, match="item #0 is not a number"):
        A.add_word(("x",), 1)
    A.add_word((1, 2, 3), 7)
    A.make_automaton()
    with pytest.raises(TypeError, match="tuple required"):
        A.iter([1, 2, 3])
    with pytest.raises(TypeError, match="bytes expected"):
        A.keys((1,))
    assert pickle.loads(pickle.dumps(A)).get((1, 2, 3)) == 7


@pytest.mark.gpu
def test_sequence_search_matches_the_reference(D):
    for case in CASES:
        A = build(D, case, finalise=True)
        for h, it, il, ir in zip(case["hays"], case["iter"], case["iter_long"], case["iter_range"]):
            h = tuple(h)
            assert [list(m) for m in A.iter(h)] == it
            assert [list(m) for m in A.iter_long(h)] == il
            if ir is not None:
                assert [list(m) for m in A.iter(h, 1, len(h) - 1)] == ir
            found = []
            A.find_all(h, lambda i, v: found.append([i, v]))
            assert found == it
        assert [[list(m) for m in r] for r in A.iter_batch([tuple(h) for h in case["hays"]])] == case["iter"]


def test_sequence_dumps_written_here_load_in_the_reference(D, tmp_path):
    """KEY_SEQUENCE pickles / save files written here carry the integers as the bytes build's uint16
    letters: the live reference (subprocess) reads them"""
    import subprocess
    ref_dir = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
    if not any(f.startswith("ahocorasick") and f.endswith(".so") for f in (os.listdir(ref_dir) if os.path.isdir(ref_dir) else [])):
        pytest.skip("oracle/_ref (the reference itself) is not built")
    jobs = []
    for i, case in enumerate(CASES[:9]):
        A = build(D, case, finalise=True)
        pkl, sav = str(tmp_path / ("c%d.pkl" % i)), str(tmp_path / ("c%d.sav" % i))
        open(pkl, "wb").write(pickle.dumps(A, protocol=2))
        if case["store"] == D.STORE_ANY:
            A.save(sav, lambda v: pickle.dumps(v, protocol=2))
        else:
            A.save(sav)
        jobs.append({"pkl": pkl, "sav": sav, "hays": case["hays"]})
    code = r'''
import sys, json, pickle
sys.path.insert(0, %r)
import ahocorasick as R
assert R.unicode == 0 and "_ref" in R.__file__
out = []
for job in json.load(sys.stdin):
    res = []
    for x in (pickle.load(open(job["pkl"], "rb")), R.load(job["sav"], pickle.loads)):
        res.append({"keys": [k.hex() for k in x.keys()], "values": list(x.values()),
                    "iter": [[list(m) for m in x.iter(tuple(h))] for h in job["hays"]]})
    out.append(res)
print(json.dumps(out))
''' % ref_dir
    r = subprocess.run([sys.executable, "-c", code], input=json.dumps(jobs), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    for case, res2 in zip(CASES[:9], json.loads(r.stdout)):
        for res in res2:
            assert res["keys"] == case["enum_keys"] and res["values"] == case["enum_values"] and res["iter"] == case["iter"]


def test_the_ctypes_mirror_hands_sequence_automata_to_the_extension():
    """pyahocorasick_amd.Automaton(store, KEY_SEQUENCE) IS the extension's automaton (one implementation of KEY_SEQUENCE): not an
    instance of the mirror class — a subclass loses its type there too —, without the numpy-level batch methods, searchable and
    picklable like any automaton of the extension (the class docstring says so)"""
    import pyahocorasick_amd as acx
    A = acx.Automaton(acx.STORE_INTS, acx.KEY_SEQUENCE)
    assert not isinstance(A, acx.Automaton) and type(A).__module__ == "ahocorasick"

    class Sub(acx.Automaton):
        pass
    S = Sub(acx.STORE_INTS, acx.KEY_SEQUENCE)
    assert type(S) is type(A) and not isinstance(S, Sub)
    assert isinstance(Sub(acx.STORE_INTS), Sub)                          # (byte automata keep the subclass)
    for name in ("scan_batch", "flat_image_bytes", "flatten_flags"):      # (add_words and iter_batch exist on the extension too)
        assert not hasattr(A, name), name
    for i, k in enumerate([(1, 2, 3), (2, 3), (60000, 5)]):
        A.add_word(k, i)
    A.make_automaton()
    assert list(A.iter((9, 1, 2, 3, 60000, 5))) == [(3, 0), (3, 1), (5, 2)]
    B = pickle.loads(pickle.dumps(A))
    assert type(B) is type(A) and list(B.iter((9, 1, 2, 3, 60000, 5))) == [(3, 0), (3, 1), (5, 2)] and sorted(B.keys()) == sorted(A.keys())
