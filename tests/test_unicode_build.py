"""SURVEY §8f N4 — the str flavour of the drop-in (dropin/unicode/ahocorasick, -DACX_UNICODE_BUILD):
keys and haystacks are str, one letter = one code point, as in the reference's unicode build.
The engine underneath is the byte automaton fed with UTF-8; indices are converted between bytes
and letters in the module.  Fixtures were produced by running the reference's unicode build
(tests/golden/make_unicode_golden.py)."""
import json
import os
import pickle
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "ref_unicode.json"), encoding="utf-8"))["cases"]


@pytest.fixture(scope="module")
def U():
    from pyahocorasick_amd.build import build_dropin, dropin_path
    so = build_dropin(verbose=False, unicode=True)
    d = os.path.dirname(dropin_path(True))
    saved = sys.modules.pop("ahocorasick", None)
    sys.path.insert(0, d)
    try:
        import ahocorasick as mod
        assert mod.__file__ == so and mod.unicode == 1
        yield mod
    finally:
        sys.path.remove(d)
        sys.modules.pop("ahocorasick", None)
        if saved is not None:
            sys.modules["ahocorasick"] = saved


def build(U, case, finalise=False):
    A = U.Automaton(case["store"])
    for k, v in zip(case["keys"], case["values"]):
        if case["store"] == U.STORE_LENGTH:
            A.add_word(k)
        else:
            A.add_word(k, v)
    if finalise:
        A.make_automaton()
    return A


def test_trie_api_and_enumeration_match_the_unicode_reference(U):
    for case in CASES:
        A = build(U, case)
        assert len(A) == len(case["keys"])
        assert list(A) == case["iter_keys"]
        for q, (keys, values) in zip(case["pats"], case["enum"]):
            assert list(A.keys(*q)) == keys and list(A.values(*q)) == values
            assert list(A.items(*q)) == list(zip(keys, values))
        for p, want in zip(case["probes"], case["probe_results"]):
            assert [A.exists(p), A.match(p), A.longest_prefix(p), A.get(p, None)] == want
        # own pickle round trip (UTF-8 payload, tagged); dumps are written in creation order, so the
        # enumeration order of letters that share a UTF-8 lead byte survives
        B = pickle.loads(pickle.dumps(A))
        assert list(B.keys()) == case["iter_keys"] and list(B.values()) == case["enum"][0][1]


def test_pickles_and_save_files_of_the_unicode_reference_load(U, tmp_path):
    """4-byte letters (src/common.h:50-56): the keys are re-inserted as UTF-8 (acx_persist.cpp)"""
    for case in CASES:
        r = case["reduce"]
        values = None if r["values_pickle"] is None else pickle.loads(bytes.fromhex(r["values_pickle"]))
        A = U.Automaton([bytes.fromhex(c) for c in r["chunks"]], *r["rest"], values)
        path = str(tmp_path / "u.sav")
        open(path, "wb").write(bytes.fromhex(case["savefile"]))
        B = U.load(path, pickle.loads)
        own = str(tmp_path / "own.sav")                                  # a file written by this flavour (marked UTF-8 payload)
        if case["store"] == U.STORE_ANY:
            B.save(own, lambda v: pickle.dumps(v, protocol=2))
        else:
            B.save(own)
        C = U.load(own, pickle.loads)
        assert list(C.keys()) == case["iter_keys"] and len(C) == len(B) and C.kind == U.AHOCORASICK
        for X in (A, B):
            assert X.kind == U.AHOCORASICK and len(X) == len(case["keys"])
            assert list(X.keys()) == case["iter_keys"] and list(X.values()) == case["enum"][0][1]
            for q, (keys, values_q) in zip(case["pats"], case["enum"]):
                assert list(X.keys(*q)) == keys


def test_str_build_rules(U):
    A = U.Automaton()
    with pytest.raises(TypeError, match="string expected"):
        A.add_word(b"bytes", 1)
    A.add_word("żółw", 1)
    assert A.get("żółw") == 1 and "żół" not in A and A.match("żół") and A.longest_prefix("żółty") == 3
    L = U.Automaton(U.STORE_LENGTH)
    L.add_word("日本語")
    assert L.get("日本語") == 3                                         # letters, not bytes
    with pytest.raises(ValueError, match="single character"):
        A.keys("ż", "??")
    with pytest.raises(ValueError):                                     # truncated 7-tuple payload
        U.Automaton([b"\x02" + b"\x00" * 31], 1, 10, 100, 0, 0, None)
    A.make_automaton()
    with pytest.raises(TypeError, match="string required"):
        A.iter(b"bytes")


@pytest.mark.gpu
def test_search_matches_the_unicode_reference(U):
    for case in CASES:
        A = build(U, case, finalise=True)
        for s in case["searches"]:
            h = s["hay"]
            assert [list(m) for m in A.iter(h)] == s["iter"]
            assert [list(m) for m in A.iter_long(h)] == s["iter_long"]
            assert [list(m) for m in A.iter(h, ignore_white_space=True)] == s["iter_ws"]
            if "range" in s:
                a, b = s["range"]
                assert [list(m) for m in A.iter(h, a, b)] == s["iter_range"]
                assert [list(m) for m in A.iter_long(h, a, b)] == s["iter_long_range"]
            found = []
            A.find_all(h, lambda i, v: found.append([i, v]))
            assert found == s["find_all"]
        hays = [s["hay"] for s in case["searches"]]
        assert [[list(m) for m in r] for r in A.iter_batch(hays)] == [s["iter"] for s in case["searches"]]
        assert [[list(m) for m in r] for r in A.iter_batch(hays, long=True)] == [s["iter_long"] for s in case["searches"]]
        if "set" in case:
            c = case["set"]
            it = A.iter(c["chunks"][0])
            first = [list(m) for m in it]
            it.set(c["chunks"][1])
            second = [list(m) for m in it]
            it.set(c["chunks"][2], True)
            third = [list(m) for m in it]
            assert [first, second, third] == c["results"]


@pytest.mark.gpu
def test_letters_outside_the_bmp_against_the_bytes_build(U):
    """4-byte UTF-8 letters (the reference's unicode build cannot be used to generate these
    fixtures: it corrupts memory on 4-byte-kind strings).  Expected results = the bytes engine on
    the UTF-8 encoding, end indices converted with Python's own codec."""
    import random
    import pyahocorasick_amd as acx
    rng = random.Random(4)
    alpha = "ab😀𝄞ż日 "
    keys = list({"".join(rng.choice(alpha.replace(" ", "")) for _ in range(rng.randint(1, 4))) for _ in range(40)})
    A = U.Automaton(U.STORE_INTS)
    B = acx.Automaton(acx.STORE_INTS)
    for i, k in enumerate(keys):
        A.add_word(k, i)
        B.add_word(k.encode("utf-8"), i)
    A.make_automaton()
    B.make_automaton()
    assert sorted(A.keys()) == sorted(keys) and sorted(A.keys("😀?", "?")) == sorted(k for k in keys if len(k) == 2 and k[0] == "😀")
    for _ in range(30):
        h = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 200)))
        hb = h.encode("utf-8")
        to_letter = {}
        pos = -1
        for ci, ch in enumerate(h):
            pos += len(ch.encode("utf-8"))
            to_letter[pos] = ci                                        # last byte of the letter -> its index
        want = [(to_letter[e], v) for e, v in B.iter(hb)]
        assert list(A.iter(h)) == want
        want_long = [(to_letter[e], v) for e, v in B.iter_long(hb)]
        assert list(A.iter_long(h)) == want_long
        if len(h) > 3:
            a, b = 1, len(h) - 1
            sub = h[a:b].encode("utf-8")
            skip = len(h[:a].encode("utf-8"))
            assert list(A.iter(h, a, b)) == [(to_letter[e + skip], v) for e, v in B.iter(sub)]


def test_dumps_written_by_the_str_flavour_load_in_the_unicode_reference(U, tmp_path):
    """pickles and save files written here ARE dumps of the reference's unicode build (4-byte letters,
    multi-byte letters decoded on the way out): the live reference (subprocess) reads them and
    enumerates / searches the same"""
    import subprocess
    ref_dir = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "unicode")
    if not os.path.isdir(ref_dir):
        pytest.skip("oracle/_ref/unicode (the reference's unicode build) is not built")
    jobs = []
    for i, case in enumerate(CASES[:9]):
        A = build(U, case, finalise=True)
        pkl, sav = str(tmp_path / ("c%d.pkl" % i)), str(tmp_path / ("c%d.sav" % i))
        open(pkl, "wb").write(pickle.dumps(A, protocol=2))
        if case["store"] == U.STORE_ANY:
            A.save(sav, lambda v: pickle.dumps(v, protocol=2))
        else:
            A.save(sav)
        jobs.append({"pkl": pkl, "sav": sav, "hays": [s["hay"] for s in case["searches"]]})
    code = r'''
import sys, json, pickle
sys.path.insert(0, %r)
import ahocorasick as R
assert R.unicode == 1 and "_ref" in R.__file__
out = []
for job in json.load(sys.stdin):
    res = []
    for x in (pickle.load(open(job["pkl"], "rb")), R.load(job["sav"], pickle.loads)):
        res.append({"keys": list(x.keys()), "values": list(x.values()), "kind": x.kind,
                    "iter": [[list(m) for m in x.iter(h)] for h in job["hays"]],
                    "iter_long": [[list(m) for m in x.iter_long(h)] for h in job["hays"]]})
    out.append(res)
print(json.dumps(out))
''' % ref_dir
    r = subprocess.run([sys.executable, "-c", code], input=json.dumps(jobs), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    for case, res2 in zip(CASES[:9], json.loads(r.stdout)):
        for res in res2:
            assert res["kind"] == U.AHOCORASICK
            assert res["keys"] == case["iter_keys"] and res["values"] == case["enum"][0][1]
            assert res["iter"] == [s["iter"] for s in case["searches"]]
            assert res["iter_long"] == [s["iter_long"] for s in case["searches"]]
