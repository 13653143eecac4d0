"""SURVEY §8f N3 — the reference's persistence formats: pickles (`Automaton.__reduce__`,
src/Automaton_pickle.c) and save files (`Automaton.save` / `ahocorasick.load`,
src/custompickle/) of the bytes build load into this engine and vice versa.

Fixtures in tests/golden/ref_persist.json were produced by running the reference
(tests/golden/make_persist_golden.py).  Both hosts are covered: the ctypes mirror
(pyahocorasick_amd.Automaton) and the CPython extension (dropin/ahocorasick).
CPU tests check structure (every key, value, kind, node count, byte-identical re-dump);
the GPU test checks that loaded automata search like the reference did.
"""
import json
import os
import pickle
import struct
import subprocess
import sys

import pytest

import pyahocorasick_amd as acx

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CASES = json.load(open(os.path.join(HERE, "golden", "ref_persist.json")))["cases"]
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
HAVE_REF = any(f.startswith("ahocorasick") and f.endswith(".so") for f in (os.listdir(REF_DIR) if os.path.isdir(REF_DIR) else []))


def dropin():
    """the CPython extension under its real name (pickle looks the class up as ahocorasick.Automaton);
    other tests may have registered the reference module under that name"""
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
    assert ahocorasick.__file__.startswith(DROPIN_DIR)
    return ahocorasick


HOSTS = [("mirror", lambda: acx), ("dropin", dropin)]


def reduce_args(case):
    r = case["reduce"]
    if r is None:
        return ()
    values = None if r["values_pickle"] is None else pickle.loads(bytes.fromhex(r["values_pickle"]))
    return ([bytes.fromhex(c) for c in r["chunks"]], r["kind"], r["store"], r["key_type"], r["count"], r["longest_word"], values)


def check_content(A, case):
    assert len(A) == case["count"] and A.kind == case["kind"] and A.store == case["store"]
    for k, v in zip(case["keys"], case["values_json"]):
        k = bytes.fromhex(k)
        assert A.exists(k) and A.get(k) == v
        assert not A.exists(k + b"\x01\x02")


def records(chunk):
    """parse one pickle chunk into (output, fail, n, eow, [(letter, child)]) — padding dropped"""
    n_nodes, = struct.unpack_from("<q", chunk, 0)
    at, out = 8, []
    for _ in range(n_nodes):
        output, fail, n, eow = struct.unpack_from("<QQIB", chunk, at)
        at += 24
        pairs = [struct.unpack_from("<HQ", chunk, at + 10 * j) for j in range(n)]
        at += 10 * n
        out.append((output if eow else 0, fail, n, eow, pairs))     # (a removed key leaves its old value behind)
    assert at == len(chunk)
    return out


def save_records(data, store_any):
    """parse a save file into header fields + per-node records with addresses renamed 1..n"""
    assert data[:16] == b"pyahocorasick002" and data[-16:] == b"pyahocorasick002"
    kind, store, key_type = struct.unpack_from("<iii", data, 16)
    words, = struct.unpack_from("<Q", data, 32)
    longest, = struct.unpack_from("<i", data, 40)
    n_nodes, = struct.unpack_from("<Q", data, len(data) - 24)
    at, raw = 48, []
    for _ in range(n_nodes):
        addr, output, fail, n, eow = struct.unpack_from("<QQQIB", data, at)
        at += 32
        pairs = [struct.unpack_from("<HQ", data, at + 10 * j) for j in range(n)]
        at += 10 * n
        payload = b""
        if store_any and eow:
            payload = data[at:at + output]
            at += output
        raw.append((addr, output, fail, n, eow, pairs, payload))
    assert at == len(data) - 24
    name = {r[0]: i + 1 for i, r in enumerate(raw)}
    name[0] = 0
    nodes = [(r[1] if r[4] else 0, name[r[2]], r[3], r[4], [(l, name[c]) for l, c in r[5]], r[6]) for r in raw]
    return (kind, store, key_type, words, longest), nodes


@pytest.mark.parametrize("host", [h[0] for h in HOSTS])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_pickle_payload_loads(case, host):
    mod = dict(HOSTS)[host]()
    A = mod.Automaton(*reduce_args(case))
    check_content(A, case)
    if case["reduce"] is None:
        assert A.__reduce__() == (mod.Automaton, ())
        return
    # and the payload this engine writes is the reference's, record for record (ids, fail links,
    # child order, values), padding bytes aside
    cls, args = A.__reduce__()
    assert cls is mod.Automaton and args[1:6] == reduce_args(case)[1:6]
    mine = [r for c in args[0] for r in records(c)]
    theirs = [r for c in reduce_args(case)[0] for r in records(c)]
    if case["kind"] != acx.AHOCORASICK:                   # a plain trie carries no fail links
        assert all(r[1] == 0 for r in mine)
    assert mine == theirs
    assert args[6] == reduce_args(case)[6]
    B = pickle.loads(pickle.dumps(A))                     # full pickle round trip
    check_content(B, case)


@pytest.mark.parametrize("host", [h[0] for h in HOSTS])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_save_file_loads(case, host, tmp_path):
    mod = dict(HOSTS)[host]()
    path = str(tmp_path / "ref.sav")
    open(path, "wb").write(bytes.fromhex(case["savefile"]))
    A = mod.load(path, pickle.loads)
    check_content(A, case)
    out = str(tmp_path / "mine.sav")
    if case["store"] == acx.STORE_ANY:
        A.save(out, lambda v: pickle.dumps(v, protocol=2))
    else:
        A.save(out)
    any_ = case["store"] == acx.STORE_ANY
    assert save_records(open(out, "rb").read(), any_) == save_records(bytes.fromhex(case["savefile"]), any_)
    check_content(mod.load(out, pickle.loads), case)


def test_save_load_argument_rules(tmp_path):
    """src/custompickle/pyhelpers.c:4-59"""
    for _, get in HOSTS:
        mod = get()
        A = mod.Automaton(mod.STORE_INTS)
        A.add_word(b"a", 1)
        with pytest.raises(ValueError, match="exactly one argument"):
            A.save(str(tmp_path / "x"), pickle.dumps)
        with pytest.raises(TypeError, match="must be a string"):
            A.save(b"bytes-path")
        B = mod.Automaton()
        B.add_word(b"a", object)
        with pytest.raises(ValueError, match="exactly two arguments"):
            B.save(str(tmp_path / "x"))
        with pytest.raises(TypeError, match="callable"):
            B.save(str(tmp_path / "x"), 42)
        with pytest.raises(TypeError, match="serializer must return bytes"):
            B.save(str(tmp_path / "x"), lambda v: "text")
        with pytest.raises(ValueError, match="exactly two arguments"):
            mod.load(str(tmp_path / "x"))
        with pytest.raises(IOError):
            mod.load(str(tmp_path / "does-not-exist"), pickle.loads)


def test_malformed_dumps_are_rejected(tmp_path):
    case = next(c for c in CASES if c["name"] == "ints_ushers")
    chunks, kind, store, key_type, count, longest, values = reduce_args(case)
    good = chunks[0]
    bad_payloads = {
        "truncated": good[:-5],
        "zero nodes": struct.pack("<q", 0) + good[8:],
        "link out of range": good[:8 + 24 + 2] + struct.pack("<Q", 99) + good[8 + 24 + 10:],
        "link to itself": good[:8 + 24 + 2] + struct.pack("<Q", 1) + good[8 + 24 + 10:],
        "two parents": good[:8 + 24 + 12] + struct.pack("<Q", 2) + good[8 + 24 + 20:],
        "letter above 255": good[:8 + 24] + struct.pack("<H", 0x1234) + good[8 + 24 + 2:],
    }
    sav = bytes.fromhex(case["savefile"])
    bad_files = {"bad magic": b"X" + sav[1:], "truncated": sav[:-30], "bad footer": sav[:-1] + b"X",
                 "stray bytes": sav[:-24] + b"\0" * 7 + sav[-24:]}
    for _, get in HOSTS:
        mod = get()
        for what, payload in bad_payloads.items():
            with pytest.raises(ValueError):
                mod.Automaton([payload], kind, store, key_type, count, longest, values)
        with pytest.raises(ValueError):
            mod.Automaton([good, "not bytes"], kind, store, key_type, count, longest, values)
        with pytest.raises(TypeError):
            mod.Automaton(tuple(chunks), kind, store, key_type, count, longest, values)
        for what, data in bad_files.items():
            p = str(tmp_path / "bad.sav")
            open(p, "wb").write(data)
            with pytest.raises(ValueError):
                mod.load(p, pickle.loads)


@pytest.mark.skipif(not HAVE_REF, reason="oracle/_ref (the reference itself) is not built")
def test_dumps_written_here_load_in_the_reference(tmp_path):
    """the other direction, against the live reference in a subprocess (both modules are called
    `ahocorasick`): pickles and save files written by the drop-in extension"""
    D = dropin()
    jobs = []
    # an automaton built here key by key: its dump order (node creation order) is not the pre-order
    # the reference writes, which its loader must not care about
    local_keys = [b"b", b"a", b"ba", b"ab", b"bab", b"c", b"abc", b"cab", b"aa"]
    L = D.Automaton(D.STORE_INTS)
    for i, k in enumerate(local_keys):
        L.add_word(k, i)
    L.make_automaton()
    lp, ls = str(tmp_path / "local.pkl"), str(tmp_path / "local.sav")
    open(lp, "wb").write(pickle.dumps(L, protocol=2))
    L.save(ls)
    jobs.append({"name": "__local__", "pkl": lp, "sav": ls, "haystacks": [b"abcabbabaacab".hex()]})
    for case in CASES:
        if case["kind"] != acx.AHOCORASICK:
            continue
        A = D.Automaton(*reduce_args(case))
        pkl, sav = str(tmp_path / (case["name"] + ".pkl")), str(tmp_path / (case["name"] + ".sav"))
        open(pkl, "wb").write(pickle.dumps(A, protocol=2))
        if case["store"] == acx.STORE_ANY:
            A.save(sav, lambda v: pickle.dumps(v, protocol=2))
        else:
            A.save(sav)
        jobs.append({"name": case["name"], "pkl": pkl, "sav": sav, "haystacks": case["haystacks"]})
    code = r'''
import sys, json, pickle
sys.path.insert(0, %r)
import ahocorasick as R
assert "_ref" in R.__file__
out = {}
for job in json.load(sys.stdin):
    a = pickle.load(open(job["pkl"], "rb"))
    b = R.load(job["sav"], pickle.loads)
    res = []
    for x in (a, b):
        hs = [bytes.fromhex(h) for h in job["haystacks"]]
        res.append({"keys": [k.hex() for k in sorted(x.keys())], "values": [x.get(k) for k in sorted(x.keys())],
                    "iter": [[list(m) for m in x.iter(h)] for h in hs], "iter_long": [[list(m) for m in x.iter_long(h)] for h in hs],
                    "stats": {k: x.get_stats()[k] for k in ("nodes_count", "words_count", "longest_word")}})
    out[job["name"]] = res
print(json.dumps(out))
''' % REF_DIR
    r = subprocess.run([sys.executable, "-c", code], input=json.dumps(jobs), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout)
    for res in got["__local__"]:
        assert sorted(res["keys"]) == sorted(k.hex() for k in local_keys) and res["stats"]["words_count"] == len(local_keys)
        want = []
        hay = b"abcabbabaacab"
        for end in range(len(hay)):                     # every key ending here, longest first
            for k in sorted((k for k in local_keys if hay[:end + 1].endswith(k)), key=len, reverse=True):
                want.append([end, local_keys.index(k)])
        assert res["iter"] == [want]
    for case in CASES:
        if case["kind"] != acx.AHOCORASICK:
            continue
        for res in got[case["name"]]:
            assert res["keys"] == case["keys"] and res["values"] == case["values_json"]
            assert res["iter"] == case["iter"] and res["iter_long"] == case["iter_long"]
            assert res["stats"] == case["stats"]


@pytest.mark.gpu
@pytest.mark.parametrize("host", [h[0] for h in HOSTS])
def test_loaded_automata_search_like_the_reference(host, tmp_path):
    mod = dict(HOSTS)[host]()
    for case in CASES:
        if case["kind"] != acx.AHOCORASICK:
            continue
        path = str(tmp_path / "ref.sav")
        open(path, "wb").write(bytes.fromhex(case["savefile"]))
        for A in (mod.Automaton(*reduce_args(case)), mod.load(path, pickle.loads)):
            for h, it, il in zip(case["haystacks"], case["iter"], case["iter_long"]):
                h = bytes.fromhex(h)
                assert [list(m) for m in A.iter(h)] == it
                assert [list(m) for m in A.iter_long(h)] == il
