"""The walk over the HOST trie (csrc/acx_hostwalk.cpp, include/acx.h §4b): BASELINE.json's config 1 — "4-key he/her/hers/she
automaton, Automaton.iter() over a 1 KB ASCII haystack on CPU (plumbing, no GPU)" — processes without a device, and
haystacks that do not pay a launch.  CPU tests: the walk against
  (1) the reference's own known answer for config 1 (tests/test_unit.py:532-545) and the committed golden fixtures
      (reference test vectors, reference-generated cases, white-space fixtures — the same files the GPU suite checks the
      kernels with),
  (2) the pinned oracle on seeded random automata (small alphabets: long fail chains), contexts, white space, slices,
  (3) the reference itself (oracle/_ref, when built) for streams through set(),
through both host sides (the ctypes mirror and the CPython extension).  The GPU suite runs with the walk switched off
(tests/conftest.py): nothing here is a GPU parity claim."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

import pyahocorasick_amd as acx
from pyahocorasick_amd import _lib
from helpers import build_pair, expected_pairs, load_json
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VECTORS = load_json("ref_vectors.json")["vectors"]
RANDOM = load_json("ref_random.json")
WS = load_json("ref_ws.json")


@pytest.fixture(autouse=True)
def _walk_everything():
    """every search of this file is walked on the host (up to the walk's own 1 MiB cap), whatever the box has"""
    l = _lib.lib()
    l.acx_set_host_walk_bytes(1 << 20)
    before = l.acx_host_walk_calls()
    yield
    assert l.acx_host_walk_calls() > before, "this test never reached the host walk"


def _dropin_script(tmp_path, body):
    """run `body` in a fresh interpreter that imports the drop-in extension as `ahocorasick` (two extension modules of one name —
    the drop-in and the reference's own build under oracle/_ref — do not mix in one process: CPython caches single-phase
    extension modules by name), every search walked on the host; returns the JSON it prints last"""
    import json
    from pyahocorasick_amd.build import build_dropin, DROPIN_DIR
    build_dropin(verbose=False)
    script = tmp_path / "case.py"
    script.write_text("import json, sys, random, time\nimport ahocorasick\nassert ahocorasick.__file__.startswith(%r)\n" % DROPIN_DIR + body)
    env = dict(os.environ)
    env["PYTHONPATH"] = DROPIN_DIR + os.pathsep + ROOT
    env["ACX_HOST_WALK_BYTES"] = str(1 << 20)
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def _touch_walk():
    """(the fixture's counter: a test whose searches all run in a subprocess walks one string itself)"""
    A, _ = build_pair([b"x"])
    assert list(A.iter(b"x")) == [(0, 0)]


# ------------------------------------------------------------------ config 1
CONFIG1_KEYS = (b"he", b"her", b"hers", b"she")


def _config1_haystack():
    rnd = random.Random(1)
    words = [b"he", b"she", b"hers", b"her", b"ushers", b"the", b"quick", b"brown", b"fox", b"shell", b"x", b"heir"]
    out = bytearray()
    while len(out) < 1024:
        out += rnd.choice(words) + b" "
    return bytes(out[:1024])


_CONFIG1_BODY = """
hay = bytes.fromhex(%r)
D = ahocorasick.Automaton()
for w in (b"he", b"her", b"hers", b"she"):
    D.add_word(w, w.decode())
D.make_automaton()
fa = []
D.find_all(hay, lambda i, v: fa.append((i, v)))
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    for _ in range(200):
        for _m in D.iter(hay):
            pass
    best = min(best, (time.perf_counter() - t) / 200)
print(json.dumps({"known": list(D.iter(b"_sherhershe_")), "known_long": list(D.iter_long(b"_sherhershe_")),
                  "iter": list(D.iter(hay)), "iter_long": list(D.iter_long(hay)), "find_all": fa, "us": best * 1e6,
                  "walks": ahocorasick.host_walk_calls()}))
"""


def test_config1_known_answer_and_1kb_haystack(tmp_path):
    """reference tests/test_unit.py:532-545 (the he/her/hers/she automaton over "_sherhershe_"), then BASELINE config 1's 1 KB
    ASCII haystack: ctypes mirror == CPython extension (a fresh process without a GPU scan in it) == oracle; and the 1 KB
    iter() takes tens of microseconds (reference: 18-26 us for 33 matches; as a GPU scan: ~90 us) — the bound
    is generous, a loaded CI core, the point is the order of magnitude"""
    A = acx.Automaton()
    O = orc.Oracle()
    for i, w in enumerate(CONFIG1_KEYS):
        A.add_word(w, w.decode())
        O.add_word(w, i)
    A.make_automaton(); O.make_automaton()
    exp = [(3, "she"), (3, "he"), (4, "her"), (6, "he"), (7, "her"), (8, "hers"), (10, "she"), (10, "he")]
    hay = _config1_haystack()
    want = [(i, CONFIG1_KEYS[v].decode()) for i, v in O.iter(hay)]
    want_long = [(i, CONFIG1_KEYS[v].decode()) for i, v in O.iter_long(hay)]
    assert len(want) > 100
    assert list(A.iter(b"_sherhershe_")) == exp and list(A.iter_long(b"_sherhershe_")) == [(3, "she"), (8, "hers"), (10, "he")]
    assert list(A.iter(hay)) == want and list(A.iter_long(hay)) == want_long
    r = _dropin_script(tmp_path, _CONFIG1_BODY % hay.hex())
    tup = lambda xs: [tuple(x) for x in xs]
    assert tup(r["known"]) == exp and tup(r["known_long"]) == [(3, "she"), (8, "hers"), (10, "he")]
    assert tup(r["iter"]) == want and tup(r["find_all"]) == want and tup(r["iter_long"]) == want_long
    assert r["walks"] >= 1000
    print("config 1: iter() over 1 KB through the drop-in, %d matches: %.1f us" % (len(want), r["us"]))
    assert r["us"] < 500


# ------------------------------------------------------------------ golden fixtures (the files of tests/test_gpu_parity.py)
def _case_values(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    if c["store"] in ("length", "ints_default"):
        return keys, None
    return keys, c["values"]


@pytest.mark.parametrize("v", VECTORS, ids=[v["id"] for v in VECTORS])
def test_reference_test_vectors(v):
    keys = [bytes.fromhex(k) for k in v["keys_hex"]]
    hay = bytes.fromhex(v["hay_hex"])
    A, _ = build_pair(keys)
    exp = expected_pairs(v["expected"])
    rng = [] if v["start"] is None else ([v["start"]] if v["end"] is None else [v["start"], v["end"]])
    if v["mode"] == "iter":
        got = list(A.iter(hay, *rng))
    elif v["mode"] == "iter_long":
        got = list(A.iter_long(hay, *rng))
    else:
        got = []
        A.find_all(hay, lambda i, val: got.append((i, val)), *rng)
    assert got == exp, v["source"]


@pytest.mark.parametrize("c", RANDOM["cases"], ids=[c["id"] for c in RANDOM["cases"]])
def test_reference_generated_fixtures(c):
    keys, values = _case_values(c)
    A, _ = build_pair(keys, values, c["store"])
    hays = [bytes.fromhex(h["hay_hex"]) for h in c["hays"]]
    assert A.iter_batch(hays) == [expected_pairs(h["iter"]) for h in c["hays"]]
    assert A.iter_batch(hays, long=True) == [expected_pairs(h["iter_long"]) for h in c["hays"]]
    for h, hay in zip(c["hays"], hays):
        if "slice" in h:
            s, e = h["slice"]["start"], h["slice"]["end"]
            assert list(A.iter(hay, s, e)) == expected_pairs(h["slice"]["iter"])
            if s < len(hay):
                assert list(A.iter_long(hay, s, e)) == expected_pairs(h["slice"]["iter_long"])
                got = []
                A.find_all(hay, lambda i, v: got.append((i, v)), s, e)
                assert got == expected_pairs(h["slice"]["find_all"])
    it = A.iter(b"")                                       # streaming through set(): context and shift carried across chunks
    for part_hex, exp in zip(c["chunks"]["parts_hex"], c["chunks"]["iter_set"]):
        it.set(bytes.fromhex(part_hex))
        assert list(it) == expected_pairs(exp)


@pytest.mark.parametrize("c", RANDOM["special"], ids=[c["id"] for c in RANDOM["special"]])
def test_reference_special_cases(c):
    keys = [bytes.fromhex(k) for k in c["keys_hex"]]
    A, _ = build_pair(keys, c["values"], c["store"])
    hay = bytes.fromhex(c["hay_hex"])
    assert list(A.iter(hay)) == expected_pairs(c["iter"])
    assert list(A.iter_long(hay)) == expected_pairs(c["iter_long"])


def test_white_space_fixtures_written_by_the_reference():
    """tests/golden/ref_ws.json: iter(..., ignore_white_space=True) as the reference answers it — whole strings, slices and
    streams through set()"""
    n = 0
    for c in WS["cases"]:
        keys = [bytes.fromhex(k) for k in c["keys_hex"]]
        A, _ = build_pair(keys)
        for h in c["hays"]:
            hay = bytes.fromhex(h["hay_hex"])
            assert list(A.iter(hay, ignore_white_space=True)) == expected_pairs(h["iter_ws"]), c["id"]
            assert list(A.iter(hay)) == expected_pairs(h["iter"]), c["id"]
            n += 1
            if "slice" in h:
                sl = h["slice"]
                assert list(A.iter(hay, sl["start"], sl["end"], ignore_white_space=True)) == expected_pairs(sl["iter_ws"]), c["id"]
        it = A.iter(b"", ignore_white_space=True)
        for part_hex, exp in zip(c["stream"]["chunks_hex"], c["stream"]["iter_ws_set"]):
            it.set(bytes.fromhex(part_hex))
            assert list(it) == expected_pairs(exp), c["id"]
    assert n >= 40


# ------------------------------------------------------------------ seeded differential against the oracle
def _random_case(rnd, alphabet, n_keys, max_len):
    keys = set()
    n_keys = min(n_keys, sum(len(set(alphabet)) ** k for k in range(1, max_len + 1)) // 2 + 1)
    while len(keys) < n_keys:
        keys.add(bytes(rnd.choice(alphabet) for _ in range(rnd.randint(1, max_len))))
    return sorted(keys)


@pytest.mark.parametrize("seed", range(12))
def test_random_automata_against_the_oracle(seed):
    rnd = random.Random(1000 + seed)
    alphabet = [b"ab", b"abc", b"ACGT", b"ab \t", bytes(range(256))][seed % 5]
    keys = _random_case(rnd, alphabet, rnd.randint(1, 60), rnd.randint(2, 9))
    A, O = build_pair(keys)
    hays = [bytes(rnd.choice(alphabet) for _ in range(rnd.choice([0, 1, 2, 7, 64, 300, 2500]))) for _ in range(24)]
    off = np.zeros(len(hays) + 1, dtype=np.int64)
    np.cumsum([len(h) for h in hays], out=off[1:])
    data = b"".join(hays)
    for mode in (acx.ACX_SCAN_ALL, acx.ACX_SCAN_LONG):
        res = A.scan_batch(data, off, mode, index_base=np.arange(len(hays), dtype=np.int32) * 7)
        mo, e, v = O.batch(data, off, mode)
        base = np.repeat(np.arange(len(hays), dtype=np.int32) * 7, np.diff(mo))
        assert np.array_equal(res.offsets, mo) and np.array_equal(res.end_index, e + base) and np.array_equal(res.value, v)
    for hay in hays[:8]:
        assert list(A.iter(hay, ignore_white_space=True)) == O.iter(hay, ignore_ws=True)
        if len(hay) > 3:
            s, e = rnd.randrange(len(hay)), rnd.randrange(len(hay) + 1)
            if s <= e:
                assert list(A.iter(hay, s, e)) == O.iter(hay, s, e)
    # a stream: the oracle walks the concatenation, the product continues from the context bytes
    whole = b"".join(hays[:6])
    it = A.iter(b"")
    got = []
    for h in hays[:6]:
        it.set(h)
        got += list(it)
    assert got == O.iter(whole)


_STREAMS_BODY = """
from oracle import orc
ref = orc.load_reference()
rnd = random.Random(77)
keys = set()
while len(keys) < 40:
    keys.add(bytes(rnd.choice(b"abc") for _ in range(rnd.randint(1, 6))))
keys = sorted(keys)
chunks = [bytes(rnd.choice(b"abc") for _ in range(rnd.choice([0, 1, 3, 10, 40, 200]))) for _ in range(30)]
out = {"have_ref": ref is not None, "chunks": [c.hex() for c in chunks], "keys": [k.hex() for k in keys]}
for name, m in (("dropin", ahocorasick), ("ref", ref)):
    if m is None:
        continue
    B = m.Automaton(m.STORE_INTS)
    for i, k in enumerate(keys):
        B.add_word(k, i)
    B.make_automaton()
    for long_mode in (False, True):
        it = (B.iter_long if long_mode else B.iter)(chunks[0])
        got = [list(it)]
        for k, c in enumerate(chunks[1:]):
            it.set(c, k % 11 == 10)
            # (iter_long only: after a set() in the middle of a position's outputs the reference's iter keeps draining them —
            #  the one documented difference, tests/test_gpu_parity.py::test_iterator_set_semantics_vs_reference)
            if long_mode and k % 5 == 3:                    # abandoned after two matches: set() continues from there
                part = []
                for x in it:
                    part.append(x)
                    if len(part) == 2:
                        break
                got.append(part)
            else:
                got.append(list(it))
        out[name + ("_long" if long_mode else "")] = got
out["walks"] = ahocorasick.host_walk_calls()
print(json.dumps(out))
"""


def test_streams_through_set_against_the_reference_itself(tmp_path):
    """iter().set() and iter_long().set() of the drop-in, chunk by chunk (resets in between, iter_long also abandoned
    half-way), against the reference's own extension (oracle/_ref) when it is built; iter also against the oracle's walk"""
    _touch_walk()
    r = _dropin_script(tmp_path, _STREAMS_BODY)
    assert r["walks"] > 0
    if r["have_ref"]:
        assert r["dropin"] == r["ref"] and r["dropin_long"] == r["ref_long"]
    assert sum(len(x) for x in r["dropin"]) > 100 and sum(len(x) for x in r["dropin_long"]) > 20
    # the oracle on the pieces between two resets (set(c, True) for k % 11 == 10 starts over)
    keys = [bytes.fromhex(k) for k in r["keys"]]
    chunks = [bytes.fromhex(c) for c in r["chunks"]]
    _, O = build_pair(keys)
    pieces, cur = [], [0]
    for k in range(1, len(chunks)):
        if (k - 1) % 11 == 10:
            pieces.append(cur); cur = []
        cur.append(k)
    pieces.append(cur)
    for piece in pieces:
        got = [tuple(x) for k in piece for x in r["dropin"][k]]
        assert got == O.iter(b"".join(chunks[k] for k in piece))


def test_a_carried_host_state_never_reaches_the_device():
    """iter_long's state across set() is -(arena node) - 1 on the host walk: scan_batch with such a state is walked on the host
    whatever the limit says, and the C entry point maps an unknown node to the root"""
    A, O = build_pair([b"abcd", b"bc", b"cde"])
    r1 = A.scan_batch(b"ab", [0, 2], acx.ACX_SCAN_LONG)
    assert r1.final_state is not None and int(r1.final_state[0]) < 0
    _lib.lib().acx_set_host_walk_bytes(0)                      # nothing but carried states and empty batches go to the host now
    r2 = A.scan_batch(b"cd", [0, 2], acx.ACX_SCAN_LONG, init_state=r1.final_state, index_base=[2])
    assert list(zip(r2.end_index.tolist(), r2.value.tolist())) == O.iter_long(b"abcd")
    r3 = A.scan_batch(b"cd", [0, 2], acx.ACX_SCAN_LONG, init_state=[-10**6])
    assert list(zip(r3.end_index.tolist(), r3.value.tolist())) == O.iter_long(b"cd")


def test_a_carried_host_state_takes_a_chunk_beyond_the_walks_cap(tmp_path):
    """iter_long(small chunk) leaves a NEGATIVE state when the chunk ends inside a key; set() with a chunk beyond ACX_HOSTWALK_MAX_BYTES
    then has to go on on the host walk (the state means nothing to a device image) instead of being refused — through the ctypes
    mirror's scan_batch and through the extension's iter_long().set()"""
    keys = [b"abcd", b"bc", b"cdeab", b"dea"]
    A, O = build_pair(keys)
    rnd = random.Random(7)
    big = bytes(rnd.choice(b"abcde") for _ in range((1 << 20) + 4097))
    r1 = A.scan_batch(b"xxab", [0, 4], acx.ACX_SCAN_LONG)
    assert int(r1.final_state[0]) < 0                          # "ab": inside "abcd"
    _lib.lib().acx_set_host_walk_bytes(2048)                    # (the default limit of a process with a device)
    r2 = A.scan_batch(big, [0, len(big)], acx.ACX_SCAN_LONG, init_state=r1.final_state, index_base=[4])
    want = O.iter_long(b"xxab" + big)
    got = list(zip(r1.end_index.tolist(), r1.value.tolist())) + list(zip(r2.end_index.tolist(), r2.value.tolist()))
    assert got == want and len(want) > 1000
    _lib.lib().acx_set_host_walk_bytes(1 << 20)
    body = """
import random
rnd = random.Random(7)
big = bytes(rnd.choice(b"abcde") for _ in range((1 << 20) + 4097))
D = ahocorasick.Automaton()
for i, w in enumerate((b"abcd", b"bc", b"cdeab", b"dea")):
    D.add_word(w, i)
D.make_automaton()
ahocorasick.set_host_walk_bytes(2048)
it = D.iter_long(b"xxab")
got = list(it)
it.set(big)
got += list(it)
print(json.dumps({"got": got, "walks": ahocorasick.host_walk_calls()}))
"""
    r = _dropin_script(tmp_path, body)
    assert [tuple(x) for x in r["got"]] == want and r["walks"] >= 2


# ------------------------------------------------------------------ the reference's own test-suite, on the CPU
@pytest.mark.parametrize("flavour", ["bytes", "unicode"])
def test_reference_suite_against_dropin_on_the_host_walk(flavour):
    """tests/golden/ref_suite (the reference's tests, verbatim) against the drop-in with every search walked on the host:
    exactly the counts the reference's own builds give (143 / 9, 147 / 7).  tests/test_gpu_ref_suite.py runs the same files
    with the walk switched off: every search a GPU scan."""
    import re
    from pyahocorasick_amd.build import build_dropin
    build_dropin(verbose=False, unicode=(flavour == "unicode"))
    _lib.lib().acx_trie_scan_host                              # (the fixture's counter: this process walks nothing itself)
    A, _ = build_pair([b"x"])
    list(A.iter(b"x"))
    suite = os.path.join(ROOT, "tests", "golden", "ref_suite")
    mod_dir = os.path.join(ROOT, "dropin") if flavour == "bytes" else os.path.join(ROOT, "dropin", "unicode")
    env = dict(os.environ)
    env["PYTHONPATH"] = mod_dir + os.pathsep + suite
    env["ACX_HOST_WALK_BYTES"] = str(1 << 20)
    cmd = [sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider", "--rootdir", suite, "-c", os.devnull, "-rf", suite]
    p = subprocess.run(cmd, env=env, cwd=suite, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    summary = out.strip().splitlines()[-1] if out.strip() else ""
    assert "failed" not in summary, out[-3000:]
    m = re.search(r"(\d+) passed", summary)
    assert m and int(m.group(1)) == (143 if flavour == "bytes" else 147), summary
