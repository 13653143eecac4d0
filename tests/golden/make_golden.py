#!/usr/bin/env python3
"""Generate tests/golden/ref_random.json by running the REFERENCE ITSELF
(WojciechMula/pyahocorasick v2.2.0, bytes build, compiled by oracle/Makefile from
/root/reference into oracle/_ref/).  Run in the build container (the GPU box has no
/root/reference and only ever reads the committed JSON):

    make -C oracle && python tests/golden/make_golden.py

Seeded, deterministic.  Every expected list below is `list(A.iter(...))`,
`list(A.iter_long(...))`, a find_all() callback trace or an iter().set() chunk trace of
the reference; nothing is computed by this repository's code.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

ref = orc.load_reference()
if ref is None:
    sys.exit("oracle/_ref is missing: run `make -C oracle` in a container that has /root/reference")
assert ref.unicode == 0, "need the bytes build"

rng = random.Random(20260923)

ALPHABETS = {
    "ab": b"ab",
    "abc": b"abc",
    "dna": b"ACGT",
    "dnaN": b"ACGTN",
    "high": bytes([0x61, 0x80, 0xFF, 0x7F, 0x00]),
    "text": b"ehrs_ ",
    "wide": bytes(range(256)),
}


def rand_bytes(alpha, lo, hi):
    return bytes(rng.choice(alpha) for _ in range(rng.randint(lo, hi)))


def tolist(it):
    return [[int(i), int(v)] for i, v in it]


def build(store, keys, values):
    A = ref.Automaton(store)
    for k, v in zip(keys, values):
        if store == ref.STORE_LENGTH:
            A.add_word(k)
        elif v is None:
            A.add_word(k)                    # default: 1-based insertion ordinal (src/Automaton.c:238-242)
        else:
            A.add_word(k, v)
    A.make_automaton()
    return A


def trace_find_all(A, hay, *rng_args):
    out = []
    A.find_all(hay, lambda i, v: out.append([int(i), int(v)]), *rng_args)
    return out


def one_case(cid, alpha_name, n_keys, klo, khi, store_name, hlo, hhi, n_hay):
    alpha = ALPHABETS[alpha_name]
    keys = []
    seen = set()
    while len(keys) < n_keys:
        k = rand_bytes(alpha, klo, khi)
        if k in seen and rng.random() < 0.8:
            continue
        seen.add(k)
        keys.append(k)            # occasional duplicates on purpose: value is overwritten
    store = {"ints": ref.STORE_INTS, "ints_default": ref.STORE_INTS, "length": ref.STORE_LENGTH,
             "any": ref.STORE_ANY}[store_name]
    if store_name == "ints":
        pool = [0, 1, -1, 7, 2**31 - 1, -2**31, 2**31, 2**32 + 5, 2**40 + 5, -3, 123456789]
        values = [rng.choice(pool) if rng.random() < 0.5 else rng.randint(-10**6, 10**6) for _ in keys]
    elif store_name == "any":
        values = list(range(len(keys)))   # python ints as the stored objects
    else:
        values = [None] * len(keys)
    A = build(store, keys, values)
    hays = []
    for _ in range(n_hay):
        if rng.random() < 0.3 and keys:
            # plant keys so that deep states and overlaps actually occur
            parts = []
            while sum(map(len, parts)) < hhi // 2:
                parts.append(rng.choice(keys) if rng.random() < 0.6 else rand_bytes(alpha, 0, 3))
            hay = b"".join(parts)[:hhi]
        else:
            hay = rand_bytes(alpha, hlo, hhi)
        ent = {"hay_hex": hay.hex(), "iter": tolist(A.iter(hay)), "iter_long": tolist(A.iter_long(hay)),
               "find_all": trace_find_all(A, hay)}
        if len(hay) >= 2:
            s = rng.randint(0, len(hay) - 1)
            e = rng.randint(s, len(hay))
            ent["slice"] = {"start": s, "end": e, "iter": tolist(A.iter(hay, s, e)),
                            "iter_long": tolist(A.iter_long(hay, s, e)),
                            "find_all": trace_find_all(A, hay, s, e)}
        hays.append(ent)
    # chunked streaming through set() (src/AutomatonSearchIter.c:303-368)
    whole = rand_bytes(alpha, hhi, 2 * hhi + 2)
    cuts = sorted(rng.sample(range(len(whole) + 1), min(3, len(whole))))
    parts = [whole[a:b] for a, b in zip([0] + cuts, cuts + [len(whole)])]
    it = A.iter(b"")
    trace = []
    for p in parts:
        it.set(p)
        trace.append(tolist(it))
    return {"id": cid, "alphabet": alpha_name, "store": store_name, "keys_hex": [k.hex() for k in keys],
            "values": values, "hays": hays,
            "chunks": {"parts_hex": [p.hex() for p in parts], "iter_set": trace,
                       "whole_iter": tolist(A.iter(whole))}}


def main():
    cases = []
    cid = 0
    plan = [
        # alphabet, n_keys, klo, khi, store, hlo, hhi, n_hay, repeats
        ("ab", 4, 1, 4, "ints", 0, 24, 4, 6),
        ("ab", 12, 1, 6, "ints_default", 0, 40, 4, 5),
        ("abc", 10, 1, 5, "length", 0, 40, 4, 5),
        ("dna", 30, 2, 8, "ints", 10, 80, 4, 5),
        ("dnaN", 40, 3, 10, "any", 10, 80, 3, 4),
        ("high", 8, 1, 4, "ints", 0, 30, 4, 6),
        ("text", 6, 1, 5, "length", 0, 50, 4, 5),
        ("wide", 40, 1, 4, "ints", 0, 60, 3, 4),
    ]
    for alpha, nk, klo, khi, store, hlo, hhi, nh, rep in plan:
        for _ in range(rep):
            cases.append(one_case("rand%03d" % cid, alpha, nk, klo, khi, store, hlo, hhi, nh))
            cid += 1

    # hand-picked behaviours probed in SURVEY §8(c) — expected values come from the reference run here
    special = []

    def sp(name, store, kv, hay, note):
        A = ref.Automaton(store)
        for k, v in kv:
            if store == ref.STORE_LENGTH or v is None:
                A.add_word(k)
            else:
                A.add_word(k, v)
        A.make_automaton()
        special.append({"id": name, "note": note,
                        "store": {ref.STORE_INTS: "ints", ref.STORE_LENGTH: "length"}[store],
                        "keys_hex": [k.hex() for k, _ in kv], "values": [v for _, v in kv],
                        "hay_hex": hay.hex(), "iter": tolist(A.iter(hay)), "iter_long": tolist(A.iter_long(hay))})

    I, L = ref.STORE_INTS, ref.STORE_LENGTH
    sp("int_truncation", I, [(b"ab", 2**40 + 5), (b"b", -3)], b"xabx", "values truncated to C int by Py_BuildValue('ii')")
    sp("high_bytes", I, [(b"\xff\x80a", 7), (b"\x80", 9)], b"zz\xff\x80a\x80", "bytes >= 0x80 (sign-extended letters)")
    sp("she_length", L, [(k, None) for k in (b"he", b"her", b"hers", b"she")], b"_sherhershe_", "STORE_LENGTH")
    sp("she_default_ints", I, [(k, None) for k in (b"he", b"her", b"hers", b"she")], b"_sherhershe_", "default = insertion ordinal")
    for name, ks, hay in [
        ("long_q1", [b"abcd", b"bc", b"c"], b"abcx"),
        ("long_q2", [b"abcd", b"bc"], b"abcabcd"),
        ("long_q3", [b"ab", b"abcde", b"cd"], b"abcdx"),
        ("long_q4", [b"a", b"ab", b"abc", b"bcd"], b"abcd"),
        ("long_q5", [b"xay", b"a"], b"xaxay"),
        ("long_q6", [b"aaa", b"a"], b"aaaaa"),
        ("nested_suffixes", [b"a" * n for n in range(1, 41)], b"a" * 50),
        ("empty_haystack", [b"a"], b""),
        ("no_match", [b"abc"], b"xyzxyz"),
    ]:
        sp(name, I, [(k, i + 1) for i, k in enumerate(ks)], hay, "iter_long quirk / edge case")

    out = {"reference": "WojciechMula/pyahocorasick v2.2.0 bytes build (oracle/_ref)",
           "generator": "tests/golden/make_golden.py seed 20260923", "cases": cases, "special": special}
    path = os.path.join(HERE, "ref_random.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "cases,", len(special), "special")


if __name__ == "__main__":
    main()
