#!/usr/bin/env python3
"""Generate tests/golden/ref_ws.json by running the REFERENCE ITSELF (bytes build, oracle/_ref/): what
Automaton.iter(string, start, end, ignore_white_space=True) returns — whole strings, slices, and streams continued with
iter().set(chunk) — on texts whose keys are broken up by white space (src/AutomatonSearchIter.c:269-274).  Run in the
build container:

    make -C oracle && python tests/golden/make_ws_golden.py

Seeded, deterministic; nothing below is computed by this repository's code.
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
rng = random.Random(20260924)
WS = b" \t\n\x0b\x0c\r"


def tolist(it):
    return [[int(i), int(v)] for i, v in it]


def spaced(word):
    """the word with white space between some of its letters"""
    out = bytearray()
    for b in word:
        out.append(b)
        if rng.random() < 0.5:
            out += bytes(rng.choice(WS) for _ in range(rng.randint(1, 3)))
    return bytes(out)


cases = []
for cid, alpha, nkeys, kmax in (("words", b"ehrsu", 12, 5), ("dna", b"ACGT", 60, 9), ("bytes", bytes(range(256)), 80, 4),
                                ("ws_in_keys", b"ab \t", 20, 4)):
    keys = sorted({bytes(rng.choice(alpha) for _ in range(rng.randint(1, kmax))) for _ in range(nkeys)})
    A = ref.Automaton(ref.STORE_INTS)
    for i, k in enumerate(keys):
        A.add_word(k, i)
    A.make_automaton()
    hays = []
    for _ in range(12):
        parts = []
        for _ in range(rng.randint(1, 12)):
            r = rng.random()
            if r < 0.5:
                parts.append(spaced(rng.choice(keys)))
            elif r < 0.8:
                parts.append(bytes(rng.choice(alpha) for _ in range(rng.randint(0, 6))))
            else:
                parts.append(bytes(rng.choice(WS) for _ in range(rng.randint(1, 4))))
        hay = b"".join(parts)
        e = {"hay_hex": hay.hex(), "iter_ws": tolist(A.iter(hay, ignore_white_space=True)), "iter": tolist(A.iter(hay))}
        if len(hay) > 4:
            s = rng.randint(0, len(hay) // 2)
            t = rng.randint(s, len(hay))
            e["slice"] = {"start": s, "end": t, "iter_ws": tolist(A.iter(hay, s, t, ignore_white_space=True))}
        hays.append(e)
    # a stream: one text cut at random places, continued with set()
    text = b"".join(spaced(rng.choice(keys)) + bytes(rng.choice(alpha + WS) for _ in range(rng.randint(0, 3))) for _ in range(30))
    cuts = sorted({0, len(text)} | {rng.randint(0, len(text)) for _ in range(6)})
    chunks = [text[a:b] for a, b in zip(cuts, cuts[1:])]
    it = A.iter(b"", ignore_white_space=True)
    trace = []
    for c in chunks:
        it.set(c)
        trace.append(tolist(it))
    cases.append({"id": cid, "keys_hex": [k.hex() for k in keys], "hays": hays,
                  "stream": {"chunks_hex": [c.hex() for c in chunks], "iter_ws_set": trace, "whole_iter_ws": tolist(A.iter(text, ignore_white_space=True))}})

out = {"reference": "WojciechMula/pyahocorasick v2.2.0, bytes build (oracle/_ref)", "generator": "tests/golden/make_ws_golden.py", "cases": cases}
with open(os.path.join(HERE, "ref_ws.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote ref_ws.json:", sum(len(c["hays"]) for c in cases), "haystacks,", len(cases), "streams")
