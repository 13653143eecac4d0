#!/usr/bin/env python3
"""Known-answer vectors TRANSCRIBED from the reference's own tests and docs
(WojciechMula/pyahocorasick v2.2.0; paths relative to the reference root).  Pure data:
does not need the reference to run.  Writes tests/golden/ref_vectors.json.

Each vector: keys (inserted in this order; value = the key's 0-based ordinal), a haystack,
the call (iter / iter_long / find_all, optional start/end) and the expected list of
(end_index, key) — keys are written as text here and stored as hex in the JSON (the
bytes build sees UTF-8 bytes, reference tests/pytestingutils.py:15-18 `conv`).
"""
import json
import os

V = []


def vec(id, source, keys, hay, expected, mode="iter", start=None, end=None, note=""):
    enc = lambda s: s if isinstance(s, bytes) else s.encode("utf-8")
    keys_b = [enc(k) for k in keys]
    exp = []
    for idx, k in expected:
        kb = enc(k)
        # a key added twice keeps its node; the value is the LAST ordinal written
        ordinal = max(i for i, kk in enumerate(keys_b) if kk == kb)
        exp.append([idx, ordinal])
    V.append(dict(id=id, source=source, mode=mode, start=start, end=end, note=note,
                  keys_hex=[k.hex() for k in keys_b], hay_hex=enc(hay).hex(), expected=exp))


SHE = ["he", "her", "hers", "she"]
SHE_POS = [(3, "she"), (3, "he"), (4, "her"), (6, "he"), (7, "her"), (8, "hers"), (10, "she"), (10, "he")]

vec("unit_iter2", "tests/test_unit.py:532-545,713-721", SHE, "_sherhershe_", SHE_POS)
vec("unit_find_all2", "tests/test_unit.py:606-618", SHE, "_sherhershe_", SHE_POS, mode="find_all")
vec("unit_iter3_slice_4_9", "tests/test_unit.py:723-737", SHE, "_sherhershe_",
    [(6, "he"), (7, "her"), (8, "hers")], start=4, end=9,
    note="iter(s[4:9]) shifted by 4: 'rhers' -> he@2, her@3, hers@4")
vec("unit_find_all3_slice_4_9", "tests/test_unit.py:620-638", SHE, "_sherhershe_",
    [(6, "he"), (7, "her"), (8, "hers")], mode="find_all", start=4, end=9)
vec("basic_iter_2_8", "tests/test_basic.py:18-32",
    b"he e hers his she hi him man he".split(), b"he rshershidamanza ",
    [(6, "she"), (6, "he"), (6, "e")], start=2, end=8,
    note="order within a position: state first, then its fail chain (longest first)")
vec("basic_find_all_2_11", "tests/test_basic.py:34-50",
    b"he e hers his she hi him man he".split(), b"he rshershidamanza ",
    [(6, "she"), (6, "he"), (6, "e"), (8, "hers"), (10, "hi")], mode="find_all", start=2, end=11)
vec("issue8_utf8_bytes", "tests/test_issue_8.py:31-37,73-83",
    ["wąż", "mąż", "żółć", "aż", "waży"], "wyważyć", [(5, "aż"), (6, "waży")],
    note="UTF-8 byte offsets in the bytes build")
vec("issue53_utf8_offset", "tests/test_issue_53.py:32-45", ["wounded"],
    "Winning \U0001F629 so gutted, can't do anything for 4 weeks... Myth. #wounded", [(70, "wounded")])
vec("issue53_ascii", "tests/test_issue_53.py:42-45", ["wounded"],
    "Winning so gutted, can't do anything for 4 weeks... Myth. #wounded", [(65, "wounded")])
vec("issue10_case1", "tests/test_issue_10.py:15-25", ["S"], "SSS", [(0, "S"), (1, "S"), (2, "S")], start=0, end=3)
vec("issue10_case2", "tests/test_issue_10.py:28-38", ["S"], "SSS", [(0, "S"), (1, "S")], start=0, end=2)
vec("unit_bug_search", "tests/test_unit.py:1102-1115", ["GT-C3303", "SAMSUNG-GT-C3303K/"],
    "SAMSUNG-GT-C3303i/1.0 NetFront/3.5 Profile/MIDP-2.0 Configuration/CLDC-1.1", [(15, "GT-C3303")])
vec("unit_iter_long", "tests/test_unit.py:1493-1504", ["he", "here", "her"], "he here her",
    [(1, "he"), (6, "here"), (10, "her")], mode="iter_long")
vec("issue133_iter_long_1", "tests/test_issue_133.py:15-24", ["b", "abc"], "abb", [(1, "b"), (2, "b")], mode="iter_long")
vec("issue133_iter_long_2", "tests/test_issue_133.py:27-37", ["b", "c", "abd"], "abc", [(1, "b"), (2, "c")], mode="iter_long")
vec("issue133_iter_long_multibyte", "tests/test_issue_133.py:40-53", ["知识产权", "国家知识产权局"], "国家知识产权",
    [(17, "知识产权")], mode="iter_long")
vec("doc_iter_long", "docs/automaton_iter_long.rst:30-46", ["he", "her", "here"], "he here her",
    [(1, "he"), (6, "here"), (10, "her")], mode="iter_long")
vec("doc_iter_vs_iter_long", "docs/automaton_iter_long.rst:47-48", ["he", "her", "here"], "he here her",
    [(1, "he"), (4, "he"), (5, "her"), (6, "here"), (9, "he"), (10, "her")])

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_vectors.json")
    with open(out, "w") as f:
        json.dump({"reference": "WojciechMula/pyahocorasick v2.2.0 (bytes build)", "vectors": V}, f, indent=1)
    print("wrote", out, len(V), "vectors")
