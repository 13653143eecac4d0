"""Generate tests/golden/ref_persist.json by RUNNING THE REFERENCE (oracle/_ref build, bytes build):
pickle payloads (Automaton.__reduce__) and save files (Automaton.save) of a few automata, with
the keys, values and search results the reference itself reports for them.

    make -C oracle && python tests/golden/make_persist_golden.py

Used by tests/test_persistence.py.  The reference cannot travel to the GPU box;
these fixtures can.
"""
import json
import os
import pickle
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "_ref"))
import ahocorasick as ref  # noqa: E402

assert ref.unicode == 0, "fixtures must come from the bytes build"


def dump(name, A, haystacks):
    cls, args = A.__reduce__()
    case = {"name": name, "kind": A.kind, "store": A.store, "count": len(A)}
    if args:
        chunks, kind, store, key_type, count, longest, values = args
        case["reduce"] = {"chunks": [c.hex() for c in chunks], "kind": kind, "store": store, "key_type": key_type,
                          "count": count, "longest_word": longest,
                          "values_pickle": None if values is None else pickle.dumps(values, protocol=2).hex()}
    else:
        case["reduce"] = None
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "a.sav")
        if A.store == ref.STORE_ANY:
            A.save(path, lambda v: pickle.dumps(v, protocol=2))
        else:
            A.save(path)
        case["savefile"] = open(path, "rb").read().hex()
    keys = sorted(A.keys())
    case["keys"] = [k.hex() for k in keys]
    case["values_json"] = [A.get(k) for k in keys]            # ints, or JSON-able objects
    case["stats"] = {k: A.get_stats()[k] for k in ("nodes_count", "words_count", "longest_word")}
    case["haystacks"] = [h.hex() for h in haystacks]
    if A.kind == ref.AHOCORASICK:
        case["iter"] = [[list(m) for m in A.iter(h)] for h in haystacks]
        case["iter_long"] = [[list(m) for m in A.iter_long(h)] for h in haystacks]
    return case


def main():
    rng = random.Random(2024)
    cases = []

    A = ref.Automaton(ref.STORE_INTS)
    for i, w in enumerate([b"he", b"her", b"hers", b"she"]):
        A.add_word(w, i + 10)
    A.make_automaton()
    cases.append(dump("ints_ushers", A, [b"ushers", b"", b"hehehershe"]))

    A = ref.Automaton(ref.STORE_LENGTH)
    for w in [b"abc", b"bcd", b"c", b"abcdefgh"]:
        A.add_word(w)
    cases.append(dump("length_trie_not_finalised", A, []))

    A = ref.Automaton()
    for w, v in [(b"ab", ["x", 1]), (b"b", [2]), (b"abc", None), (b"zzz", "a string"), (b"zz", {"k": [1, 2]})]:
        A.add_word(w, v)
    A.make_automaton()
    cases.append(dump("any_objects", A, [b"xabczzzz", b"bbb"]))

    A = ref.Automaton(ref.STORE_INTS)
    keys = sorted({bytes(rng.choice(b"abc") for _ in range(rng.randint(1, 9))) for _ in range(150)})
    rng.shuffle(keys)                          # insertion order matters; keep it reproducible
    for k in keys:
        A.add_word(k, rng.randint(-2**31, 2**31 - 1))
    A.make_automaton()
    cases.append(dump("ints_random_abc", A, [bytes(rng.choice(b"abcd") for _ in range(200)) for _ in range(4)]))

    A = ref.Automaton(ref.STORE_INTS)
    for i in range(256):
        A.add_word(bytes([i, 255 - i, i]), i)
    A.add_word(bytes(range(256)), 1000)
    A.make_automaton()
    cases.append(dump("all_256_byte_values", A, [bytes(range(256)) * 2, bytes([7, 248, 7, 0, 255, 0])]))

    A = ref.Automaton(ref.STORE_INTS)
    for i, w in enumerate([b"abc", b"abcd", b"abd", b"b", b"bcd"]):
        A.add_word(w, i)
    A.remove_word(b"abcd")
    A.remove_word(b"b")
    A.make_automaton()
    cases.append(dump("after_remove_word", A, [b"abcdabdbcd"]))

    A = ref.Automaton()
    cases.append(dump("empty", A, []))

    out = os.path.join(HERE, "ref_persist.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_persist_golden.py", "reference": "pyahocorasick bytes build (oracle/_ref)",
                   "cases": cases}, f, indent=0)
    print("wrote", out, os.path.getsize(out), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
