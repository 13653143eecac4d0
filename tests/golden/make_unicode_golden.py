"""Generate tests/golden/ref_unicode.json by RUNNING THE REFERENCE'S UNICODE BUILD
(oracle/_ref/unicode, -DAHOCORASICK_UNICODE: keys and haystacks are str, one letter = one code
point): trie API, enumeration and search results.

    make -C oracle && python tests/golden/make_unicode_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "_ref", "unicode"))
import ahocorasick as ref  # noqa: E402

assert ref.unicode == 1
# no letters outside the BMP: with 4-byte-kind str objects (e.g. an emoji in the haystack) the reference's
# unicode build corrupts memory (6 crashes in 6 runs of this script here); those are covered in
# tests/test_unicode_build.py against the bytes build instead
ALPHA = "abcżółć日本語ß€ "


def main():
    rng = random.Random(11)
    cases = []
    for trial in range(18):
        alpha = ALPHA if trial % 3 else "ab ć"
        keys = sorted({"".join(rng.choice(alpha.replace(" ", "")) for _ in range(rng.randint(1, 5))) for _ in range(rng.randint(1, 30))})
        rng.shuffle(keys)                      # insertion order matters (child order); keep it reproducible
        store = [ref.STORE_INTS, ref.STORE_LENGTH, ref.STORE_ANY][trial % 3]
        A = ref.Automaton(store)
        vals = []
        for i, k in enumerate(keys):
            if store == ref.STORE_LENGTH:
                A.add_word(k)
                vals.append(None)
            elif store == ref.STORE_INTS:
                A.add_word(k, i - 3)
                vals.append(i - 3)
            else:
                A.add_word(k, [i, k])
                vals.append([i, k])
        pats = [[]]
        for _ in range(8):
            q = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 4)))]
            if rng.random() < 0.6:
                q.append(rng.choice(["?", "ł", "日"]))
                if rng.random() < 0.6:
                    q.append(rng.choice([0, 1, 2]))
            pats.append(q)
        probes = ["".join(rng.choice(alpha) for _ in range(rng.randint(0, 6))) for _ in range(8)]
        case = {"store": store, "keys": keys, "values": vals, "pats": pats, "probes": probes,
                "enum": [[list(A.keys(*q)), list(A.values(*q))] for q in pats], "iter_keys": list(A),
                "probe_results": [[A.exists(p), A.match(p), A.longest_prefix(p), A.get(p, None)] for p in probes]}
        A.make_automaton()
        # persistence of the unicode build: __reduce__ payload and save file (4-byte letters)
        import pickle
        import tempfile
        cls, args = A.__reduce__()
        case["reduce"] = {"chunks": [c.hex() for c in args[0]], "rest": list(args[1:6]),
                          "values_pickle": None if args[6] is None else pickle.dumps(args[6], protocol=2).hex()}
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "a.sav")
            if store == ref.STORE_ANY:
                A.save(path, lambda v: pickle.dumps(v, protocol=2))
            else:
                A.save(path)
            case["savefile"] = open(path, "rb").read().hex()
        hays = []
        for _ in range(5):
            h = "".join(rng.choice(alpha) for _ in range(rng.randint(0, 120)))
            if keys and rng.random() < 0.7:
                h += rng.choice(keys) * 2 + rng.choice(alpha)
            hays.append(h)
        searches = []
        for h in hays:
            n = len(h)
            s0 = rng.randint(0, max(0, n - 1)) if n else 0
            e0 = rng.randint(s0, n)
            item = {"hay": h, "iter": [list(m) for m in A.iter(h)], "iter_long": [list(m) for m in A.iter_long(h)],
                    "iter_ws": [list(m) for m in A.iter(h, ignore_white_space=True)]}
            if n:
                item["range"] = [s0, e0]
                item["iter_range"] = [list(m) for m in A.iter(h, s0, e0)]
                item["iter_long_range"] = [list(m) for m in A.iter_long(h, s0, e0)]
            found = []
            A.find_all(h, lambda i, v: found.append([i, v]))
            item["find_all"] = found
            searches.append(item)
        # streaming: set() continues on the next chunk with the state kept
        if len(hays) >= 2 and hays[0]:
            it = A.iter(hays[0])
            first = [list(m) for m in it]
            it.set(hays[1])
            second = [list(m) for m in it]
            it.set(hays[2], True)
            third = [list(m) for m in it]
            case["set"] = {"chunks": hays[:3], "results": [first, second, third]}
        case["searches"] = searches
        cases.append(case)
    out = os.path.join(HERE, "ref_unicode.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_unicode_golden.py", "cases": cases}, f, ensure_ascii=False)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
