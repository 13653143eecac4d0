"""Generate tests/golden/ref_items.json by RUNNING THE REFERENCE (oracle/_ref, bytes build): what
keys() / values() return for prefix / wildcard / how patterns, and get_stats().

    make -C oracle && python tests/golden/make_items_golden.py

items() is derived as zip(keys, values): the reference's bytes build returns mangled keys from
items() (it formats its uint16 letter buffer with "y#": b'b\\x00a' for b'bab'), which this engine
does not reproduce.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "_ref"))
import ahocorasick as ref  # noqa: E402


def main():
    rng = random.Random(77)
    cases = []
    for trial in range(24):
        alpha = [b"ab", b"abc", b"abcd?"][trial % 3]
        store = [ref.STORE_INTS, ref.STORE_LENGTH, ref.STORE_ANY][(trial // 3) % 3]
        keys = sorted({bytes(rng.choice(alpha) for _ in range(rng.randint(1, 6))) for _ in range(rng.randint(0, 30))})
        rng.shuffle(keys)                      # insertion order matters; keep it reproducible
        A = ref.Automaton(store)
        vals = []
        for i, k in enumerate(keys):
            if store == ref.STORE_LENGTH:
                A.add_word(k)
                vals.append(None)
            elif store == ref.STORE_INTS:
                v = rng.choice([i - 5, 2**31 - 1, -2**31])
                A.add_word(k, v)
                vals.append(v)
            else:
                A.add_word(k, [i, k.hex()])
                vals.append([i, k.hex()])
        finalised = bool(keys) and trial % 2 == 0
        if finalised:
            A.make_automaton()
        queries = [[]]
        for _ in range(12):
            q = [bytes(rng.choice(alpha) for _ in range(rng.randint(0, 5))).hex()]
            if rng.random() < 0.6:
                q.append(b"?".hex())
                if rng.random() < 0.6:
                    q.append(rng.choice([0, 1, 2]))
            queries.append(q)
        res = []
        for q in queries:
            args = [bytes.fromhex(x) if isinstance(x, str) else x for x in q]
            res.append({"keys": [k.hex() for k in A.keys(*args)], "values": list(A.values(*args))})
        cases.append({"store": store, "keys": [k.hex() for k in keys], "values": vals, "finalised": finalised,
                      "queries": queries, "results": res, "iter": [k.hex() for k in A], "stats": A.get_stats()})
    out = os.path.join(HERE, "ref_items.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_items_golden.py", "cases": cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
