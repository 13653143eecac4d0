"""Generate tests/golden/ref_sequence.json by RUNNING THE REFERENCE (oracle/_ref, bytes build) on
KEY_SEQUENCE automata: keys are tuples of integers 0..65535.

    make -C oracle && python tests/golden/make_sequence_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle", "_ref"))
import ahocorasick as ref  # noqa: E402


def main():
    rng = random.Random(8)
    cases = []
    for trial in range(24):
        hi = [5, 300, 65535][trial % 3]

        def letter():
            return rng.choice([rng.randint(0, 5), rng.randint(0, hi)])

        keys = sorted({tuple(letter() for _ in range(rng.randint(1, 5))) for _ in range(rng.randint(1, 25))})
        rng.shuffle(keys)
        store = [ref.STORE_INTS, ref.STORE_LENGTH, ref.STORE_ANY][(trial // 3) % 3]
        probes = [tuple(letter() for _ in range(rng.randint(0, 6))) for _ in range(8)]
        hays = []
        for _ in range(4):
            h = [letter() for _ in range(rng.randint(0, 60))]
            if rng.random() < 0.7:
                h += list(rng.choice(keys)) * 2
            hays.append(h)
        pats = [[]] + [[bytes(rng.randint(0, 5) for _ in range(rng.randint(0, 3))).hex()] +
                       ([bytes([rng.randint(0, 5)]).hex()] if rng.random() < 0.5 else []) for _ in range(5)]
        A = ref.Automaton(store, ref.KEY_SEQUENCE)
        for i, k in enumerate(keys):
            if store == ref.STORE_LENGTH:
                A.add_word(k)
            elif store == ref.STORE_INTS:
                A.add_word(k, i - 3)
            else:
                A.add_word(k, [i, list(k)])
        case = {"store": store, "keys": keys, "probes": probes, "hays": hays, "pats": pats,
                "enum_keys": [x.hex() for x in A.keys()], "enum_values": list(A.values()),
                "pat_results": [[[x.hex() for x in A.keys(*[bytes.fromhex(q) for q in p])],
                                 list(A.values(*[bytes.fromhex(q) for q in p]))] for p in pats],
                "probe_results": [[A.exists(p), A.match(p), A.longest_prefix(p), A.get(p, None), p in A] for p in probes]}
        A.make_automaton()
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
        case["iter"] = [[list(m) for m in A.iter(tuple(h))] for h in hays]
        case["iter_long"] = [[list(m) for m in A.iter_long(tuple(h))] for h in hays]
        case["iter_range"] = [[list(m) for m in A.iter(tuple(h), 1, len(h) - 1)] if len(h) > 3 else None for h in hays]
        cases.append(case)
    out = os.path.join(HERE, "ref_sequence.json")
    with open(out, "w") as f:
        json.dump({"generator": "tests/golden/make_sequence_golden.py", "cases": cases}, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
