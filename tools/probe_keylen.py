"""Walk time against the key-length range (100k ACGT keys, 1M x 150 B reads): separates the cost of
reporting matches from the cost of walking (profiles/r02_experiments.md)."""
import sys, json
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pyahocorasick_amd import Automaton, STORE_INTS
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from pyahocorasick_amd.workloads import dna_keys, dna_reads
for klo, khi in ((8, 32), (8, 10), (8, 9), (9, 9), (12, 32)):
    keys = dna_keys(100_000, seed=0, klo=klo, khi=khi)
    reads = dna_reads(keys, 1_000_000, 150, seed=1)
    A = Automaton(STORE_INTS)
    for i, k in enumerate(keys): A.add_word(bytes(k), i)
    A.make_automaton()
    img = Image.from_automaton(A)
    n, L = reads.shape
    d = DeviceBuffer.from_numpy(reads.reshape(-1), pad=64)
    sc = Scanner(img)
    best = None
    for _ in range(6):
        sc.scan(d, n * L, n, stride=L, timing=True)
        t = sc.timing_ms()
        best = t if best is None or t["walk"] < best["walk"] else best
    print(json.dumps({"key_len": [klo, khi], "D": img.itop_depth, "states": int(img.num_states), "matches": int(sc.num_matches()), **{k: round(v, 4) for k, v in best.items()}}), flush=True)
