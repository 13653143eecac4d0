"""Seeded synthetic workloads of BASELINE.json / SURVEY.md §8(d).  Shared by bench.py and
the test-suite so both scan exactly the same bytes.  Pure numpy; no GPU, no oracle."""
import random

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def dna_keys(n_keys, seed=0, klo=8, khi=32):
    """n_keys unique ACGT strings, length U[klo,khi], sorted then shuffled (insertion order)."""
    rng = random.Random(seed)
    keys = set()
    while len(keys) < n_keys:
        keys.add("".join(rng.choice("ACGT") for _ in range(rng.randint(klo, khi))).encode())
    keys = sorted(keys)
    rng.shuffle(keys)
    return keys


def dna_reads(keys, n_reads, read_len, seed=1, plant=True):
    """uint8[n_reads, read_len] uniform over ACGT; every even-indexed read gets one uniformly
    chosen key planted at a uniform offset."""
    r = np.random.default_rng(seed)
    reads = np.ascontiguousarray(_ACGT[r.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)])
    if plant and len(keys):
        kcat = np.frombuffer(b"".join(keys), dtype=np.uint8)
        klen = np.array([len(k) for k in keys], dtype=np.int64)
        koff = np.concatenate([[0], np.cumsum(klen)])[:-1]
        rows = np.arange(0, n_reads, 2)
        which = r.integers(0, len(keys), size=len(rows))
        fits = klen[which] <= read_len
        rows, which = rows[fits], which[fits]
        pos = (r.random(len(rows)) * (read_len - klen[which] + 1)).astype(np.int64)
        for length in np.unique(klen[which]):          # vectorised per key length
            sel = np.flatnonzero(klen[which] == length)
            cols = pos[sel][:, None] + np.arange(length)[None, :]
            src = koff[which[sel]][:, None] + np.arange(length)[None, :]
            reads[rows[sel][:, None], cols] = kcat[src]
    return reads


def dna_workload(n_keys, n_reads, read_len, seed=0, klo=8, khi=32, plant=True):
    keys = dna_keys(n_keys, seed, klo, khi)
    return keys, dna_reads(keys, n_reads, read_len, seed + 1, plant)
