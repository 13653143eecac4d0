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


# ----------------------------------------------------------------------------------------
# config 3: 100k multi-word keys over a lowercase vocabulary, space-separated text corpus
# ----------------------------------------------------------------------------------------
def text_vocab(n_tokens, seed=2):
    r = np.random.default_rng(seed)
    lens = r.integers(3, 13, size=n_tokens)
    letters = r.integers(97, 123, size=int(lens.sum()), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [letters[offs[i]:offs[i + 1]].tobytes() for i in range(n_tokens)]


def text_keys(vocab, n_keys, seed=3):
    """keys of 8-32 B: consecutive vocabulary tokens joined by single spaces, truncated"""
    rng = random.Random(seed)
    keys = set()
    while len(keys) < n_keys:
        i = rng.randrange(len(vocab))
        want = rng.randint(8, 32)
        s = vocab[i]
        j = i + 1
        while len(s) < want:
            s = s + b" " + vocab[j % len(vocab)]
            j += 1
        keys.add(s[:want])
    keys = sorted(keys)
    rng.shuffle(keys)
    return keys


def text_corpus(vocab, n_bytes, seed=4):
    """>= n_bytes of space-separated tokens sampled uniformly, cut to exactly n_bytes (uint8[])"""
    r = np.random.default_rng(seed)
    vcat = np.frombuffer(b" ".join(vocab) + b" ", dtype=np.uint8)
    vlen = np.array([len(v) + 1 for v in vocab], dtype=np.int64)       # token + its trailing space
    voff = np.concatenate([[0], np.cumsum(vlen)])[:-1]
    out = np.empty(n_bytes + 64, dtype=np.uint8)
    pos = 0
    while pos < n_bytes:
        k = max(1024, int((n_bytes - pos) / 8.5) + 16)
        which = r.integers(0, len(vocab), size=k)
        lens = vlen[which]
        ends = np.cumsum(lens)
        keep = int(np.searchsorted(ends, n_bytes + 32 - pos, side="right"))
        keep = max(1, min(keep, k))
        which, lens, ends = which[:keep], lens[:keep], ends[:keep]
        starts = ends - lens
        # gather: for each output byte, source = voff[token] + (byte index within token)
        tok_of = np.repeat(np.arange(keep), lens)
        within = np.arange(int(ends[-1])) - np.repeat(starts, lens)
        chunk = vcat[voff[which][tok_of] + within]
        take = min(len(chunk), len(out) - pos)
        out[pos:pos + take] = chunk[:take]
        pos += take
    return out[:n_bytes].copy()


# ----------------------------------------------------------------------------------------
# config 4: Snort-style byte signatures, packet payloads
# ----------------------------------------------------------------------------------------
def snort_signatures(n_sigs, seed=5, lo=4, hi=128):
    """unique signatures, length log-uniform in [lo,hi]; half uniform bytes 0-255, half printable ASCII"""
    r = np.random.default_rng(seed)
    sigs = set()
    while len(sigs) < n_sigs:
        m = n_sigs - len(sigs)
        lens = np.exp(r.uniform(np.log(lo), np.log(hi + 1), size=m)).astype(np.int64).clip(lo, hi)
        binary = r.random(m) < 0.5
        for L, b in zip(lens.tolist(), binary.tolist()):
            if b:
                sigs.add(r.integers(0, 256, size=L, dtype=np.uint8).tobytes())
            else:
                sigs.add(r.integers(32, 127, size=L, dtype=np.uint8).tobytes())
    sigs = sorted(sigs)
    random.Random(seed).shuffle(sigs)
    return sigs


def packet_payloads(sigs, n_bytes, seed=6, plo=64, phi=1500, plant_frac=0.01):
    """-> (data uint8[], offsets int64[n+1]): packets of length U[plo,phi], uniform bytes,
    plant_frac of the packets carry one planted signature"""
    r = np.random.default_rng(seed)
    n_est = int(n_bytes / ((plo + phi) / 2)) + 8
    lens = r.integers(plo, phi + 1, size=n_est)
    ends = np.cumsum(lens)
    n = int(np.searchsorted(ends, n_bytes, side="left")) + 1
    lens = lens[:n]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    data = r.integers(0, 256, size=int(off[-1]), dtype=np.uint8)
    planted = np.flatnonzero(r.random(n) < plant_frac)
    for p in planted.tolist():
        s = sigs[int(r.integers(0, len(sigs)))]
        if len(s) <= lens[p]:
            o = int(off[p] + r.integers(0, lens[p] - len(s) + 1))
            data[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    return data, off
