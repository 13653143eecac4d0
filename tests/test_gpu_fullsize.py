"""BASELINE.json configs 3 and 4 at their named size on one GPU (parity on a deterministic
1 % sample of chunks / packets against the oracle, whose CPU speed makes the full comparison infeasible)."""
import numpy as np
import pytest

import pyahocorasick_amd as acx
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from pyahocorasick_amd import workloads as W
from oracle import orc

pytestmark = [pytest.mark.gpu, pytest.mark.slow]


def _pair(keys):
    A = acx.Automaton(acx.STORE_INTS)
    O = orc.Oracle()
    for i, k in enumerate(keys):
        A.add_word(k, i)
        O.add_word(k, i)
    A.make_automaton()
    O.make_automaton()
    return A, O


def test_config4_one_million_signatures_one_gib_of_packets_sampled():
    """1,000,000 Snort-style signatures (4..128 B, half binary), 1 GiB of ragged packets on one GPU
    (config 4's share of two GPUs): every 100th packet is compared with the oracle, all offsets are checked
    for monotony, and the total equals the sum over a second scan of the two halves (batch-split invariance)"""
    sigs = W.snort_signatures(1_000_000, seed=5)
    data, off = W.packet_payloads(sigs, 1 << 30, seed=6)
    A, O = _pair(sigs)
    img = Image.from_automaton(A)
    n = len(off) - 1
    d_hay = DeviceBuffer.from_numpy(data, pad=64)
    d_off = DeviceBuffer.from_numpy(off)
    sc = Scanner(img)
    total = sc.scan(d_hay, len(data), n, dev_off=d_off)
    moff, e, v, _ = sc.fetch()
    assert moff[0] == 0 and moff[-1] == total == len(e) and np.all(np.diff(moff) >= 0)
    assert total > n // 200                                    # 1 % of the packets carry a planted signature
    sample = np.arange(0, n, 100)
    soff = np.concatenate([[0], np.cumsum(off[sample + 1] - off[sample])]).astype(np.int64)
    sdata = np.concatenate([data[off[h]:off[h + 1]] for h in sample])
    mo, oe, ov = O.batch_records(sdata, soff, 0)
    for k, h in enumerate(sample):
        assert np.array_equal(e[moff[h]:moff[h + 1]], oe[mo[k]:mo[k + 1]]) and np.array_equal(v[moff[h]:moff[h + 1]], ov[mo[k]:mo[k + 1]]), h
    assert mo[-1] > 0
    half = n // 2
    d_off2 = DeviceBuffer.from_numpy(off[half:] - off[half])
    sc2 = Scanner(img)
    t2 = sc2.scan(d_hay.ptr.value + int(off[half]), len(data) - int(off[half]), n - half, dev_off=d_off2)
    assert t2 == total - moff[half]


def test_config3_two_gib_shard_sampled():
    """the 100k multi-word text keys of config 3 over ONE 2 GiB shard (what each of 8 GPUs gets of the 16 GB
    corpus), scanned as a single haystack; every 100th 64 KiB chunk is compared with the oracle (the chunk
    plus longest_word-1 bytes of left context, matches ending inside the chunk)"""
    vocab = W.text_vocab(1_000_000, seed=2)
    keys = W.text_keys(vocab, 100_000, seed=3)
    A, O = _pair(keys)
    longest = max(len(k) for k in keys)
    nbytes = (2 << 30) - (1 << 16)                             # end_index is a C int: one haystack stays below 2^31
    piece = 64 << 20
    parts = [W.text_corpus(vocab, min(piece, nbytes - o), seed=4 + o // piece) for o in range(0, nbytes, piece)]
    corpus = np.concatenate(parts)
    del parts
    for j, k in enumerate(keys[:5000]):                        # plant: something to find in every region
        p = (j * 429_497 + 12_345) % (nbytes - 64)
        corpus[p:p + len(k)] = np.frombuffer(k, dtype=np.uint8)
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(corpus, pad=64)
    d_off = DeviceBuffer.from_numpy(np.array([0, nbytes], dtype=np.int64))
    sc = Scanner(img)
    total = sc.scan(d_hay, nbytes, 1, dev_off=d_off)
    moff, e, v, _ = sc.fetch()
    assert moff[-1] == total == len(e) and total >= 5000 and np.all(np.diff(e.astype(np.int64)) >= 0)
    CH = 1 << 16
    starts = np.arange(0, nbytes - CH, 100 * CH)
    ctx = [max(0, int(s) - (longest - 1)) for s in starts]
    soff = np.concatenate([[0], np.cumsum([int(s) + CH - c for s, c in zip(starts, ctx)])]).astype(np.int64)
    sdata = np.concatenate([corpus[c:int(s) + CH] for s, c in zip(starts, ctx)])
    mo, oe, ov = O.batch_records(sdata, soff, 0)
    checked = 0
    for k, (s, c) in enumerate(zip(starts, ctx)):
        want_e = oe[mo[k]:mo[k + 1]].astype(np.int64) + c
        keep = want_e >= s
        lo, hi = np.searchsorted(e, s, side="left"), np.searchsorted(e, s + CH, side="left")
        assert np.array_equal(e[lo:hi], want_e[keep]) and np.array_equal(v[lo:hi], ov[mo[k]:mo[k + 1]][keep]), int(s)
        checked += int(keep.sum())
    assert checked > 0
