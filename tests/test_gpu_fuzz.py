"""A fixed-seed slice of tools/fuzz_gpu.py inside the GPU suite (random alphabets of 2..256 symbols, key sets, equal and
ragged batches, the stream / general / serial kernels, iter_long, final states: everything against the oracle), and the
`min_hay_len` contract of acx_scan_params: a batch that breaks its promise is noticed on the device and scanned again."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from helpers import build_pair                                     # noqa: E402
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner   # noqa: E402


def test_fuzz_slice_fixed_seed():
    import fuzz_gpu
    rng = np.random.default_rng(20240924)
    matches = 0
    for trial in range(110):                                         # ~ 60 s on an MI355X
        m, _ = fuzz_gpu.one_case(rng, trial)
        matches += m
    assert matches > 0


def test_fuzz_stream4_slice_fixed_seed():
    """tools/fuzz_stream4.py: the batches k_ppm_stream4 takes (four-letter alphabets, fixed strides and — every other case — the same reads
    cut to ragged lengths as an offsets batch, empty haystacks included; keys of up to 33 letters; bytes of no key, nested keys, dense
    dictionaries, runs of tiles per wave, index_base) against the oracle, k_ppm_stream and k_ppm_scan; every case checks that the plan
    names k_ppm_stream4"""
    import fuzz_stream4
    rng = np.random.default_rng(424242)
    matches = 0
    for trial in range(24):                                          # ~ 20 s on an MI355X
        matches += fuzz_stream4.one_case(rng, trial)
    assert matches > 0


def _check(A, O, flat, off, **kw):
    img = Image.from_automaton(A)
    sc = Scanner(img)
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    d_off = DeviceBuffer.from_numpy(off)
    sc.scan(d_hay, len(flat), len(off) - 1, dev_off=d_off, **kw)
    moff, e, v, _ = sc.fetch()
    mo, oe, ov = O.batch(flat.tobytes(), off, 0)
    assert np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)
    return img


def test_min_hay_len_promise_broken_is_detected_and_rescanned():
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))]) for k in rng.integers(2, 12, size=400)})
    A, O = build_pair(keys)
    # (a) one 3-byte haystack in a batch that promises 8: few starts per tile, the stream kernel copes
    lens = [200] * 50 + [3] + [200] * 50
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=int(off[-1]))])
    GENERAL = 1 << 19         # the general stream kernel (k_ppm_stream's offsets form: starts through the queue, the promise checked) instead of k_ppm_stream4's
    img = _check(A, O, flat, off, min_hay_len=8, variant=GENERAL)
    assert img.ppm_kernel(stride=0, has_offsets=True, variant=GENERAL, min_hay_len=8, dev_hay=0, n_hay=len(lens)) == "stream"
    # (four letters, keys of up to 33: k_ppm_stream4's offsets form — the starts are a bitmap there, any lengths, no promise to break)
    assert img.ppm_kernel(stride=0, has_offsets=True, variant=0, min_hay_len=8, dev_hay=0, n_hay=len(lens)) == "stream4"
    _check(A, O, flat, off, min_hay_len=8)
    # (a') a few EMPTY haystacks: two starts at one position, nowhere near too many per tile — noticed as well
    lens = [200] * 20 + [0] + [200] * 20 + [0, 0] + [5] + [200] * 20 + [0]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=int(off[-1]))])
    _check(A, O, flat, off, min_hay_len=8, variant=GENERAL)
    _check(A, O, flat, off, min_hay_len=8)
    # (b) thousands of 3-byte haystacks, still promising 8: more starts in a tile than the kernel has room for;
    # it raises its flag and the result comes from a second scan on the general kernels
    lens = [3] * 5000 + [300] * 20 + [1] * 3000 + [0] * 10 + [2] * 999
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=int(off[-1]))])
    _check(A, O, flat, off, min_hay_len=8, variant=GENERAL)
    _check(A, O, flat, off, min_hay_len=8)
    _check(A, O, flat, off, min_hay_len=0)
    # the same through the asynchronous entry
    img = Image.from_automaton(A)
    sc = Scanner(img)
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    d_off = DeviceBuffer.from_numpy(off)
    sc.scan(d_hay, len(flat), len(off) - 1, dev_off=d_off, min_hay_len=8, asynchronous=True)
    sc.wait()
    moff, e, v, _ = sc.fetch()
    mo, oe, ov = O.batch(flat.tobytes(), off, 0)
    assert np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)


def test_one_scanner_through_batches_of_every_shape():
    """What a result object carries from one scan to the next — control words that the gather of a fixed-stride scan
    zeroes, block sums kept in two alternating sets, buffers that grow — survives any order of batch shapes: fixed
    stride small and large (a record pool that overflows and is issued again), offsets, a broken promise, other kernel
    families in between, asynchronous and not."""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))]) for k in rng.integers(3, 12, size=500)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    sc = Scanner(img)
    shapes = [("stride", 300, 120), ("stride", 60000, 150), ("offsets", 500, 0), ("stride", 40, 64), ("stride", 200000, 100),
              ("broken", 800, 0), ("stride", 5000, 33), ("serial", 700, 90), ("general", 900, 80), ("stride", 60000, 150),
              ("stride", 1, 4096), ("offsets", 3000, 0), ("stride", 70000, 151)]
    for i, (kind, n, L) in enumerate(shapes):
        asynchronous = i % 2 == 1
        if kind in ("stride", "serial", "general"):
            flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=n * L)])
            off = np.arange(n + 1, dtype=np.int64) * L
            variant = {"stride": 0, "serial": 1 << 23, "general": (1 << 24) | (1 << 28)}[kind]
            d_hay = DeviceBuffer.from_numpy(flat, pad=64)
            sc.scan(d_hay, len(flat), n, stride=L, variant=variant, asynchronous=asynchronous)
        else:
            lens = rng.integers(8, 300, size=n)
            if kind == "broken":
                lens[::7] = 0
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=int(off[-1]))])
            d_hay = DeviceBuffer.from_numpy(flat, pad=64)
            d_off = DeviceBuffer.from_numpy(off)
            sc.scan(d_hay, len(flat), n, dev_off=d_off, min_hay_len=8, asynchronous=asynchronous)
        moff, e, v, _ = sc.fetch()
        mo, oe, ov = O.batch(flat.tobytes(), off, 0)
        assert np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov), (i, kind, n, L)


def test_stream4_offsets_batches_one_scanner_many_batches():
    """k_ppm_stream4's offsets form keeps a bitmap of the haystacks' starts per result; the gather of a scan zeroes the words its scan used, so that the
    next scan of the same result scatters into a clean bitmap without a memset.  One Scanner, batches of different sizes and shapes one after the
    other (a larger one in between makes the buffer grow), empty haystacks, bytes of no key — every offset and record against the oracle every time"""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))]) for k in rng.integers(3, 30, size=3000)})
    A, O = build_pair(keys)
    img = Image.from_automaton(A)
    sc = Scanner(img)
    shapes = [(3000, 100, 150, 0.0), (500, 0, 40, 0.02), (20000, 60, 300, 0.0), (3000, 100, 150, 0.0), (40, 2000, 9000, 0.001), (7000, 1, 90, 0.0)]
    for n, lo, hi, p_other in shapes:
        lens = rng.integers(lo, hi + 1, size=n, dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        flat = np.ascontiguousarray(alpha[rng.integers(0, 4, size=int(off[-1]))])
        for h in rng.integers(0, n, size=n // 3):                        # plant keys
            k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
            if lens[h] >= len(k):
                p = int(off[h] + rng.integers(0, lens[h] - len(k) + 1))
                flat[p:p + len(k)] = k
        if p_other:
            flat[rng.random(len(flat)) < p_other] = ord("N")
        assert img.ppm_kernel(stride=0, has_offsets=True, min_hay_len=8, dev_hay=0, n_hay=n) == "stream4"
        d_hay = DeviceBuffer.from_numpy(flat, pad=64)
        d_off = DeviceBuffer.from_numpy(off)
        sc.scan(d_hay, len(flat), n, dev_off=d_off, min_hay_len=8)
        moff, e, v, _ = sc.fetch()
        mo, oe, ov = O.batch(flat.tobytes(), off, 0)
        assert mo[-1] > 0 and np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov), (n, lo, hi)
