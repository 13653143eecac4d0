"""Randomised differential run on the GPU against the oracle: the stream kernel (fixed stride, and offsets with
min_hay_len), the general position-parallel kernel and the serial walks, many alphabets (2 .. 256 symbols) /
key sets / haystack shapes — equal lengths, ragged offsets, now and then a batch large enough that every wave
takes a run of several tiles.  iter_long and the final states ride along.  Not part of the test suite
(tests/test_gpu_ppm.py / test_gpu_parity.py hold the pinned cases); use it after touching the kernels:
    python tools/fuzz_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_pair                                     # noqa: E402
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner   # noqa: E402


STREAM, GENERAL, SERIAL = 1 << 28, (1 << 24) | (1 << 28), 1 << 23


def one_case(rng, trial):
    sigma = int(rng.choice([2, 3, 4, 4, 4, 5, 8, 12, 16, 20, 64, 200, 256]))
    alpha = rng.choice(256, size=sigma, replace=False).astype(np.uint8)
    n_keys = int(rng.choice([1, 5, 50, 500, 5000, 20000]))
    kmax = int(rng.choice([3, 8, 14, 40, 120]))
    keys = list({bytes(rng.choice(alpha, size=int(k)).tobytes()) for k in rng.integers(1, kmax + 1, size=n_keys)})
    if rng.random() < 0.2:                                           # chains of nested keys: many outputs per state
        c = bytes([int(alpha[0])])
        keys += [c * k for k in range(1, int(rng.integers(2, 45)))]
        keys = list(dict.fromkeys(keys))
    A, O = build_pair(keys)
    big = rng.random() < 0.08                                        # > 8 M positions: runs of tiles per wave
    n, L = (int(rng.integers(60000, 120000)), int(rng.integers(150, 320))) if big else (int(rng.integers(1, 600)), int(rng.integers(1, 700)))
    foreign = rng.random() < 0.5
    pool = np.concatenate([alpha, rng.choice(256, size=3).astype(np.uint8)]) if foreign else alpha
    reads = np.ascontiguousarray(pool[rng.integers(0, len(pool), size=(n, L))])
    for i in range(0, n, 2 if not big else 50):
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        if len(k) <= L:
            o = int(rng.integers(0, L - len(k) + 1))
            reads[i, o:o + len(k)] = k
    flat = reads.reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * L
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    shapes = [("equal", off)]
    if n * L >= 64:                                                  # the same bytes cut at random places, no piece shorter than 8
        cuts = np.unique(np.concatenate([[0, n * L], rng.integers(0, n * L + 1, size=max(1, n // 2))]))
        keep = [0]
        for c in cuts[1:]:
            if c - keep[-1] >= 8 and n * L - c >= 8 or c == n * L:
                keep.append(int(c))
        if keep[-1] != n * L:
            keep[-1] = n * L
        if len(keep) >= 2 and min(np.diff(keep)) >= 8:
            shapes.append(("ragged", np.asarray(keep, dtype=np.int64)))
    fins_equal = []
    total = 0
    for name, o_arr in shapes:
        nh = len(o_arr) - 1
        mo, oe, ov = O.batch(flat.tobytes(), o_arr, 0)
        total += len(oe)
        d_off = DeviceBuffer.from_numpy(o_arr)
        shortest = int(np.diff(o_arr).min())
        entries = [dict(dev_off=d_off), dict(dev_off=d_off, min_hay_len=shortest)] + ([dict(stride=L)] if name == "equal" else [])
        for variant in (0, STREAM, GENERAL, SERIAL):
            sc = Scanner(img)
            for kw in entries:
                sc.scan(d_hay, n * L, nh, want_final_state=True, variant=variant, **kw)
                moff, e, v, fin = sc.fetch()
                if not (np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)):
                    raise SystemExit("MISMATCH trial %d %s variant %#x %s sigma %d keys %d kmax %d n %d L %d"
                                     % (trial, name, variant, sorted(kw), sigma, len(keys), kmax, n, L))
                if name == "equal":
                    fins_equal.append(fin)
    from pyahocorasick_amd import ACX_SCAN_LONG
    lo, le, lv = O.batch(flat.tobytes(), off, 1)                    # iter_long
    # the position-parallel form (where it applies: else the serial walk answers) synchronous and asynchronous, the sweep straight from the
    # record pool (variant bit 26), the serial walk (bit 25)
    for variant, asyn in ((0, False), (0, True), (1 << 27, False), (1 << 27, True), (1 << 26, False), (1 << 26, True), (1 << 25, False)):
        sc = Scanner(img)
        sc.scan(d_hay, n * L, n, stride=L, mode=ACX_SCAN_LONG, variant=variant, asynchronous=asyn)
        moff, e, v, _ = sc.fetch()
        if not (np.array_equal(moff, lo) and np.array_equal(e, le) and np.array_equal(v, lv)):
            raise SystemExit("ITER_LONG MISMATCH trial %d variant %#x async %s sigma %d keys %d kmax %d n %d L %d" % (trial, variant, asyn, sigma, len(keys), kmax, n, L))
    for fin in fins_equal[1:]:
        if not np.array_equal(fin, fins_equal[0]):
            raise SystemExit("FINAL STATE MISMATCH trial %d" % trial)
    return total, img.itop_depth


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, trials, matches, depths = time.time(), 0, 0, {}
    while time.time() - t0 < seconds:
        m, d = one_case(rng, trials)
        trials += 1
        matches += m
        depths[d] = depths.get(d, 0) + 1
    print("fuzz ok: %d cases, %d matches, itop depths %s, %.0f s" % (trials, matches, dict(sorted(depths.items())), time.time() - t0))


if __name__ == "__main__":
    main()
