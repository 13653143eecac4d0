"""Randomised differential run on the GPU: itop walk (1 and 2 items per lane), plain walk, stride and
offsets entry, against the oracle — many alphabets / key sets / haystack shapes.  Not part of the
test suite (tests/test_gpu_parity.py holds the pinned cases); use it after touching the kernels:
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


def one_case(rng, trial):
    sigma = int(rng.choice([2, 3, 4, 4, 4, 5, 8, 12, 16, 20]))
    alpha = rng.choice(256, size=sigma, replace=False).astype(np.uint8)
    n_keys = int(rng.choice([1, 5, 50, 500, 5000, 20000]))
    kmax = int(rng.choice([3, 8, 14, 40]))
    keys = list({bytes(rng.choice(alpha, size=int(k)).tobytes()) for k in rng.integers(1, kmax + 1, size=n_keys)})
    if rng.random() < 0.2:                                           # chains of nested keys: many outputs per state
        c = bytes([int(alpha[0])])
        keys += [c * k for k in range(1, int(rng.integers(2, 45)))]
        keys = list(dict.fromkeys(keys))
    A, O = build_pair(keys)
    n, L = int(rng.integers(1, 600)), int(rng.integers(1, 700))
    foreign = rng.random() < 0.5
    pool = np.concatenate([alpha, rng.choice(256, size=3).astype(np.uint8)]) if foreign else alpha
    reads = np.ascontiguousarray(pool[rng.integers(0, len(pool), size=(n, L))])
    for i in range(0, n, 2):
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        if len(k) <= L:
            o = int(rng.integers(0, L - len(k) + 1))
            reads[i, o:o + len(k)] = k
    off = np.arange(n + 1, dtype=np.int64) * L
    mo, oe, ov = O.batch(reads.tobytes(), off, 0)
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(reads.reshape(-1), pad=64)
    d_off = DeviceBuffer.from_numpy(off)
    fins = []
    for variant in (0, 1 << 17, 1 << 16):
        sc = Scanner(img)
        for kw in (dict(stride=L), dict(dev_off=d_off)):
            sc.scan(d_hay, n * L, n, want_final_state=True, variant=variant, **kw)
            moff, e, v, fin = sc.fetch()
            ok = np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)
            if not ok:
                raise SystemExit("MISMATCH trial %d variant %d %s sigma %d keys %d kmax %d n %d L %d itop_depth %d"
                                 % (trial, variant, list(kw), sigma, len(keys), kmax, n, L, img.itop_depth))
            fins.append(fin)
    from pyahocorasick_amd import ACX_SCAN_LONG
    lo, le, lv = O.batch(reads.tobytes(), off, 1)                   # iter_long
    sc = Scanner(img)
    sc.scan(d_hay, n * L, n, stride=L, mode=ACX_SCAN_LONG)
    moff, e, v, _ = sc.fetch()
    if not (np.array_equal(moff, lo) and np.array_equal(e, le) and np.array_equal(v, lv)):
        raise SystemExit("ITER_LONG MISMATCH trial %d sigma %d keys %d" % (trial, sigma, len(keys)))
    for fin in fins[1:]:
        if not np.array_equal(fin, fins[0]):
            raise SystemExit("FINAL STATE MISMATCH trial %d" % trial)
    return len(oe), img.itop_depth


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
