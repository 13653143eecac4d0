"""Randomised differential run of k_ppm_stream4 (acx_ppm_stream4.hip) on the GPU: four-letter alphabets that a shift tells
apart, dictionaries whose image has C = 9 / F = 10, keys of at most 33 letters, fixed strides 8 .. 2047 and — every other case —
the same reads cut to ragged lengths (empty ones too) as an OFFSETS batch — the batches the
kernel takes — against the oracle and against k_ppm_stream (variant bit 19) / k_ppm_scan, with what the kernel has special paths for:
bytes of no key, haystacks shorter than the longest key, many keys ending at one position (nested keys: the general
enumeration), dense dictionaries (a tile holds more candidates than a round), batches large enough that every wave takes
a run of tiles, index_base.  Every case asserts that the scan plan really names k_ppm_stream4.
    python tools/fuzz_stream4.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_pair                                     # noqa: E402
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner   # noqa: E402

STREAM_ANY = (1 << 19) | (1 << 28)
ALPHABETS = [b"ACGT", b"acgt", b"ACGU", b"\x00\x01\x02\x03", b"AEIM", b"0123", b"\x10\x50\x90\xd0"]


def one_case(rng, trial):
    alpha = np.frombuffer(ALPHABETS[int(rng.integers(0, len(ALPHABETS)))], dtype=np.uint8)
    n_keys = int(rng.choice([40, 400, 4000, 40000, 200000]))
    kmin, kmax = int(rng.choice([1, 4, 8, 10])), int(rng.choice([10, 12, 20, 33]))
    kmin = min(kmin, kmax)
    keys = list({bytes(rng.choice(alpha, size=int(k)).tobytes()) for k in rng.integers(kmin, kmax + 1, size=n_keys)})
    if not any(len(k) >= 10 for k in keys):
        keys.append(bytes(rng.choice(alpha, size=12).tobytes()))
    if rng.random() < 0.3:                                           # nested keys: many records per position
        c = bytes([int(alpha[int(rng.integers(0, 4))])])
        keys += [c * k for k in range(1, int(rng.integers(3, 34)))]
        keys = list(dict.fromkeys(keys))
    vals = rng.integers(-2**31, 2**31, size=len(keys)).tolist()
    A, O = build_pair(keys, vals)
    big = rng.random() < 0.15                                        # > 8 M positions: runs of tiles per wave
    L = int(rng.choice([8, 9, 31, 33, 64, 150, 151, 255, 1000, 2047]))
    n = int(rng.integers(9_000_000, 12_000_000)) // L if big else int(rng.integers(1, 4000))
    foreign = rng.random() < 0.4
    pool = np.concatenate([alpha] * 8 + [rng.choice(256, size=2).astype(np.uint8)]) if foreign else alpha
    reads = np.ascontiguousarray(pool[rng.integers(0, len(pool), size=(n, L))])
    for i in range(0, n, 2 if not big else 20):
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        if len(k) <= L:
            o = int(rng.integers(0, L - len(k) + 1))
            reads[i, o:o + len(k)] = k
    flat = reads.reshape(-1)
    off = np.arange(n + 1, dtype=np.int64) * L
    img = Image.from_automaton(A)
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    if img.ppm_kernel(stride=L, dev_hay=d_hay.ptr.value, n_hay=n) != "stream4":
        raise SystemExit("trial %d: the plan is %r, not stream4 (alphabet %r, %d keys %d..%d, stride %d)"
                         % (trial, img.ppm_kernel(stride=L, dev_hay=d_hay.ptr.value, n_hay=n), bytes(alpha), len(keys), kmin, kmax, L))
    base = rng.integers(0, 1000, size=n).astype(np.int32) if rng.random() < 0.3 else None
    d_base = DeviceBuffer.from_numpy(base) if base is not None else None
    mo, oe, ov = O.batch(flat.tobytes(), off, 0)
    if base is not None:
        oe = oe + np.repeat(base, np.diff(mo)).astype(np.int32)
    for variant in (0, STREAM_ANY):
        sc = Scanner(img)
        sc.scan(d_hay, n * L, n, stride=L, dev_index_base=d_base, variant=variant)
        moff, e, v, _ = sc.fetch()
        if not (np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)):
            raise SystemExit("MISMATCH trial %d variant %#x alphabet %r keys %d (%d..%d) n %d L %d foreign %s"
                             % (trial, variant, bytes(alpha), len(keys), kmin, kmax, n, L, foreign))
    total = len(oe)
    if rng.random() < 0.5:
        # the offsets form: the same reads cut to ragged lengths, back to back (k_ppm_start_bits, the limits from the start bitmap, k_ppm_gather_pos<true>)
        lo = int(rng.choice([0, 1, 8, max(1, L // 2)]))
        lens = rng.integers(lo, L + 1, size=n, dtype=np.int64)
        if rng.random() < 0.3:
            lens[rng.integers(0, n, size=max(1, n // 7))] = 0         # runs of empty haystacks
        keep = np.arange(L, dtype=np.int64)[None, :] < lens[:, None]
        flat2 = np.ascontiguousarray(reads[keep])
        off2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        if len(flat2) == 0:
            return total
        d_hay2 = DeviceBuffer.from_numpy(flat2, pad=64)
        d_off2 = DeviceBuffer.from_numpy(off2)
        if img.ppm_kernel(stride=0, has_offsets=True, min_hay_len=8, dev_hay=d_hay2.ptr.value, n_hay=n) != "stream4":
            raise SystemExit("trial %d: the plan of the offsets batch is not stream4" % trial)
        mo2, oe2, ov2 = O.batch(flat2.tobytes(), off2, 0)
        if base is not None:
            oe2 = oe2 + np.repeat(base, np.diff(mo2)).astype(np.int32)
        for variant in (0, (1 << 24) | (1 << 28)):                   # k_ppm_stream4's offsets form; k_ppm_scan (any lengths)
            sc = Scanner(img)
            sc.scan(d_hay2, len(flat2), n, dev_off=d_off2, dev_index_base=d_base, min_hay_len=8 if variant == 0 else 0, variant=variant)
            moff, e, v, _ = sc.fetch()
            if not (np.array_equal(moff, mo2) and np.array_equal(e, oe2) and np.array_equal(v, ov2)):
                raise SystemExit("MISMATCH (offsets) trial %d variant %#x alphabet %r keys %d (%d..%d) n %d L %d lo %d foreign %s"
                                 % (trial, variant, bytes(alpha), len(keys), kmin, kmax, n, L, lo, foreign))
        total += len(oe2)
    return total


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, trials, matches = time.time(), 0, 0
    while time.time() - t0 < seconds:
        matches += one_case(rng, trials)
        trials += 1
    print("fuzz_stream4 ok: %d cases, %d matches, %.0f s" % (trials, matches, time.time() - t0))


if __name__ == "__main__":
    main()
