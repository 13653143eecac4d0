"""stand-alone repro of tests/test_gpu_long.py::test_random_dictionaries_fixed_stride_and_offsets[seed] with the differences printed"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import os
from pyahocorasick_amd import _lib
if os.environ.get("ACX_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["ACX_LIB"])
import pyahocorasick_amd as acx
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from helpers import build_pair
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(500 + seed)
alpha = np.frombuffer([b"ACGT", b"ab", b"abcdefghijklmnopqrstuvwxyz ", bytes(range(256))][seed % 4], dtype=np.uint8)
n_keys = int(rng.choice([30, 300, 3000]))
kmax = int(rng.choice([6, 12, 30]))
keys = list({bytes(rng.choice(alpha, size=int(k))) for k in rng.integers(1 if seed % 2 else 3, kmax + 1, size=n_keys)})
long_key = bytes(rng.choice(alpha, size=20))
keys = list(dict.fromkeys(keys + [long_key, long_key[:9], long_key[3:12], long_key[5:], long_key[7:9]]))
vals = [int(x) for x in rng.integers(-2**31, 2**31, size=len(keys))]
A, O = build_pair(keys, vals)
n, L = 3000, int(rng.choice([24, 100, 151]))
reads = alpha[rng.integers(0, len(alpha), size=(n, L))]
for i in range(0, n, 3):
    k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
    if len(k) <= L:
        o = int(rng.integers(0, L - len(k) + 1)); reads[i, o:o + len(k)] = k
reads[1, :20] = np.frombuffer(long_key, dtype=np.uint8)
flat = np.ascontiguousarray(reads.reshape(-1))
off = np.arange(n + 1, dtype=np.int64) * L
print("keys", len(keys), "kmax", kmax, "L", L)
img = Image.from_automaton(A)
d_hay = DeviceBuffer.from_numpy(flat, pad=64)
mo, oe, ov = O.batch(flat.tobytes(), off, 1)
sc = Scanner(img)
sc.scan(d_hay, len(flat), n, mode=acx.ACX_SCAN_LONG, variant=0, stride=L)
moff, e, v, _ = sc.fetch()
ok = np.array_equal(moff, mo) and np.array_equal(e, oe) and np.array_equal(v, ov)
print("ok", ok, "totals", moff[-1], mo[-1])
if not ok:
    bad = np.flatnonzero(np.diff(moff) != np.diff(mo))
    print("haystacks with other counts:", bad[:20], len(bad))
    hs = list(bad[:4])
    if not hs:
        d = np.flatnonzero((e != oe) | (v != ov))
        print("diff records", len(d), d[:10])
        hs = sorted(set(int(np.searchsorted(mo, x, side="right") - 1) for x in d[:4]))
    sys.path.insert(0, "/root/repo/tests")
    import test_iter_long_plan_cpu as T
    from oracle import orc
    dk, dv, reals, longest = T.long_dictionary(A)
    OD = orc.Oracle()
    for k, x in zip(dk, dv):
        OD.add_word(k, x)
    OD.make_automaton()
    for h in hs:
        hay = bytes(reads[h])
        recs = OD.iter(hay)
        print("haystack", h, hay)
        print("  got ", list(zip(e[moff[h]:moff[h+1]].tolist(), v[moff[h]:moff[h+1]].tolist())))
        print("  want", list(zip(oe[mo[h]:mo[h+1]].tolist(), ov[mo[h]:mo[h+1]].tolist())))
        print("  model", T.sweep_records_compact([a for a, _ in recs], [b for _, b in recs], reals, longest))
        print("  D records", [(a, (b & 0xFFFFFFFF) >> 30, ((b & 0xFFFFFFFF) >> 24) & 63, ((b & 0xFFFFFFFF) >> 18) & 63, b & 0x3FFFF) for a, b in recs])
