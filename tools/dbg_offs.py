"""stand-alone repro for k_ppm_stream4 on OFFSETS batches: four-letter keys, ragged reads (lengths U[lo, hi], optionally empty ones and bytes
of no key), every offset and record against the oracle.    python tools/dbg_offs.py [n_reads] [lo] [hi] [n_keys] [seed] [p_other] [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from pyahocorasick_amd.workloads import dna_workload
from helpers import build_pair
a = sys.argv[1:]
n = int(a[0]) if len(a) > 0 else 4096
lo = int(a[1]) if len(a) > 1 else 100
hi = int(a[2]) if len(a) > 2 else 150
nk = int(a[3]) if len(a) > 3 else 2000
seed = int(a[4]) if len(a) > 4 else 0
p_other = float(a[5]) if len(a) > 5 else 0.0
variant = int(a[6]) if len(a) > 6 else 0
keys, reads = dna_workload(nk, n, hi, seed=seed)
rng = np.random.default_rng(seed + 7)
lens = rng.integers(lo, hi + 1, size=n, dtype=np.int64)
keep = np.arange(hi, dtype=np.int64)[None, :] < lens[:, None]
flat = np.ascontiguousarray(reads[keep])
if p_other > 0:
    m = rng.random(len(flat)) < p_other
    flat[m] = ord("N")
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
A, O = build_pair(keys)
img = Image.from_automaton(A)
d_flat = DeviceBuffer.from_numpy(flat, pad=64)
d_off = DeviceBuffer.from_numpy(off)
mhl = max(int(lens.min()), 0)
print("plan", img.ppm_kernel(stride=0, has_offsets=True, dev_hay=d_flat.ptr.value, n_hay=n, min_hay_len=mhl), "min_hay_len", mhl, flush=True)
sc = Scanner(img)
tot = sc.scan(d_flat, len(flat), n, dev_off=d_off, min_hay_len=mhl, variant=variant)
moff, e, v, _ = sc.fetch()
mo, oe, ov = O.batch(flat.tobytes(), off, 0)
ok_o = np.array_equal(moff, mo); ok_r = len(e) == len(oe) and np.array_equal(e, oe) and np.array_equal(v, ov)
print("total", tot, "oracle", mo[-1], "offsets ok", ok_o, "records ok", ok_r, flush=True)
if not (ok_o and ok_r):
    for h in range(n):
        g = list(zip(e[moff[h]:moff[h + 1]].tolist(), v[moff[h]:moff[h + 1]].tolist()))
        w = list(zip(oe[mo[h]:mo[h + 1]].tolist(), ov[mo[h]:mo[h + 1]].tolist()))
        if g != w or moff[h] != mo[h]:
            print("first bad haystack", h, "off", off[h], "len", lens[h], "moff", moff[h], mo[h], "got", len(g), "want", len(w))
            print(" got ", g[:16]); print(" want", w[:16])
            break
