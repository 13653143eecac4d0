"""stand-alone repro for k_ppm_stream4's deeper walks: a small four-letter dictionary with keys beyond ten letters, fixed stride; the first
haystack whose records differ from the oracle's is printed.    python tools/dbg_deep.py [n_reads] [stride] [n_keys] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from pyahocorasick_amd.workloads import dna_workload
from helpers import build_pair
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
nk = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 0
keys, reads = dna_workload(nk, n, L, seed=seed)
A, O = build_pair(keys)
img = Image.from_automaton(A)
d_hay = DeviceBuffer.from_numpy(reads.reshape(-1), pad=64)
print("plan", img.ppm_kernel(stride=L, dev_hay=d_hay.ptr.value, n_hay=n), flush=True)
sc = Scanner(img)
tot = sc.scan(d_hay, n * L, n, stride=L)
moff, e, v, _ = sc.fetch()
off = np.arange(n + 1, dtype=np.int64) * L
mo, oe, ov = O.batch(reads.tobytes(), off, 0)
print("total", tot, "oracle", mo[-1], "offsets ok", np.array_equal(moff, mo), "records ok", len(e) == len(oe) and np.array_equal(e, oe) and np.array_equal(v, ov), flush=True)
klen = np.array([len(k) for k in keys])
for h in range(n):
    g = list(zip(e[moff[h]:moff[h + 1]].tolist(), v[moff[h]:moff[h + 1]].tolist()))
    w = list(zip(oe[mo[h]:mo[h + 1]].tolist(), ov[mo[h]:mo[h + 1]].tolist()))
    if g != w:
        print("first bad haystack", h, "got", len(g), "want", len(w))
        print(" got ", [(a, b, int(klen[b]) if 0 <= b < len(keys) else -1) for a, b in g][:24])
        print(" want", [(a, b, int(klen[b])) for a, b in w][:24])
        break
