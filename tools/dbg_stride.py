import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from helpers import build_pair
stride = int(sys.argv[1]); mode = sys.argv[2]
rng = np.random.default_rng(stride)
a = np.frombuffer(b"ACGT", dtype=np.uint8)
lo, hi = (1, 12) if "short" in mode else (6, 12)
keys = list({bytes(rng.choice(a, size=int(k)).tobytes()) for k in rng.integers(lo, hi, size=400)})
A, O = build_pair(keys)
n = max(3, 20000 // stride)
reads = np.ascontiguousarray(a[rng.integers(0, 4, size=(n, stride))])
off = np.arange(n + 1, dtype=np.int64) * stride
img = Image.from_automaton(A)
d_hay = DeviceBuffer.from_numpy(reads.reshape(-1), pad=64)
base = rng.integers(0, 1000, size=n).astype(np.int32)
d_base = DeviceBuffer.from_numpy(base) if "base" in mode else None
sc = Scanner(img)
print("scan", stride, mode, flush=True)
sc.scan(d_hay, n * stride, n, stride=stride, dev_index_base=d_base, want_final_state="final" in mode)
moff, e, v, fin = sc.fetch()
mo, oe, ov = O.batch(reads.tobytes(), off, 0)
if d_base is not None:
    oe = oe + np.repeat(base, np.diff(mo)).astype(np.int32)
print("ok", np.array_equal(moff, mo), np.array_equal(e, oe), np.array_equal(v, ov), len(oe), flush=True)
bad = np.nonzero(moff != mo)[0]
print("bad", len(bad), bad[:12].tolist(), moff[bad[:12]].tolist(), "good", np.nonzero(moff == mo)[0][:12].tolist(), np.nonzero(moff == mo)[0][-12:].tolist(), flush=True)
