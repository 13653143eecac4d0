import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from helpers import build_pair
A, O = build_pair([b"he", b"her", b"hers", b"she"])
hay = b"_sherhershe_ ushers he"
off = np.array([0, len(hay)], dtype=np.int64)
img = Image.from_automaton(A)
d_hay = DeviceBuffer.from_numpy(np.frombuffer(hay, dtype=np.uint8), pad=64)
d_off = DeviceBuffer.from_numpy(off)
variant = int(sys.argv[1]); mhl = int(sys.argv[2]); fin = int(sys.argv[3])
print("plan", img.ppm_kernel(has_offsets=True, variant=variant, min_hay_len=mhl, dev_hay=d_hay.ptr.value, n_hay=1), flush=True)
sc = Scanner(img)
sc.scan(d_hay, len(hay), 1, dev_off=d_off, variant=variant, min_hay_len=mhl, want_final_state=bool(fin))
moff, e, v, f = sc.fetch()
print("ok", list(zip(e.tolist(), v.tolist())) == O.iter(hay), moff.tolist(), flush=True)
