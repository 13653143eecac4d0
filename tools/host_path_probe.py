"""Host to host (acx_scan_host_ctx + acx_result_fetch_host) on config 2's batch, call by call, optionally with a
development build that prints where the time goes (tools/build_variant.sh NAME -DACX_HOST_TRACE with SRC=acx_capi.hip).
    python tools/host_path_probe.py [--lib build/variants/libacx_trace.so] [--calls 6] [--group-mb 32]
Prints per call: ms of acx_scan_host_ctx, ms of acx_result_fetch_host, GB/s of haystack; checks the records of the last
call against a device-resident scan of the same batch (count and checksum)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--calls", type=int, default=6)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--keys", type=int, default=100_000)
    ap.add_argument("--group-mb", type=float, default=0)
    args = ap.parse_args()
    os.environ.setdefault("ACX_WITH_TORCH", "1")
    from pyahocorasick_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    import pyahocorasick_amd as acx
    from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
    rng = np.random.default_rng(7)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    keys = list({bytes(alpha[rng.integers(0, 4, size=int(k))].tobytes()) for k in rng.integers(8, 33, size=args.keys)})
    A = acx.Automaton(acx.STORE_INTS)
    A.add_words(keys, range(len(keys)))
    A.make_automaton()
    img = Image.from_automaton(A)
    n, L = args.reads, args.read_len
    reads = alpha[rng.integers(0, 4, size=(n, L))]
    for i in range(0, n, 2):
        k = np.frombuffer(keys[int(rng.integers(0, len(keys)))], dtype=np.uint8)
        o = int(rng.integers(0, L - len(k) + 1))
        reads[i, o:o + len(k)] = k
    flat = np.ascontiguousarray(reads.reshape(-1))
    off = np.arange(n + 1, dtype=np.int64) * L
    lib = _lib.lib()
    if args.group_mb:
        lib.acx_set_host_group_bytes(int(args.group_mb * (1 << 20)))
    res = C.c_void_p()
    p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for c in range(args.calls):
        t0 = time.perf_counter()
        _lib.check(lib.acx_scan_host_ctx(img.handle, flat.ctypes.data, off.ctypes.data, n, None, None, None, 0, C.byref(res)))
        t1 = time.perf_counter()
        _lib.check(lib.acx_result_fetch_host(res, C.byref(p1), C.byref(p2), C.byref(p3)))
        t2 = time.perf_counter()
        print("call %d: scan_host_ctx %.3f ms, fetch_host %.3f ms, total %.3f ms = %.1f GB/s"
              % (c, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3, flat.size / (t2 - t0) / 1e9), flush=True)
    total = lib.acx_result_num_matches(res)
    moff = np.ctypeslib.as_array(C.cast(p1, C.POINTER(C.c_int64)), shape=(n + 1,)).copy()
    m = np.ctypeslib.as_array(C.cast(p2, C.POINTER(C.c_int32)), shape=(int(moff[-1]), 2)).copy()
    # the same batch, device-resident
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    sc = Scanner(img)
    sc.scan(d_hay, n * L, n, stride=L)
    moff2, e2, v2, _ = sc.fetch()
    ok = np.array_equal(moff, moff2) and np.array_equal(m[:, 0], e2) and np.array_equal(m[:, 1], v2)
    print("records: %d (result says %d), equal to the device-resident scan: %s" % (len(m), total, ok))
    if not ok:
        raise SystemExit(1)
    lib.acx_result_free(res)


if __name__ == "__main__":
    main()
