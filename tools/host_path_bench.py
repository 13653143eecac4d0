"""PCIe-inclusive rate of the host entry point (acx_scan_host: H2D + scan + D2H), config 2.
Reported in DESIGN.md §4; never the bench `value`."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pyahocorasick_amd import Automaton, STORE_INTS            # noqa: E402
from pyahocorasick_amd.workloads import dna_keys, dna_reads    # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    keys = dna_keys(100_000, seed=1)
    reads = dna_reads(keys, n_reads, 150, seed=2)
    A = Automaton(STORE_INTS)
    A.add_words(keys, range(len(keys)))
    A.make_automaton()
    data = np.ascontiguousarray(reads).reshape(-1)
    off = np.arange(n_reads + 1, dtype=np.int64) * 150
    A.scan_batch(data, off)                                    # warm-up: image upload, buffers
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        res = A.scan_batch(data, off)
        ts.append(time.perf_counter() - t0)
    t = min(ts)
    # the C-ABI alone (acx_scan_host + acx_result_fetch_host: pageable H2D, kernels, D2H into the result's pinned
    # buffers), without the copies the Python mirror makes of the fetched arrays
    import ctypes as C
    from pyahocorasick_amd._lib import lib, check
    img = A._ensure_image()
    tc = []
    for _ in range(5):
        t0 = time.perf_counter()
        check(lib().acx_scan_host(img.handle, 0, data.ctypes.data, off.ctypes.data, n_reads, None, None, C.byref(A._result)))
        p_off, p_m, p_f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().acx_result_fetch_host(A._result, C.byref(p_off), C.byref(p_m), C.byref(p_f)))
        tc.append(time.perf_counter() - t0)
    # what iter() / find_all / iter_batch call: no final states (acx_scan_host_ctx without contexts)
    tx = []
    for _ in range(5):
        t0 = time.perf_counter()
        check(lib().acx_scan_host_ctx(img.handle, data.ctypes.data, off.ctypes.data, n_reads, None, None, None, 0, C.byref(A._result)))
        p_off, p_m, p_f = C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(lib().acx_result_fetch_host(A._result, C.byref(p_off), C.byref(p_m), C.byref(p_f)))
        tx.append(time.perf_counter() - t0)
    print(json.dumps({"reads": n_reads, "c_abi_no_final_states_best_s": min(tx), "GBps_haystack_pcie_inclusive_c_abi_no_final_states": data.size / min(tx) / 1e9, "bytes": int(data.size), "matches": int(res.num_matches()),
                      "python_mirror_best_s": t, "GBps_haystack_pcie_inclusive_python_mirror": data.size / t / 1e9,
                      "c_abi_best_s": min(tc), "GBps_haystack_pcie_inclusive_c_abi": data.size / min(tc) / 1e9}))


if __name__ == "__main__":
    main()
