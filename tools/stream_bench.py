#!/usr/bin/env python3
"""Streams on the position-parallel kernels: S streams arrive in chunks of --chunk bytes; every chunk is
scanned with the last longest_word - 1 bytes of its stream in front of it as context (acx_scan_params.dev_skip).  Prints
the rate of such a chunk batch beside the rate of the same bytes scanned as independent haystacks (no context) and beside
the old way (carried states: the serial walks)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyahocorasick_amd as acx
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner
from pyahocorasick_amd.workloads import dna_keys

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=2048)
ap.add_argument("--chunk", type=int, default=65536)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
keys = dna_keys(100_000, seed=0)
A = acx.Automaton(acx.STORE_INTS)
A.add_words(keys, range(len(keys)))
A.make_automaton()
img = Image.from_automaton(A)
halo = max(len(k) for k in keys) - 1
rng = np.random.default_rng(1)
S, CH = a.streams, a.chunk
L = halo + CH                                   # a slot: context, then the chunk
buf = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(S, L))]
d_hay = DeviceBuffer.from_numpy(np.ascontiguousarray(buf).reshape(-1), pad=64)
d_skip = DeviceBuffer.from_numpy(np.full(S, halo, dtype=np.int32))
d_zero = DeviceBuffer.from_numpy(np.zeros(S, dtype=np.int32))
out = {"streams": S, "chunk": CH, "context": halo, "bytes_per_batch": S * CH}
for name, kw in (("with_context_stream_kernel", dict(stride=L, dev_skip=d_skip)),
                 ("no_context_same_bytes", dict(stride=L)),
                 ("carried_states_serial_walks", dict(stride=L, dev_init_state=d_zero, want_final_state=True))):
    sc = Scanner(img)
    ts = []
    for _ in range(a.reps):
        sc.scan(d_hay, S * L, S, timing=True, **kw)
        ts.append(sc.timing_ms()["total"])
    ms = float(np.median(ts))
    out[name] = {"ms": round(ms, 4), "GBps_of_chunk_bytes": round(S * CH / ms / 1e6, 1), "matches": sc.num_matches()}
out["context_vs_no_context"] = round(out["with_context_stream_kernel"]["GBps_of_chunk_bytes"] / out["no_context_same_bytes"]["GBps_of_chunk_bytes"], 3)
print(json.dumps(out))
