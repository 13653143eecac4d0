#!/usr/bin/env python3
"""Kernel-variant microbenchmark on one GPU (torch-free; talks to libacx only).

    python tools/microbench.py --variants 0,1,256,257 --reads 1000000

Prints one JSON line per variant with the median per-kernel HIP-event times.  Variant
encoding: see acx_launch_walk_all in pyahocorasick_amd/csrc/acx_kernels.hip.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyahocorasick_amd as acx  # noqa: E402
from pyahocorasick_amd.device import DeviceBuffer, Image, Scanner  # noqa: E402
from pyahocorasick_amd.workloads import dna_keys, dna_reads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--keys", type=int, default=100_000)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--mode", default="iter")
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--alphabet", default="dna", choices=["dna", "alnum"])
    ap.add_argument("--layout", default="stride", choices=["stride", "offsets", "one"],
                    help="stride: fixed-length reads (direct path); offsets: same reads through a device "
                         "offsets array (chunked path); one: the whole buffer as ONE haystack (chunk+halo)")
    args = ap.parse_args()

    t0 = time.time()
    if args.alphabet == "dna":
        keys = dna_keys(args.keys, seed=0)
    else:   # SURVEY §8(d) secondary "no-match" variant: 62-symbol keys, ACGT reads (shallow walk)
        import random
        rng = random.Random(0)
        al = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
        ks = set()
        while len(ks) < args.keys:
            ks.add("".join(rng.choice(al) for _ in range(rng.randint(8, 32))).encode())
        keys = sorted(ks)
        rng.shuffle(keys)
    A = acx.Automaton(acx.STORE_INTS)
    for i, k in enumerate(keys):
        A.add_word(k, i)
    A.make_automaton()
    img = Image.from_automaton(A)
    reads = dna_reads(keys if args.alphabet == "dna" else [], args.reads, args.read_len, seed=1)
    n, L = reads.shape
    d_hay = DeviceBuffer.from_numpy(reads.reshape(-1), pad=64)
    print(json.dumps({"setup_s": round(time.time() - t0, 2), "states": img.num_states, "classes": img.num_classes,
                      "image_mb": round(img.nbytes / 1e6, 1), "reads": n, "read_len": L}), flush=True)
    mode = acx.ACX_SCAN_ALL if args.mode == "iter" else acx.ACX_SCAN_LONG
    sc = Scanner(img)
    d_off = None
    n_items, stride = n, L
    if args.layout == "offsets":
        d_off = DeviceBuffer.from_numpy(np.arange(n + 1, dtype=np.int64) * L)
        stride = 0
    elif args.layout == "one":
        d_off = DeviceBuffer.from_numpy(np.array([0, n * L], dtype=np.int64))
        n_items, stride = 1, 0
    ref_total = None
    for v in [int(x) for x in args.variants.split(",")]:
        ts = {"walk": [], "scan": [], "expand": [], "total": []}
        total = 0
        for _ in range(args.reps):
            total = sc.scan(d_hay, n * L, n_items, dev_off=d_off, stride=stride, mode=mode, timing=True, variant=v)
            t = sc.timing_ms()
            for k in ts:
                ts[k].append(t[k])
        if ref_total is None and not (v >> 8) & 1:
            ref_total = total
        med = {k: round(float(np.median(x)), 4) for k, x in ts.items()}
        H = n * L
        print(json.dumps({"variant": v, "matches": total, "matches_ok": (total == ref_total) or bool((v >> 8) & 1),
                          "ms": med, "min_walk_ms": round(min(ts["walk"]), 4),
                          "walk_GBps_haystack": round(H / med["walk"] / 1e6, 1),
                          "total_GBps_haystack": round(H / med["total"] / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
