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
    ap.add_argument("--alphabet", default="dna", choices=["dna", "alnum", "text", "snort"],
                    help="dna/alnum: fixed-length reads; text: config-3 style corpus as ONE haystack; "
                         "snort: config-4 style signatures over ragged packets")
    ap.add_argument("--bytes", type=int, default=256 << 20, help="haystack bytes for text/snort")
    ap.add_argument("--check", type=int, default=0,
                    help="compare N haystacks (a deterministic sample; for ONE long haystack: N chunks of 64 KiB) of the "
                         "first variant's result with the oracle (oracle/ac_oracle.c): `sample_ok` in its line")
    ap.add_argument("--min-hay-len", type=int, default=-1, help="offsets batches: acx_scan_params.min_hay_len (-1: the true minimum)")
    ap.add_argument("--layout", default="stride", choices=["stride", "offsets", "one"],
                    help="stride: fixed-length reads (direct path); offsets: same reads through a device "
                         "offsets array (chunked path); one: the whole buffer as ONE haystack (chunk+halo)")
    ap.add_argument("--flatten-flags", type=int, default=0, help="layout options of the flat image (acx_flatten_ex, ACX_FLATTEN_*; 32: 12-byte hot cells)")
    ap.add_argument("--lib", default=None, help="another build of libacx.so (development variants): loaded instead of the one in the package")
    args = ap.parse_args()
    if args.lib:
        from pyahocorasick_amd import _lib
        _lib.LIB_PATH = os.path.abspath(args.lib)

    t0 = time.time()
    pre_off = None
    if args.alphabet == "dna":
        keys = dna_keys(args.keys, seed=0)
    elif args.alphabet == "text":
        from pyahocorasick_amd.workloads import text_corpus, text_keys, text_vocab
        vocab = text_vocab(1_000_000 if args.keys >= 100_000 else 10 * args.keys, seed=2)
        keys = text_keys(vocab, args.keys, seed=3)
    elif args.alphabet == "snort":
        from pyahocorasick_amd.workloads import packet_payloads, snort_signatures
        keys = snort_signatures(args.keys, seed=5)
    else:   # SURVEY §8(d) secondary "no-match" variant: 62-symbol keys, ACGT reads (shallow walk)
        import random
        rng = random.Random(0)
        al = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789"
        ks = set()
        while len(ks) < args.keys:
            ks.add("".join(rng.choice(al) for _ in range(rng.randint(8, 32))).encode())
        keys = sorted(ks)
        rng.shuffle(keys)
    t_gen = time.time() - t0
    t1 = time.time()
    A = acx.Automaton(acx.STORE_INTS)
    A.flatten_flags = args.flatten_flags
    A.add_words(keys, range(len(keys)))
    t_add = time.time() - t1; t1 = time.time()
    A.make_automaton()
    t_make = time.time() - t1; t1 = time.time()
    img = Image.from_automaton(A)
    t_img = time.time() - t1
    if args.alphabet == "text":
        flat = text_corpus(vocab, args.bytes, seed=4)
        n, L = 1, len(flat)
        pre_off = np.array([0, L], dtype=np.int64)
        args.layout = "pre"
    elif args.alphabet == "snort":
        flat, pre_off = packet_payloads(keys, args.bytes, seed=6)
        n, L = len(pre_off) - 1, 0
        args.layout = "pre"
    else:
        reads = dna_reads(keys if args.alphabet == "dna" else [], args.reads, args.read_len, seed=1)
        n, L = reads.shape
        flat = reads.reshape(-1)
    total_bytes = int(flat.size)
    d_hay = DeviceBuffer.from_numpy(flat, pad=64)
    print(json.dumps({"setup_s": round(time.time() - t0, 2), "gen_s": round(t_gen, 2), "add_s": round(t_add, 2),
                      "make_automaton_s": round(t_make, 2), "flatten_upload_s": round(t_img, 2),
                      "states": img.num_states, "classes": img.num_classes,
                      "image_mb": round(img.nbytes / 1e6, 1), "haystacks": n, "bytes": total_bytes}), flush=True)
    mode = acx.ACX_SCAN_ALL if args.mode == "iter" else acx.ACX_SCAN_LONG
    sc = Scanner(img)
    d_off = None
    n_items, stride = n, L
    if args.layout == "pre":
        d_off = DeviceBuffer.from_numpy(pre_off)
        stride = 0
    elif args.layout == "offsets":
        d_off = DeviceBuffer.from_numpy(np.arange(n + 1, dtype=np.int64) * L)
        stride = 0
    elif args.layout == "one":
        d_off = DeviceBuffer.from_numpy(np.array([0, n * L], dtype=np.int64))
        n_items, stride = 1, 0
    ref_total = None
    host_off = pre_off if pre_off is not None else (np.arange(n + 1, dtype=np.int64) * L if args.layout != "one" else np.array([0, n * L], dtype=np.int64))
    mhl = args.min_hay_len if args.min_hay_len >= 0 else (int(np.diff(host_off).min()) if d_off is not None else 0)

    def check_sample(sc):
        """a deterministic sample of the result against the oracle (never the run against itself)"""
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import orc
        O = orc.Oracle()
        for i, k in enumerate(keys):
            O.add_word(k, i)
        O.make_automaton()
        moff, e, vv, _ = sc.fetch()
        m = 0 if args.mode == "iter" else 1
        nh = len(host_off) - 1
        if nh > 1:
            pick = np.unique(np.linspace(0, nh - 1, min(args.check, nh)).astype(np.int64))
            soff = np.concatenate([[0], np.cumsum(host_off[pick + 1] - host_off[pick])]).astype(np.int64)
            sdata = np.concatenate([flat[host_off[h]:host_off[h + 1]] for h in pick])
            mo, oe, ov = O.batch_records(sdata, soff, m)
            return all(np.array_equal(e[moff[h]:moff[h + 1]], oe[mo[k]:mo[k + 1]]) and np.array_equal(vv[moff[h]:moff[h + 1]], ov[mo[k]:mo[k + 1]])
                       for k, h in enumerate(pick))
        if m:
            return None                                   # iter_long of one long haystack cannot be cut into chunks
        CH, ctxlen = 1 << 16, max(len(k) for k in keys) - 1
        starts = np.unique(np.linspace(0, max(0, len(flat) - CH), args.check).astype(np.int64))
        ctx = [max(0, int(s0) - ctxlen) for s0 in starts]
        soff = np.concatenate([[0], np.cumsum([min(len(flat), int(s0) + CH) - c for s0, c in zip(starts, ctx)])]).astype(np.int64)
        sdata = np.concatenate([flat[c:min(len(flat), int(s0) + CH)] for s0, c in zip(starts, ctx)])
        mo, oe, ov = O.batch_records(sdata, soff, 0)
        ok = True
        for k, (s0, c) in enumerate(zip(starts, ctx)):
            we = oe[mo[k]:mo[k + 1]].astype(np.int64) + c
            keep = we >= s0
            lo, hi = np.searchsorted(e, s0, side="left"), np.searchsorted(e, min(len(flat), s0 + CH), side="left")
            ok = ok and np.array_equal(e[lo:hi], we[keep]) and np.array_equal(vv[lo:hi], ov[mo[k]:mo[k + 1]][keep])
        return bool(ok)

    sample_ok = None
    for v in [int(x) for x in args.variants.split(",")]:
        ts = {"walk": [], "scan": [], "expand": [], "total": []}
        total = 0
        for _ in range(args.reps):
            total = sc.scan(d_hay, total_bytes, n_items, dev_off=d_off, stride=stride, mode=mode, timing=True, variant=v, min_hay_len=mhl)
            t = sc.timing_ms()
            for k in ts:
                ts[k].append(t[k])
        if ref_total is None and not (v >> 8) & 1:
            ref_total = total
            if args.check:
                sample_ok = check_sample(sc)
        med = {k: round(float(np.median(x)), 4) for k, x in ts.items()}
        H = total_bytes
        print(json.dumps({"variant": v, "matches": total, "same_total_as_first_variant": (total == ref_total) or bool((v >> 8) & 1),
                          "sample_ok_vs_oracle": sample_ok,
                          "ms": med, "min_walk_ms": round(min(ts["walk"]), 4),
                          "walk_GBps_haystack": round(H / med["walk"] / 1e6, 1),
                          "total_GBps_haystack": round(H / med["total"] / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
