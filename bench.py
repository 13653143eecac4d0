#!/usr/bin/env python3
"""bench.py — headline benchmark of the batch Aho-Corasick scan on MI355X.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
100,000 unique ACGT keys of length U[8,32] (seed 0) -> one flattened automaton;
batches of 1,000,000 x 150 B DNA-style reads (every even read has a planted key), resident in HBM
before the timed region.  A pass of the hot path scans ONE batch (scan kernel + gather -> match
records and per-read offsets in HBM); `ms_per_step` is the time of one such pass.  A pass is a third
of a millisecond, so each of the K driver steps is R passes (`inner_repeats`, chosen so that the
timed region lasts >= --min-timed-ms: a 6 ms region would let one barrier's skew decide a multi-GPU
line): the timed region is K x R passes, `value` = all bytes scanned / its duration.  The loop ROTATES
over --batches distinct batches (default 4 x 150 MB = 600 MB, beyond the 256 MiB Infinity Cache), so the
haystack bytes of a pass come from HBM, not from a cache that the previous pass filled.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

    --mode iter_long           config 5 (AutomatonSearchIterLong on the same workload)
    --workload c3              config-3 shape on one GPU: 100k multi-word text keys, one text shard,
                               every step scans a different quarter of it as ONE haystack
    --workload c4              config-4 shape on one GPU: Snort-style byte signatures (--keys, default
                               1,000,000), ragged packets 64..1500 B
    --scaling strong           N > 1: ONE fixed corpus (c2: the read batches, c3: the text, c4: the packets) is cut
                               into N contiguous shards (reads by count, packets by BYTES, text with a
                               longest_word-1 halo); default "weak": every rank scans its own batches of the full size
    --workload c2o | c2k       config 2 as the GENERAL stream kernel takes it: the reads as an offsets batch of ragged
                               lengths U[100,150]; keys of 8-64 letters
    --configs all|none|LIST    N = 1, default workload only (LIST: a comma list of their names): after the headline measurement the other named
                               single-GPU configurations are measured in the same run and reported under
                               "configs": c5 (iter_long, same automaton and batches), c2_offsets, c2_long_keys, c3 and
                               c4 (two batches each) — each with value, ms_per_step, roofline and a sample-limited
                               cpu_baseline

N > 1: one process per GPU; rank 0 builds + flattens the automaton and the flat image is replicated
with ONE RCCL broadcast; no data-path collective; time = max over ranks, value = all ranks' bytes /
time.  Prints one JSON line (rank 0).  `roofline` and `cpu_baseline` are explained in DESIGN.md.
"""
import argparse
import hashlib
import json
import os
import sys
import time

# An asynchronous scan finishes on a side stream (libacx: the record gather runs under the next batch's scan kernel).  The
# HIP runtime maps streams onto 4 hardware queues per process by default; with RCCL initialised in the same process its
# streams take some, the side stream then shares a queue with the scan stream and the overlap is gone (measured under
# torchrun, one rank: 0.426 -> 0.408 ms per step with 8 queues).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# the sources a dominant kernel is made of: a PMC traffic figure (profiles/traffic.json) is tied to their hash
KERNEL_SOURCES = {
    "k_ppm_stream4": ("acx_ppm_stream4.hip", "acx_ppm_device.h", "acx_ppm_layout.h", "acx_kernels.h"),
    "k_ppm_stream": ("acx_ppm_kernels.hip", "acx_ppm_device.h", "acx_ppm_layout.h", "acx_kernels.h"),
    "k_ppm_scan": ("acx_ppm_kernels.hip", "acx_ppm_device.h", "acx_ppm_layout.h", "acx_kernels.h"),
    "k_walk_long_sel": ("acx_kernels.hip", "acx_kernels.h"),
    "k_walk_itop": ("acx_kernels.hip", "acx_kernels.h"),
    "k_walk_all": ("acx_kernels.hip", "acx_kernels.h"),
}


ALL_CONFIGS = ("c5_iter_long", "c2_offsets", "c2_long_keys", "c3", "c4")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)          # (K driver steps of R passes each: --inner-repeats)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", choices=["c2", "c2o", "c2k", "c3", "c4"], default="c2",
                    help="c2o / c2k: config 2 as the general stream kernel takes it — the reads as an offsets batch of ragged lengths, keys of 8-64 letters")
    ap.add_argument("--keys", type=int, default=None, help="dictionary size (default: 100,000; c4: 1,000,000)")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--batches", type=int, default=4, help="distinct pre-staged batches the timed loop rotates over")
    ap.add_argument("--batch-mb", type=int, default=512, help="c3/c4: MiB of haystack per batch")
    ap.add_argument("--mode", choices=["iter", "iter_long"], default="iter")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--scan-streams", type=int, default=3,
                    help="streams the scans of consecutive steps alternate over (position-parallel scans; the batches of consecutive steps are "
                         "independent).  Default 3 (with --pipeline 3; 2 in round 4's two-stream leg): the blocks of step k + 1 start on the CUs step k's finished blocks leave, so the uneven end "
                         "of one launch and the launch seam (step - kernel = 35-41 us on one stream) are hidden (512 -> 546 GB/s).  Launch spans "
                         "then OVERLAP, and no per-launch duration describes the kernel any more (rocprofv3: 339 us per launch, HIP events: "
                         "0.2675 ms, tools/r4_clock_check.sh): `roofline` is therefore computed from the UNION of the dominant kernel's spans per "
                         "launch — the timed region / launches, an upper bound of it — which tools/roofline_check.py recomputes from a rocprofv3 "
                         "kernel trace; `roofline.kernel_alone` holds the launch measured with the chip to itself.  1: one stream — every clock "
                         "agrees on the launch duration (the regime of rounds 1-4 and of profiles/r5_c2_kernel_stats.md)")
    ap.add_argument("--one-stream-leg", action="store_true",
                    help="after the headline measurement, time the same steps once more with every scan on ONE stream and report it as "
                         "`one_scan_stream` (value, step, the kernel's launch duration by HIP events inside that timed region).  Not part of the "
                         "default command: its launches would mix into the averages of a rocprofv3 run of that command")
    ap.add_argument("--inner-repeats", type=int, default=0,
                    help="R: every one of the K driver steps is R passes of the hot path, each over one batch (the timed region is K x R batch "
                         "scans; ms_per_step = region / (K x R) = one batch scan, `value` = bytes / region).  0 (default): R is chosen from a "
                         "short calibration so that the region lasts --min-timed-ms whatever --steps says — a step is a third of a "
                         "millisecond, and a region of 6 ms would let one barrier's skew decide an 8-GPU line")
    ap.add_argument("--min-timed-ms", type=float, default=250.0, help="length of the timed region that the automatic --inner-repeats aims at")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="result objects kept in flight per GPU (ACX_SCAN_ASYNC): the host queues step i+1 and reads "
                         "the counters of step i-1 while step i runs; on ONE stream the kernels of consecutive steps "
                         "still run strictly one after the other.  1 = one synchronous call per step.  Default 3 (with 3 scan "
                         "streams): with 2 x 2 the host queues scan k + 2 only when gather k has completed, and the union of the scan "
                         "kernel's launch spans covered 90 %% of the timed region (profiles/r5b_c2_spans.json: 239.8 of 267.3 us per "
                         "step); profiles/r5c_pipeline_sweep.txt: 2 x 2 565.7, 3 x 3 580.1, 4 x 2 536.3, 4 x 4 557.9, 6 x 2 567.4, 6 x 3 578.4 GB/s")
    ap.add_argument("--long-depth", type=int, default=6,
                    help="iter_long: result objects in flight AND scan streams (instead of --pipeline / --scan-streams).  A step of iter_long is a scan kernel and "
                         "four smaller ones behind it (gather, sweep, placement) that find CUs only in the tails of the scan kernels: with more batches in "
                         "flight the tails are fuller.  tools/r6_depth.sh (alone in its process): 3 -> 248.9, 4 -> 255.5, 5 -> 257.4, 6 -> 258.4 GB/s; the "
                         "headline and c2_offsets are best at 3 (629.8 / 529.7 against 529-564 / 491-500 at 4-6).  Behind the headline: ten lines of the default "
                         "command within 257.6-259.9 (profiles/r6_line_runs.txt).  0: as --pipeline / --scan-streams.  (Before the gathers and the sweep were "
                         "made safe against the records of a scan whose pool ran out — DESIGN.md 7, tools/r6_crash.sh — config 5 moved by +-4 %% from run to "
                         "run behind the headline, and with 5 or 6 results in flight four runs of about a hundred died of a GPU memory fault there.)")
    ap.add_argument("--event-every", type=int, default=4,
                    help="bracket the dominant kernel by HIP events in every N-th timed step (0: in none).  Two event records cost "
                         "the stream about 19 us of idle time per step they are in (config 2: 442 GB/s with events in every step, "
                         "468 in every fourth, 472 in none)")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the host-to-host leg (end_to_end_GBps): it scans the batch in groups, whose short launches would mix into "
                         "the per-kernel averages of a rocprofv3 --stats run of this command")
    ap.add_argument("--cpu-sample-reads", type=int, default=None,
                    help="haystacks timed on the CPU baseline legs (default: the whole first batch; 0 disables)")
    ap.add_argument("--verify", action="store_true", help="check the first batch's GPU output against the oracle (all records)")
    ap.add_argument("--configs", default=None,
                    help="the other named single-GPU configurations in the same run: all | none | a comma list of their names "
                         "(c5_iter_long,c2_offsets,c2_long_keys,c3,c4); default: all for the default command on one GPU")
    ap.add_argument("--dummy-streams", type=int, default=0,
                    help="make (and use once) this many HIP streams before anything else — what RCCL's and a caller's own streams do to the order in "
                         "which the scan and side streams get their hardware queues (include/acx.h, ACX_SCAN_ASYNC; tools/r6_queues.sh)")
    ap.add_argument("--verbose", action="store_true",
                    help="print the FULL line (every field, the prose ones included: > 15 KB for the default command).  Default: the compact "
                         "line (< 7 KB: the driver keeps an 8 KB tail of the output) — the contract's fields, `roofline` and `cpu_baseline` of "
                         "every configuration, and {name: [GB/s, roofline.frac]} of all of them in `config.all`")
    ap.add_argument("--full-json", default=None, help="also write the full line to this file (profiles/r6_bench_full.json is one)")
    ap.add_argument("--lib", default=None, help="another build of libacx.so (development A/B, tools/build_variant.sh) instead of the package's")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the ranks come up over gloo, the blob is broadcast and validated, every rank stages its shard on the host as "
                         "the real run does, byte and haystack counts are reduced and printed as one JSON line (tests/test_parallel_cpu.py)")
    ap.add_argument("--launch-check", action="store_true",
                    help="start the ranks (gloo, no GPU), agree on the world size, print {\"n_gpus\": N, \"launch_check\": true} and exit: "
                         "tests/test_parallel_cpu.py runs the self-launch path of --gpus N with it")
    args = ap.parse_args()
    if args.configs not in (None, "all", "none") and set(args.configs.split(",")) - set(ALL_CONFIGS):
        ap.error("--configs: all, none or a comma list of %s" % ", ".join(ALL_CONFIGS))
    return args


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks here (one process per GPU,
    rendezvous on 127.0.0.1) and hand their exit code back.  Under torch.distributed.run (WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def launch_check(args):
    """--launch-check: the ranks come up (gloo), agree on the world size, rank 0 prints one JSON line"""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        seen = int(t.item())
        rank = dist.get_rank()
        dist.barrier()
        dist.destroy_process_group()
    else:
        seen, rank = 1, 0
    if seen != args.gpus:
        raise SystemExit("launch check: %d ranks answered, --gpus %d" % (seen, args.gpus))
    if rank == 0:
        print(json.dumps({"n_gpus": seen, "launch_check": True}), flush=True)


def dry_run(args):
    """--dry-run: everything of a multi-rank run that needs no GPU, over gloo — the ranks come up and agree on the world
    size, rank 0 builds and flattens the dictionary, ONE broadcast replicates the blob (checksum verified on every rank),
    every rank stages its shard of --batches batches on the host exactly as the real run does (weak or strong), and the byte
    and haystack counts are reduced as the real line reduces them.  Nothing is scanned (the product has no CPU scan); rank 0
    prints {"dry_run": true, "n_gpus", "bytes_total", "corpus_bytes_per_batch", "per_rank": [...]}.
    tests/test_parallel_cpu.py runs it with small sizes and checks shard cover / disjointness against the oracle."""
    import hashlib as hl
    sys.stdout.flush()
    json_fd = os.dup(1)                                      # (gloo reports its connections on fd 1: the JSON line gets a descriptor of its own)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        dist.init_process_group("gloo")
    import pyahocorasick_amd as acx
    from pyahocorasick_amd import _lib
    from pyahocorasick_amd.parallel import broadcast_blob
    import ctypes as C
    n_keys = args.keys or (1_000_000 if args.workload == "c4" else 100_000)
    keys, vocab = build_keys(args.workload, n_keys)
    strong = args.scaling == "strong" and world > 1
    if strong and args.mode == "iter_long" and args.workload == "c3":
        raise SystemExit("iter_long restarts depend on everything in front of a position: ONE haystack does not shard (config 3 is an iter workload)")
    blob = None
    if rank == 0:
        A = acx.Automaton(acx.STORE_INTS)
        A.add_words(keys, range(len(keys)))
        A.make_automaton()
        blob = A.flat_image_bytes()
    t = broadcast_blob(blob, src=0)
    got = t.numpy().tobytes()
    _lib.check(_lib.lib().acx_blob_validate((C.c_char * len(got)).from_buffer_copy(got), len(got)))     # header + checksum, on every rank
    batches, _, _, corpus_bytes = make_batches(torch, None, args.workload, keys, vocab, max(1, args.batches), args.reads, args.read_len,
                                               args.batch_mb, rank, world, strong)
    mine = [sum(b[1] for b in batches), sum(b[2] for b in batches), len(got)]
    sha = hl.sha256()
    for b in batches:
        sha.update(np.ascontiguousarray(b[0]).tobytes())
    rows = [mine + [int(sha.hexdigest()[:12], 16)]]
    if world > 1:
        tt = torch.zeros(world, 4, dtype=torch.int64)
        tt[rank] = torch.tensor(rows[0], dtype=torch.int64)
        dist.all_reduce(tt)
        rows = tt.tolist()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(json_fd, (json.dumps({"dry_run": True, "n_gpus": world, "scaling": "strong" if strong else "weak", "mode": args.mode, "workload": args.workload,
                          "batches": len(batches), "bytes_total": sum(r[0] for r in rows), "haystacks_total": sum(r[1] for r in rows),
                          "corpus_bytes_per_batch": int(np.mean(corpus_bytes)), "corpus_bytes_all_batches": int(np.sum(corpus_bytes)),
                          "blob_bytes": rows[0][2], "blob_bytes_equal_on_all_ranks": len({r[2] for r in rows}) == 1,
                          "per_rank": [{"rank": i, "bytes": r[0], "haystacks": r[1], "shard_sha48": "%012x" % r[3]} for i, r in enumerate(rows)]}) + "\n").encode())


def kernel_source_hash(kernel):
    h = hashlib.sha256()
    for name in KERNEL_SOURCES.get(kernel, ()):
        with open(os.path.join(ROOT, "pyahocorasick_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


# ---- CPU baseline: the reference itself (oracle/_ref) or the plain-C port, 1 core and all cores --------------
_CPU = {}


def _cpu_worker(span):
    a, b = span
    scan, hays = _CPU["scan"], _CPU["hays"]
    n = 0
    t0 = time.perf_counter()
    for i in range(a, b):
        for _ in scan(hays[i]):
            n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(keys, hays, mode):
    """hays: list of bytes.  Times Automaton.iter / iter_long of the reference drained per haystack on ONE
    core, then on ALL host cores (multiprocessing fork, automaton inherited copy-on-write, haystacks
    partitioned contiguously).  Falls back to the plain-C port (oracle/ac_oracle.c) when oracle/_ref is absent.
    Baseline only — never on the product path."""
    import multiprocessing as mp
    from oracle import orc
    nbytes = sum(len(h) for h in hays)
    ref = orc.load_reference()
    cores = os.cpu_count() or 1
    if ref is None:
        O = orc.Oracle()
        for i, k in enumerate(keys):
            O.add_word(k, i)
        O.make_automaton()
        data = b"".join(hays)
        off = np.concatenate([[0], np.cumsum([len(h) for h in hays])]).astype(np.int64)
        m = 0 if mode == "iter" else 1
        t0 = time.perf_counter()
        n1 = O.batch_count(data, off, m)
        dt1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        mo, _, _ = O.batch_records(data, off, m, threads=cores)
        dta = time.perf_counter() - t0
        return {"value": nbytes / dt1 / 1e9, "unit": "GB/s", "cores": 1, "kind": "port", "matches_per_s": n1 / dt1, "seconds": round(dt1, 3),
                "all_cores": {"value": nbytes / dta / 1e9, "unit": "GB/s", "cores": cores, "seconds": round(dta, 3), "matches_per_s": int(mo[-1]) / dta},
                "host_cpus": cores, "sample": "%d haystacks, %.1f MB, oracle/ac_oracle.c" % (len(hays), nbytes / 1e6)}
    A = ref.Automaton(ref.STORE_INTS)
    for i, k in enumerate(keys):
        A.add_word(k, i)
    A.make_automaton()
    _CPU["scan"] = A.iter if mode == "iter" else A.iter_long
    _CPU["hays"] = hays
    _cpu_worker((0, min(len(hays), 2000)))                  # warm-up pass, discarded
    n1, dt1 = _cpu_worker((0, len(hays)))
    # all cores: fork AFTER the automaton exists; each worker times its own contiguous share
    workers = max(1, min(cores, len(hays)))
    cuts = np.linspace(0, len(hays), workers + 1).astype(int)
    spans = [(int(cuts[i]), int(cuts[i + 1])) for i in range(workers)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        parts = pool.map(_cpu_worker, spans, chunksize=1)
    dta = time.perf_counter() - t0
    na = sum(p[0] for p in parts)
    return {"value": nbytes / dt1 / 1e9, "unit": "GB/s", "cores": 1, "kind": "reference",
            "matches_per_s": n1 / dt1, "seconds": round(dt1, 3),
            # value: the parallel compute time = the slowest worker's own loop (what a long-lived pool would deliver);
            # wall_incl_fork_s also counts forking and joining the workers
            "all_cores": {"value": nbytes / max(p[1] for p in parts) / 1e9, "unit": "GB/s", "cores": workers,
                          "seconds": round(max(p[1] for p in parts), 3), "wall_incl_fork_s": round(dta, 3),
                          "matches_per_s": na / max(p[1] for p in parts)},
            "host_cpus": cores,
            "sample": "%d haystacks of batch 0 (%.1f MB), Automaton.%s of the reference drained per haystack; "
                      "1 core, then %d forked workers (automaton inherited copy-on-write, contiguous shares)" % (len(hays), nbytes / 1e6, mode, workers)}


WORKLOAD_NAMES = {
    "c2": "config2: %d ACGT keys 8-32 B, %d x %d B reads per batch",
    "c2o": "config2 delivered as an offsets batch: %d ACGT keys 8-32 B, %d reads per batch cut to U[100,%d] B (the general stream kernel)",
    "c2k": "config2 with longer keys: %d ACGT keys 8-64 B, %d x %d B reads per batch (the general stream kernel)",
    "c3": "config3 shape: %d multi-word text keys, %d MiB of text per batch scanned as ONE haystack",
    "c4": "config4 shape: %d Snort-style byte signatures 4-128 B, %d MiB of packets 64-1500 B per batch",
}


def build_keys(workload, n_keys):
    from pyahocorasick_amd import workloads as W
    vocab = None
    if workload in ("c2", "c2o"):
        keys = W.dna_keys(n_keys, seed=0)
    elif workload == "c2k":
        keys = W.dna_keys(n_keys, seed=0, klo=8, khi=64)
    elif workload == "c3":
        vocab = W.text_vocab(1_000_000 if n_keys >= 100_000 else 10 * n_keys, seed=2)
        keys = W.text_keys(vocab, n_keys, seed=3)
    else:
        keys = W.snort_signatures(n_keys, seed=5)
    return keys, vocab


def strong_shard(workload, data, off, rank, world, longest):
    """--scaling strong: rank `rank`'s contiguous share of ONE corpus that every rank generated identically.
    c2: `data` is uint8[n_reads, L], cut by count; c4: packets (data, off), cut so that the shares are balanced by BYTES
    (SURVEY §8e); c3: one text, shares of equal length plus longest - 1 bytes of left halo (exact for iter).
    Returns (data, off) of the share (off: None / rebased to 0 / [0, len])."""
    from pyahocorasick_amd.parallel import halo_shard, shard_range, shard_range_by_bytes
    if workload == "c2":
        lo, hi = shard_range(len(data), rank, world)
        return data[lo:hi], None
    if workload == "c4":
        lo, hi = shard_range_by_bytes(off, rank, world)
        return data[off[lo]:off[hi]], off[lo:hi + 1] - off[lo]
    s0, lo, hi = halo_shard(len(data), rank, world, longest)
    return data[s0:hi], np.array([0, hi - s0], dtype=np.int64)


def make_batches(torch, dev, workload, keys, vocab, n_batches, reads, read_len, batch_mb, rank, world, strong):
    """this rank's batches, resident in HBM: [(device bytes, capacity, n haystacks, device offsets or None, stride, shortest)],
    the haystacks of batch 0 for the CPU baseline, batch 0 as (bytes, offsets) for the host-to-host leg, bytes of the whole
    corpus of one step over all ranks.  dev = None (--dry-run, no GPU): the batches stay numpy arrays on the host."""
    from pyahocorasick_amd import workloads as W
    longest = max(len(k) for k in keys)
    batches, host0, e2e0, corpus_bytes = [], None, None, []
    for b in range(n_batches):
        seed = 1 + b + (0 if strong else 16 * rank)          # strong: every rank generates the SAME corpus and keeps its shard
        if workload in ("c2", "c2k"):
            r = W.dna_reads(keys, reads, read_len, seed=seed)
            corpus_bytes.append(r.size)
            if strong:
                r, _ = strong_shard("c2", r, None, rank, world, longest)
            n, L = r.shape
            flat, off = r.reshape(-1), None
            if b == 0:
                host0 = [r[i].tobytes() for i in range(n)]
                e2e0 = (np.ascontiguousarray(flat), np.arange(n + 1, dtype=np.int64) * L)
        elif workload == "c2o":
            # the reads of config 2 with ragged lengths U[100, read_len], back to back, delivered by offsets
            r = W.dna_reads(keys, reads, read_len, seed=seed)
            lens = np.random.default_rng(1000 + seed).integers(min(100, read_len), read_len + 1, size=len(r), dtype=np.int64)
            keep = np.arange(read_len, dtype=np.int64)[None, :] < lens[:, None]
            flat = np.ascontiguousarray(r[keep])
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            corpus_bytes.append(len(flat))
            n, L = len(lens), 0
            if b == 0:
                host0 = [flat[off[i]:off[i + 1]].tobytes() for i in range(min(n, 200_000))]
                e2e0 = (flat, off)
        elif workload == "c3":
            nbytes = batch_mb << 20
            flat = np.concatenate([W.text_corpus(vocab, min(64 << 20, nbytes - o), seed=4 + 64 * seed + o // (64 << 20))
                                   for o in range(0, nbytes, 64 << 20)])
            corpus_bytes.append(len(flat))
            if strong:            # one corpus, contiguous shards, longest_word-1 bytes of left halo (exact for iter)
                flat, _ = strong_shard("c3", flat, None, rank, world, longest)
            n, L, off = 1, 0, np.array([0, len(flat)], dtype=np.int64)
            if b == 0:
                host0 = [flat[i:i + (1 << 16)].tobytes() for i in range(0, min(len(flat), 32 << 20), 1 << 16)]
                e2e0 = (np.ascontiguousarray(flat), off)
        else:
            flat, off = W.packet_payloads(keys, batch_mb << 20, seed=6 + seed)
            corpus_bytes.append(len(flat))
            if strong:            # packets are independent haystacks of unequal length: contiguous shards balanced by BYTES
                flat, off = strong_shard("c4", flat, off, rank, world, longest)
            n, L = len(off) - 1, 0
            if b == 0:
                m = min(len(off) - 1, int(np.searchsorted(off, 32 << 20)))
                host0 = [flat[off[i]:off[i + 1]].tobytes() for i in range(m)]
                e2e0 = (np.ascontiguousarray(flat), off)
        # offsets batches: the shortest haystack, which the caller of acx_scan_batch vouches for (min_hay_len)
        shortest = int(np.diff(off).min()) if off is not None and len(off) > 1 else 0
        if dev is None:
            batches.append((np.ascontiguousarray(flat), len(flat), n, off, L, shortest))
            continue
        d_hay = torch.empty(len(flat) + 64, dtype=torch.uint8, device=dev)
        d_hay[: len(flat)].copy_(torch.from_numpy(np.ascontiguousarray(flat)))
        d_off = torch.from_numpy(np.ascontiguousarray(off)).to(dev) if off is not None else None
        batches.append((d_hay, len(flat), n, d_off, L, shortest))
    if dev is not None:
        torch.cuda.synchronize()
    return batches, host0, e2e0, corpus_bytes


SCAN_STREAMS = 3
LONG_DEPTH = 0                          # --long-depth: results in flight and scan streams of an iter_long measurement (0: as the others)
ACX_LONG_MODE = None                    # acx.ACX_SCAN_LONG once the package is imported (main)
_SCAN_STREAMS = []                      # the process's extra scan streams (measure())
MIN_TIMED_MS = 250.0


def measure(torch, dist, dev, image, batches, mode, steps, warmup, P, event_every, variant, repeats=0):
    """K x R timed passes rotating over the batches (barrier + synchronize on both sides, max over ranks by the caller; R =
    `repeats`, or — 0 — chosen from a short calibration so that the region lasts MIN_TIMED_MS, the same R on every rank).
    Every collected pass's record count is compared with the count of the same batch in the untimed pre-pass."""
    from pyahocorasick_amd.device import Scanner
    B = len(batches)
    scs = [Scanner(image) for _ in range(P)]
    stream = torch.cuda.current_stream().cuda_stream
    # (iter_long's walk kernel is many small blocks: two of them share every CU for their whole length and each other's L2 —
    #  168 instead of 171 GB/s — so only the position-parallel scans alternate)
    from pyahocorasick_amd import ACX_SCAN_ALL
    # (iter_long in its position-parallel form — a scan over the dictionary of acx_long.cpp + one sweep — honours ACX_SCAN_ASYNC since
    #  round 5: scan kernel on the caller's stream, gather and sweep behind it on a side stream; the serial walk, many small blocks
    #  that share every CU, lost 2 % on two streams)
    n_streams = P if (LONG_DEPTH and mode == ACX_LONG_MODE) else min(SCAN_STREAMS, P)
    # The scan streams are made ONCE per process and shared by every configuration it measures: which hardware queue a stream gets depends on
    # the order in which the process made its streams, and a configuration whose gather overlaps its scans for most of a step is sensitive to
    # WHICH queues its streams share — `c2_long_keys` ran at 607 / 646 / 703 GB/s behind the headline alone / in a process of its own / behind
    # `c2_offsets` while every measure() made two streams of its own (round 6).  Shared streams: the line's entries are what a process of their
    # own gives (tools/roofline_check.py compares them with traces of exactly that).
    # (iter_long's further streams (--long-depth) are made when it is measured: made at the process's first measurement they cost config 5 more —
    #  237.9 / 249.3 / 243.8 GB/s in three lines against 251.9 / 254.9 / 248.6, profiles/r6_line_runs.txt)
    while len(_SCAN_STREAMS) < max(0, n_streams - 1):
        _SCAN_STREAMS.append(torch.cuda.Stream())
    extra = _SCAN_STREAMS[:max(0, n_streams - 1)]
    streams = [stream] + [x.cuda_stream for x in extra]     # slot k % P scans on stream (k % P) % len(streams)

    def step(k, timing=False):
        d_hay, cap, n, d_off, L, shortest = batches[k % B]
        return scs[k % P].scan(d_hay.data_ptr(), cap, n, dev_off=d_off.data_ptr() if d_off is not None else None,
                               stride=L, mode=mode, timing=timing, variant=variant, stream=streams[(k % P) % len(streams)], asynchronous=P > 1,
                               min_hay_len=shortest)

    for k in range(max(warmup, P, B)):                    # every batch scanned at least once before timing
        step(k)
    for x in scs:
        x.wait()
    # per-kernel times (HIP events around every kernel of a step) from separate passes over every batch
    pre = {"walk": [], "scan": [], "expand": [], "total": []}
    matches_per_batch = []
    for k in range(B):
        step(k, timing=True)
        scs[k % P].wait()
        t = scs[k % P].timing_ms()
        for key in pre:
            pre[key].append(t[key])
        matches_per_batch.append(scs[k % P].num_matches())
    pre = {k: float(np.mean(v)) for k, v in pre.items()}
    # the synchronous step (pipeline depth 1: scan and gather one after the other, the host waits for each)
    sync_ms = None
    if P > 1:
        ssc = Scanner(image)
        d_hay, cap, n, d_off, L, shortest = batches[0]
        ts = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ssc.scan(d_hay.data_ptr(), cap, n, dev_off=d_off.data_ptr() if d_off is not None else None, stride=L, mode=mode,
                     variant=variant, stream=stream, asynchronous=False, min_hay_len=shortest)
            ts.append(time.perf_counter() - t0)
        sync_ms = float(np.median(ts[1:])) * 1e3
        # (ssc stays alive until the caller drops the returned dict: freeing its buffers — 8 B of event scratch per haystack
        #  byte in iter_long mode, gigabytes — right in front of the timed region hands the driver unmapping work to do)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- R: how many passes make one driver step (calibration: a few pipelined passes, untimed region) ----------------
    R = int(repeats)
    if R <= 0:
        ncal = max(2 * P, 2 * B, 8)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for k in range(ncal):
            step(k)
        for x in scs:
            x.wait()
        torch.cuda.synchronize()
        t_pass = (time.perf_counter() - tc) / ncal
        R = int(min(100000, max(1, -(-MIN_TIMED_MS * 1e-3 // max(1e-9, steps * t_pass)))))
        if dist is not None:                                 # the same R on every rank: the largest any of them asks for
            rr = torch.tensor([R], dtype=torch.int64, device=dev)
            dist.all_reduce(rr, op=dist.ReduceOp.MAX)
            R = int(rr.item())
    passes = steps * R

    # ---- the timed region: K x R passes, rotating over the batches; the dominant kernel of every N-th pass is bracketed
    #      by HIP events on the stream it runs on (timing = 2: an event between two kernels costs ~5 us of idle GPU)
    walk_ms, timed, owner = [], {}, {}

    def collect(x):
        x.wait()
        k = owner.pop(id(x))
        got = x.num_matches()
        if got != matches_per_batch[k % B]:                # a dropped, racing or repeated step would show here
            raise SystemExit("bench: step %d reports %d matches, the pre-pass of its batch %d" % (k, got, matches_per_batch[k % B]))
        if timed.pop(id(x), False):
            walk_ms.append(x.timing_ms()["walk"])

    marks = []                                             # host clock after every collected step: a stall shows as ONE long interval
    barrier()
    t0 = time.perf_counter()
    for k in range(passes):
        if k >= P:
            collect(scs[k % P])
            marks.append(time.perf_counter())
        ev = event_every > 0 and k % event_every == 0
        timed[id(scs[k % P])] = ev
        owner[id(scs[k % P])] = k
        step(k, timing=2 if ev else False)
    for k in range(max(0, passes - P), passes):
        collect(scs[k % P])                                # every pass complete: totals read and checked, records in HBM
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    gaps = np.diff(np.array([t0] + marks)) * 1e3
    step_ms = {"median": round(float(np.median(gaps)), 4), "max": round(float(gaps.max()), 4), "argmax": int(gaps.argmax())}
    # ---- the union leg, BEHIND the timed region (it is in nothing that `value` is made of): with launches that overlap — several scan
    #      streams — no per-launch span is a duration of the kernel (each includes the time it shares the chip), and timed region / launches
    #      includes the time only gathers run.  What the algorithmic bytes of a launch can be divided by is the time the kernel occupies the
    #      chip per launch: the UNION of its launch spans / launches.  Measured live: an event pair on the scan's stream around every launch
    #      of 12 x P pipelined passes, the first and last P left out; tools/roofline_check.py recomputes the same from a rocprofv3 trace.
    union_ms = union_share = None
    if len(streams) > 1:
        tstreams = [torch.cuda.current_stream()] + extra
        U = 12 * P
        evs = []
        torch.cuda.synchronize()
        for k in range(U):
            if k >= P:
                scs[k % P].wait()
            so = tstreams[(k % P) % len(streams)]
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record(so)
            step(k)
            eb.record(so)
            evs.append((ea, eb))
        for x in scs:
            x.wait()
        torch.cuda.synchronize()
        iv = sorted((evs[0][0].elapsed_time(ea), evs[0][0].elapsed_time(eb)) for ea, eb in evs[P:U - P])
        tot, cs, ce = 0.0, None, None
        for a_, b_ in iv:
            if ce is None or a_ > ce:
                if ce is not None:
                    tot += ce - cs
                cs, ce = a_, b_
            else:
                ce = max(ce, b_)
        tot += (ce - cs) if ce is not None else 0.0
        union_ms = tot / max(1, len(iv))
        # ... and as a share of the leg's own region (first counted launch's start to the last one's end): the leg is not the timed region —
        # an event pair per launch, no totals read — and where its step differs from the timed region's, the share is what carries over
        leg_region = (max(b_ for _, b_ in iv) - min(a_ for a_, _ in iv)) if iv else 0.0
        union_share = min(1.0, tot / leg_region) if leg_region > 0 else None
    return {"dt": dt, "dt_rank": dt_rank, "walk_ms": walk_ms, "pre": pre, "step_ms": step_ms, "matches_per_batch": matches_per_batch, "sync_ms": sync_ms,
            "union_ms": union_ms, "union_share": union_share,
            "repeats": R, "passes": passes,
            "bytes_rank": sum(batches[k % B][1] for k in range(passes)), "matches_rank": sum(matches_per_batch[k % B] for k in range(passes)),
            "scanner": scs[0], "stream": stream, "keepalive": (ssc if P > 1 else None, extra), "scan_streams": len(streams)}


def roofline_entry(image, batches, m, mode_name, workload, variant, event_every):
    """the dominant kernel against the HBM roofline: algorithmic bytes per launch / its launch duration, PMC traffic from
    profiles/traffic.json when it was measured on these kernel sources.

    The launch duration: with ONE scan stream, HIP events around the kernel inside the timed region (rocprofv3's average
    agrees: profiles/r4_clock_check.txt).  With several (the default: three, three results in flight) launch spans overlap —
    the blocks of launch k + 1 start on the CUs launch k's finished blocks leave — and no per-launch clock describes the
    kernel (events 0.50 ms, rocprofv3 500 us for launches of which one completes every 0.26 ms): there the duration is the
    UNION of the kernel's launch spans / launches — the time the kernel occupies the chip per launch —, measured by
    measure()'s union leg with an event pair around every launch; tools/roofline_check.py recomputes it from a
    rocprofv3 --kernel-trace of the same command (profiles/r5_*_spans.json).  Timed region / launches (`kernel_region_bound_ms`)
    bounds it from above: the difference is the time in which only gathers run.  `step` is the
    regime-independent figure: all algorithmic bytes of a batch scan / ms_per_step."""
    B = len(batches)
    passes = m["passes"]
    d_hay, cap0, n0, d_off0, L0, shortest0 = batches[0]
    H = m["bytes_rank"] / passes                             # haystack bytes per rank per pass (mean over the rotation)
    M = m["matches_rank"] / passes
    Nh = float(np.mean([batches[k % B][2] for k in range(passes)]))
    A_bytes = H + 8 * M + 12 * Nh                            # H + 8*M + 12*N
    pre = m["pre"]
    ms_pass = m["dt_rank"] / passes * 1e3
    overlap = m["scan_streams"] > 1
    walk_ev = float(np.mean(m["walk_ms"])) if m["walk_ms"] else pre["walk"]
    # (the union leg runs behind the timed region with an event pair around every launch, which costs its stream 5–20 us of idle time: the
    #  leg's step is longer than the timed region's by about that, and the SHARE of its region that the spans cover — reported as
    #  kernel_union_share — is 4–8 % below what a rocprofv3 trace of the timed region shows on steps of 0.2–0.3 ms (0.958 against 0.995 on
    #  config 2).  The union per launch itself carries over — the idle time is not inside the spans —, and where it comes out ABOVE timed
    #  region / launches, the bound that holds in the timed region itself, the bound is the better figure)
    share = m.get("union_share")
    walk = min(m.get("union_ms") or ms_pass, ms_pass) if overlap else walk_ev
    used_ppm = mode_name == "iter" and image.ppm_kernel(stride=L0, has_offsets=d_off0 is not None, variant=variant, min_hay_len=shortest0,
                                                          dev_hay=d_hay.data_ptr(), n_hay=n0)
    if mode_name != "iter":
        # iter_long: the position-parallel scan over the dictionary of acx_long.cpp where it applies (the scan kernel is the
        # dominant one; the sweep over its records is `expand` in kernel_ms), else the serial walk.  Algorithmic bytes of the
        # task either way: the haystack in, one offset per haystack out (the records are in the pipeline figure below).
        from pyahocorasick_amd import ACX_SCAN_LONG
        long_ppm = image.ppm_kernel(stride=L0, has_offsets=d_off0 is not None, variant=variant, min_hay_len=shortest0,
                                    dev_hay=d_hay.data_ptr(), n_hay=n0, mode=ACX_SCAN_LONG)
        walk_kernel = {"stream4": "k_ppm_stream4", "stream": "k_ppm_stream", "scan": "k_ppm_scan"}.get(long_ppm, "k_walk_long_sel")
        walk_bytes = H + 12 * Nh
    elif used_ppm in ("stream", "stream4"):
        # the scan kernel reads the haystack, writes every record (to the pool) and one offset per haystack
        walk_kernel, walk_bytes = ("k_ppm_stream4" if used_ppm == "stream4" else "k_ppm_stream"), H + 8 * M + (4 if d_off0 is None else 12) * Nh
    elif used_ppm == "scan":
        walk_kernel, walk_bytes = "k_ppm_scan", H + 8 * M + 8 * (H / 256)
    else:
        walk_kernel, walk_bytes = ("k_walk_itop" if image.itop_depth > 0 and not (variant >> 16) & 1 else "k_walk_all"), H + 12 * Nh
    traffic = traffic_raw = traffic_x2 = None
    traffic_note = "profiles/traffic.json absent"
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    src_hash = kernel_source_hash(walk_kernel)
    if os.path.exists(tpath):
        try:
            ent = json.load(open(tpath)).get("%s_%s" % (workload, mode_name))
            if not ent:
                traffic_note = "no PMC entry for this workload"
            elif ent.get("kernel") != walk_kernel:
                traffic_note = "PMC entry is for kernel %s" % ent.get("kernel")
            elif ent.get("kernel_source_sha") != src_hash:
                traffic_note = "PMC entry is for other sources of %s (%s, now %s): refused" % (walk_kernel, ent.get("kernel_source_sha"), src_hash)
            else:
                traffic_raw = ent["fetch_bytes_raw"] + ent["write_bytes"]
                traffic_x2 = ent["fetch_bytes_x2_gfx950"] + ent["write_bytes"]
                # one figure: FETCH_SIZE scaled by the factor tools/fetch_calib.hip measured for THIS access mix on this chip
                # (profiles/r5_fetch_calibration.json; entries made before it existed carry none: the x2 bound stands in)
                traffic = ent.get("traffic_calibrated", traffic_x2)
                traffic_note = ent.get("note", "")
        except Exception as ex:                             # noqa: BLE001
            traffic_note = "unreadable: %s" % ex
    gpu_ms = walk_ev + pre["scan"] + pre["expand"]

    def fig(nbytes, ms):
        return {"algorithmic_bytes": nbytes, "ms": round(ms, 4), "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    return {
        "bound": "hbm", "kernel": walk_kernel,
        "achieved": walk_bytes / (walk * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": walk_bytes / (walk * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "algorithmic_bytes": walk_bytes, "kernel_avg_ms": round(walk, 4),
        "kernel_ms_source": ("union of the kernel's overlapping launch spans / launches: events around every launch of a pipelined leg behind the timed region "
                             "(tools/roofline_check.py: the same from a rocprofv3 trace); timed region / launches bounds it from above"
                             if overlap else "HIP events around the kernel on its stream, inside the timed region"),
        "kernel_region_bound_ms": round(ms_pass, 4), "kernel_union_share": None if not (overlap and share) else round(share, 4),
        "kernel_union_leg_ms": None if not (overlap and m.get("union_ms")) else round(m["union_ms"], 4),
        # HIP events around the kernel, on its stream, inside the timed region: in every N-th pass (an event
        # pair costs the stream ~19 us of idle time in the pass it is in).  With overlapping launches they are reported, not used.
        "kernel_events": {"every_nth_step": event_every, "samples": len(m["walk_ms"]), "avg_ms": round(walk_ev, 4)},
        "launches_overlap": overlap,
        # the same kernel in the pre-pass: one launch at a time, waited for (the chip to itself) — the per-launch figure that
        # rocprofv3 --stats of a --scan-streams 1 run agrees with
        "kernel_alone": dict(fig(walk_bytes, pre["walk"]), source="pre-pass: one launch at a time, HIP events"),
        "kernel_alone_ms": round(pre["walk"], 4),
        # regime-independent: ALL algorithmic bytes of a batch scan (A = H + 8 M + 12 N) over the whole step, as the driver's clock sees it
        "step": fig(A_bytes, ms_pass),
        # HBM bytes of the dominant kernel per launch from the PMC passes (FETCH_SIZE + WRITE_SIZE).  `traffic`: FETCH_SIZE scaled
        # by the factor measured for this access mix (tools/fetch_calib.hip), or — entries older than the calibration — the
        # bound /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2); `traffic_raw`: the counters as read
        "traffic": traffic, "traffic_raw": traffic_raw, "traffic_x2_bound": traffic_x2,
        "traffic_note": traffic_note, "kernel_source_sha": src_hash,
        # the whole batch scan (scan kernel + prefix sum + gather) by the kernels' own durations, A = H + 8*M + 12*N
        "pipeline": {"algorithmic_bytes": A_bytes, "gpu_ms": round(gpu_ms, 4),
                     "achieved": A_bytes / (gpu_ms * 1e-3) / 1e9, "frac": A_bytes / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "kernel_ms": {"walk": round(walk_ev, 4), "scan": round(pre["scan"], 4), "expand": round(pre["expand"], 4)}},
    }


def _progress(what):
    """ACX_BENCH_PROGRESS=1: which leg the process is in, to stderr (a leg that dies says nothing else)"""
    if os.environ.get("ACX_BENCH_PROGRESS"):
        print("[bench] " + what, file=sys.stderr, flush=True)


def other_config(torch, dev, acx, name, workload, mode_name, keys, vocab, image, batches, host0, args, n_keys, batch_mb, cpu_sample):
    """one of the other named single-GPU configurations, measured like the headline (fewer batches and steps)"""
    from pyahocorasick_amd.device import Image
    _progress("config " + name)
    t0 = time.perf_counter()
    t_build = None
    if image is None:
        A = acx.Automaton(acx.STORE_INTS)
        A.add_words(keys, range(len(keys)))
        A.make_automaton()
        image = Image.from_automaton(A)
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t0
        del A
    t0 = time.perf_counter()
    if batches is None:
        # (config 2's shapes rotate over FOUR batches as the headline does: two offsets batches are 250 MB — what the 256 MiB Infinity Cache
        #  holds — and `c2_offsets` ran 5 % faster in the line than in a process of its own, which stages four; 512 MiB batches: two)
        batches, host0, _, _ = make_batches(torch, dev, workload, keys, vocab, 4 if workload.startswith("c2") else 2, args.reads, args.read_len, batch_mb, 0, 1, False)
    t_stage = time.perf_counter() - t0
    mode = acx.ACX_SCAN_ALL if mode_name == "iter" else acx.ACX_SCAN_LONG
    steps = max(8, min(args.steps, 20))
    depth = args.long_depth if (mode_name == "iter_long" and args.long_depth > 0) else max(1, args.pipeline)
    m = measure(torch, None, dev, image, batches, mode, steps, 2, depth, args.event_every, 0, args.inner_repeats)
    out = {
        "value": m["bytes_rank"] / m["dt"] / 1e9, "unit": "GB/s", "ms_per_step": m["dt"] / m["passes"] * 1e3, "steps": steps,
        "inner_repeats": m["repeats"], "timed_region_ms": round(m["dt"] * 1e3, 3),
        "ms_per_step_synchronous": m["sync_ms"], "step_ms_host_intervals": m["step_ms"], "scan_streams": m["scan_streams"], "results_in_flight": depth,
        "matches_per_step": m["matches_rank"] / m["passes"],
        "workload": (WORKLOAD_NAMES[workload] % ((n_keys, args.reads, args.read_len) if workload.startswith("c2") else (n_keys, batch_mb)))
                    + ", Automaton.%s; %d distinct batches rotated (%.0f MB resident)" % (mode_name, len(batches), sum(b[1] for b in batches) / 1e6),
        "states": int(image.num_states), "image_mb": round(image.nbytes / 1e6, 1),
        "roofline": roofline_entry(image, batches, m, mode_name, workload, 0, args.event_every),
        "setup": {"build_flatten_upload_s": None if t_build is None else round(t_build, 3), "stage_batches_s": round(t_stage, 3)},
    }
    if cpu_sample and host0:
        out["cpu_baseline"] = cpu_baseline(keys, host0[:cpu_sample], mode_name)
    del m
    return out


def _rnd(x, n=4):
    return round(x, n) if isinstance(x, float) else x


def compact_roofline(r):
    """what the contract asks of `roofline` (bound, achieved, peak, unit, frac, traffic) + what tools/roofline_check.py recomputes it from"""
    if not r:
        return r
    return {"bound": r["bound"], "kernel": r["kernel"], "achieved": _rnd(r["achieved"], 1), "peak": r["peak"], "unit": r["unit"], "frac": _rnd(r["frac"], 5),
            "traffic": r.get("traffic"), "algorithmic_bytes": int(r["algorithmic_bytes"]), "kernel_avg_ms": r["kernel_avg_ms"],
            "kernel_alone_ms": r.get("kernel_alone_ms"), "launches_overlap": r.get("launches_overlap"),
            "step": {"algorithmic_bytes": int(r["step"]["algorithmic_bytes"]), "ms": r["step"]["ms"], "frac": _rnd(r["step"]["frac"], 5)}}


def compact_cpu(c):
    if not c:
        return c
    out = {"value": _rnd(c["value"], 5), "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "host_cpus": c.get("host_cpus"),
           "sample": c["sample"].split(",")[0] + ", %.1f s on 1 core" % c.get("seconds", 0.0)}
    if c.get("all_cores"):
        out["all_cores"] = {"value": _rnd(c["all_cores"]["value"], 4), "cores": c["all_cores"]["cores"]}
    return out


def compact_line(out):
    """The driver's line: the full object minus prose and per-step lists — short enough for the 8 KB tail the driver keeps of the
    output (two of six configurations were cut off).  `--verbose` prints, `--full-json` writes, the whole object."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "inner_repeats", "passes", "timed_region_ms", "ms_per_step", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data", "matches_per_s", "matches_per_step", "bytes_total", "ms_per_step_synchronous", "per_rank_GBps",
            "end_to_end_GBps", "setup")
    o = {k: _rnd(out[k], 5) for k in keep if k in out}
    o["per_rank_GBps"] = {k: _rnd(v, 2) for k, v in out["per_rank_GBps"].items()}
    if out["n_gpus"] > 1:
        o["ranks"] = {"reported_by_process_group": out["ranks"]["reported_by_process_group"], "took_part": out["ranks"]["took_part"],
                      "GBps": [r["GBps"] for r in out["ranks"]["per_rank"]], "kernel_alone_ms": [r["kernel_alone_ms"] for r in out["ranks"]["per_rank"]]}
    cfg = dict(out["config"])
    cfg["workload"] = cfg["workload"].split(";")[0]
    frac = lambda e: _rnd(e["roofline"]["frac"], 4) if e.get("roofline") else None
    allc = {"headline": [_rnd(out["value"], 1), frac(out)]}
    for name, e in (out.get("configs") or {}).items():
        allc[name] = [_rnd(e["value"], 1), frac(e)] if "value" in e else [None, None]
    cfg["all"] = allc                                          # {configuration: [GB/s, roofline.frac]} — `parsed` of the driver keeps `config`
    o["config"] = cfg
    o["roofline"] = compact_roofline(out.get("roofline"))
    if "cpu_baseline" in out:
        o["cpu_baseline"] = compact_cpu(out["cpu_baseline"])
    if "one_scan_stream" in out:
        o["one_scan_stream"] = {"value": _rnd(out["one_scan_stream"]["value"], 1), "ms_per_step": _rnd(out["one_scan_stream"]["ms_per_step"], 5),
                                "roofline": compact_roofline(out["one_scan_stream"]["roofline"])}
    if "configs" in out:
        cc = {}
        for name, e in out["configs"].items():
            if "value" not in e:
                cc[name] = {"error": str(e.get("error"))[:160]}
                continue
            cc[name] = {"value": _rnd(e["value"], 2), "unit": e["unit"], "ms_per_step": _rnd(e["ms_per_step"], 5), "steps": e["steps"], "inner_repeats": e["inner_repeats"],
                        "scan_streams": e["scan_streams"], "results_in_flight": e.get("results_in_flight"), "workload": e["workload"].split(":")[0].split(";")[0], "roofline": compact_roofline(e["roofline"]),
                        "setup_s": (e.get("setup") or {}).get("build_flatten_upload_s")}
            if "cpu_baseline" in e:
                cc[name]["cpu_baseline"] = compact_cpu(e["cpu_baseline"])
        o["configs"] = cc
    return o


def main():
    args = parse()
    self_launch(args)                                      # --gpus N > 1 outside torch.distributed.run: N ranks are started here
    if args.launch_check:
        return launch_check(args)
    if args.dry_run:
        return dry_run(args)
    # stdout must carry exactly ONE JSON line: RCCL prints a version banner on fd 1 when its
    # communicator comes up, so everything until the final print goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d (local %d) has no GPU: %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # under torch.distributed.run (RANK/MASTER_ADDR set) the process group is created even for one
    # rank, so the N = 1 line and the N > 1 lines go through the same code (RCCL broadcast, barriers)
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
        if dist.get_world_size() != args.gpus:              # RCCL sees fewer (or more) ranks than the line will claim
            raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, dist.get_world_size()))

    import pyahocorasick_amd as acx
    from pyahocorasick_amd import _lib
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from pyahocorasick_amd.parallel import broadcast_image
    _lib.check(_lib.lib().acx_device_set(local_rank))
    dummy_streams = []
    if args.dummy_streams > 0:
        os.environ["ACX_BENCH_DUMMY_STREAMS"] = str(args.dummy_streams)
        for _ in range(args.dummy_streams):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev).add_(1)
            dummy_streams.append(st)                            # (kept alive: their queues stay taken)
        torch.cuda.synchronize()

    # ---- dictionary: built on rank 0 (CPU), replicated by one RCCL broadcast ---------------------
    n_keys = args.keys or (1_000_000 if args.workload == "c4" else 100_000)
    keys, vocab = build_keys(args.workload, n_keys)
    t0 = time.perf_counter()
    blob = None
    if rank == 0:
        A = acx.Automaton(acx.STORE_INTS)
        A.add_words(keys, range(len(keys)))                 # (one call: 1 M signatures cost seconds of interpreter time otherwise)
        A.make_automaton()
        blob = A.flat_image_bytes()
        del A
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    # iter_long workloads: the dictionary of its position-parallel form is built once on rank 0 and travels behind the blob in the SAME
    # broadcast (parallel.broadcast_image(long_pack=True) / acx_image_set_long): no rank builds it from a device-to-host copy of its image
    runs_long = args.mode == "iter_long" or (world == 1 and args.workload == "c2" and args.variant == 0 and not args.keys
                                             and ((args.configs or "all") == "all" or "c5_iter_long" in (args.configs or "").split(",")))
    image, image_tensor = broadcast_image(blob, src=0, device=dev, long_pack=runs_long)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    del blob

    strong = args.scaling == "strong" and world > 1
    B = max(1, args.batches)
    t0 = time.perf_counter()
    batches, host0, e2e0, corpus_bytes = make_batches(torch, dev, args.workload, keys, vocab, B, args.reads, args.read_len, args.batch_mb,
                                                      rank, world, strong)
    t_stage = time.perf_counter() - t0
    mode = acx.ACX_SCAN_ALL if args.mode == "iter" else acx.ACX_SCAN_LONG
    global SCAN_STREAMS, MIN_TIMED_MS, LONG_DEPTH, ACX_LONG_MODE
    # (iter_long: --long-depth results in flight on as many scan streams, unless --pipeline / --scan-streams say otherwise)
    P = args.long_depth if (args.mode == "iter_long" and args.long_depth > 0 and "--pipeline" not in sys.argv) else max(1, args.pipeline)
    LONG_DEPTH = args.long_depth if "--scan-streams" not in sys.argv else 0
    ACX_LONG_MODE = acx.ACX_SCAN_LONG
    SCAN_STREAMS = max(1, args.scan_streams)
    MIN_TIMED_MS = max(0.0, args.min_timed_ms)
    _progress("headline")
    m = measure(torch, dist, dev, image, batches, mode, args.steps, args.warmup, P, args.event_every, args.variant, args.inner_repeats)
    _progress("headline measured")
    dt, bytes_rank, matches_rank = m["dt"], m["bytes_rank"], m["matches_rank"]
    passes = m["passes"]
    # what every rank saw for itself: throughput and step by its own clock (before the closing barrier), the dominant kernel by
    # HIP events inside the timed region and alone in the pre-pass
    mine = [bytes_rank / m["dt_rank"] / 1e9, m["dt_rank"] / passes * 1e3,
            float(np.mean(m["walk_ms"])) if m["walk_ms"] else float("nan"), m["pre"]["walk"]]
    per_rank = [mine]
    ranks_seen = 1
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        agg = torch.tensor([bytes_rank, matches_rank, 1], dtype=torch.int64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        bytes_all, matches_all, ranks_seen = int(agg[0].item()), int(agg[1].item()), int(agg[2].item())
        pr = torch.zeros(world, len(mine), dtype=torch.float64, device=dev)
        pr[rank] = torch.tensor(mine, dtype=torch.float64, device=dev)
        dist.all_reduce(pr, op=dist.ReduceOp.SUM)
        per_rank = pr.tolist()
        if ranks_seen != dist.get_world_size() or ranks_seen != args.gpus:
            raise SystemExit("--gpus %d, the process group has %d ranks, %d took part in the reduction" % (args.gpus, dist.get_world_size(), ranks_seen))
    else:
        bytes_all, matches_all = bytes_rank, matches_rank

    if args.verify and rank == 0 and args.mode == "iter":
        from oracle import orc
        O = orc.Oracle()
        for i, k in enumerate(keys):
            O.add_word(k, i)
        O.make_automaton()
        d_hay, cap, n, d_off, L, shortest = batches[0]
        sc0 = m["scanner"]
        sc0.scan(d_hay.data_ptr(), cap, n, dev_off=d_off.data_ptr() if d_off is not None else None, stride=L, mode=mode, stream=m["stream"], min_hay_len=shortest)
        off_g, e, v, _ = sc0.fetch()
        data = d_hay[:cap].cpu().numpy().tobytes()
        offs = np.arange(n + 1, dtype=np.int64) * L if d_off is None else d_off.cpu().numpy()
        mo, oe, ov = O.batch_records(data, offs, 0)
        assert np.array_equal(off_g, mo) and np.array_equal(e, oe) and np.array_equal(v, ov), "GPU result differs from the oracle"

    if rank == 0:
        ms_step = dt / passes * 1e3
        wname = WORKLOAD_NAMES[args.workload] % ((n_keys, args.reads, args.read_len) if args.workload.startswith("c2") else (n_keys, args.batch_mb))
        # bytes of the corpus one step covers over all ranks: strong scaling cuts ONE corpus (bytes_total / step stays what one
        # GPU scans alone, plus the halos of a text shard), weak scaling gives every rank its own
        out = {
            "metric": "GB/s haystack scanned, 100k-pattern automaton" if args.workload != "c4" else "GB/s haystack scanned, %d-signature automaton" % n_keys,
            "value": bytes_all / dt / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # a driver step = R passes of the hot path, each over ONE batch (so that the timed region lasts >= --min-timed-ms
            # whatever --steps says); ms_per_step = timed region / (steps x R) = one batch scan; value = all bytes / timed region
            "inner_repeats": m["repeats"], "passes": passes, "timed_region_ms": round(dt * 1e3, 3), "ms_per_driver_step": dt / args.steps * 1e3,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "matches_per_s": matches_all / dt,
            "matches_per_step": matches_all / passes / world,
            "bytes_total": bytes_all, "bytes_per_step_all_ranks": bytes_all / passes,
            "corpus_bytes_per_batch": int(np.mean(corpus_bytes)),
            "ms_per_step_synchronous": m["sync_ms"], "step_ms_host_intervals": m["step_ms"],
            "per_rank_GBps": {"min": min(r[0] for r in per_rank), "max": max(r[0] for r in per_rank)},
            "ranks": {"reported_by_process_group": (dist.get_world_size() if dist is not None else 1), "took_part": ranks_seen,
                      "backend": (dist.get_backend() if dist is not None else None),
                      "per_rank": [{"rank": i, "GBps": round(r[0], 2), "ms_per_step": round(r[1], 5),
                                    "kernel_ms_events": None if r[2] != r[2] else round(r[2], 5), "kernel_alone_ms": round(r[3], 5)}
                                   for i, r in enumerate(per_rank)]},
            "config": {"workload": wname + ", Automaton.%s; %d distinct batches rotated (%.0f MB resident per GPU)"
                                   % (args.mode, B, sum(b[1] for b in batches) / 1e6),
                       "states": int(image.num_states), "classes": int(image.num_classes),
                       "image_mb": round(image.nbytes / 1e6, 1), "variant": args.variant, "pipeline_depth": P, "scan_streams": m["scan_streams"],
                       "parallelism": "replicated automaton (1 RCCL broadcast), haystacks sharded x%d (%s)" % (world, "strong" if strong else "weak"),
                       # what the overlap of scans and their follow-up work rests on (include/acx.h, ACX_SCAN_ASYNC): hardware queues of the
                       # process, results in flight, the library's side streams per device
                       "queues": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "results_in_flight": P, "scan_streams": m["scan_streams"],
                                  "side_streams": int(_lib.lib().acx_async_streams()), "streams_made_first": int(os.environ.get("ACX_BENCH_DUMMY_STREAMS", "0"))}},
            "roofline": roofline_entry(image, batches, m, args.mode, args.workload, args.variant, args.event_every),
            "setup": {"build_flatten_s": round(t_build, 3), "broadcast_upload_s": round(t_bcast, 3), "stage_batches_s": round(t_stage, 3)},
        }
        if world == 1 and e2e0 is not None and args.mode == "iter" and not args.no_e2e:
            # PCIe-inclusive (SURVEY §8d "also report end-to-end"): host buffers in, offsets and records back in host
            # memory — acx_scan_host_ctx + acx_result_fetch_host, what Automaton.iter / find_all / iter_batch call.
            # Never `value`.  (On this box H2D and D2H do not overlap: profiles/r3_pcie_probe.txt.)
            import ctypes as C
            _progress("host-to-host leg")
            res_e2e = C.c_void_p()
            hflat, hoff = e2e0
            best = None
            for _ in range(4):
                t0 = time.perf_counter()
                _lib.check(_lib.lib().acx_scan_host_ctx(image.handle, hflat.ctypes.data, hoff.ctypes.data, len(hoff) - 1, None, None, None, 0, C.byref(res_e2e)))
                p1, p2, p3 = C.c_void_p(), C.c_void_p(), C.c_void_p()
                _lib.check(_lib.lib().acx_result_fetch_host(res_e2e, C.byref(p1), C.byref(p2), C.byref(p3)))
                dt_e = time.perf_counter() - t0
                best = dt_e if best is None or dt_e < best else best
            out["end_to_end_GBps"] = hflat.size / best / 1e9
            out["end_to_end_ms"] = best * 1e3
            _lib.lib().acx_result_free(res_e2e)
        if world == 1 and args.mode == "iter" and args.scan_streams > 1 and args.one_stream_leg:
            # The same passes with every scan on ONE stream, in a timed region of its own (same barriers): the regime in which HIP
            # events, the pre-pass and rocprofv3 agree on the kernel's launch duration (tools/r4_clock_check.sh,
            # profiles/r4_clock_check.txt).  Never `value`.
            SCAN_STREAMS = 1
            m1 = measure(torch, None, dev, image, batches, mode, args.steps, args.warmup, P, args.event_every, args.variant, m["repeats"])
            SCAN_STREAMS = max(1, args.scan_streams)
            out["one_scan_stream"] = {"value": m1["bytes_rank"] / m1["dt"] / 1e9, "unit": "GB/s", "ms_per_step": m1["dt"] / m1["passes"] * 1e3,
                                      "steps": args.steps, "inner_repeats": m1["repeats"], "step_ms_host_intervals": m1["step_ms"],
                                      "scan_streams": m1["scan_streams"],
                                      "roofline": roofline_entry(image, batches, m1, args.mode, args.workload, args.variant, args.event_every)}
            del m1
        if world == 1 and args.cpu_sample_reads != 0 and host0:
            sample = host0 if not args.cpu_sample_reads else host0[: args.cpu_sample_reads]
            out["cpu_baseline"] = cpu_baseline(keys, sample, args.mode)
        # ---- the other named single-GPU configurations, in the same run (driver-timed: BENCH_rNN.json carries them) ----
        want = args.configs or ("all" if (world == 1 and args.workload == "c2" and args.mode == "iter" and args.variant == 0 and not args.keys) else "none")
        if want != "none" and world == 1:
            cfgs = {}
            cpu_on = args.cpu_sample_reads != 0
            names = set(ALL_CONFIGS if want == "all" else want.split(","))
            if not os.environ.get("ACX_BENCH_KEEP_HEADLINE"):
                m = None                                      # the headline's scanners go first: their results, side streams and events
            try:
                if "c5_iter_long" in names:
                    cfgs["c5_iter_long"] = other_config(torch, dev, acx, "c5_iter_long", "c2", "iter_long", keys, None, image, batches, host0, args,
                                                        n_keys, args.batch_mb, 200_000 if cpu_on else 0)
            except SystemExit as ex:                          # (a failed sub-configuration must not take the headline line with it)
                cfgs["c5_iter_long"] = {"error": str(ex)}
            del m, batches, host0, e2e0                  # (m: the name; the scanners went above)
            # config 2 as the GENERAL stream kernel sees it: the same reads as an offsets batch of ragged lengths (same image)
            try:
                if "c2_offsets" in names:
                    cfgs["c2_offsets"] = other_config(torch, dev, acx, "c2_offsets", "c2o", "iter", keys, None, image, None, None, args,
                                                      n_keys, args.batch_mb, 100_000 if cpu_on else 0)
            except (SystemExit, Exception) as ex:            # noqa: BLE001
                cfgs["c2_offsets"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            image.free()
            del image, image_tensor
            torch.cuda.empty_cache()
            for name, wl, nk in (("c2_long_keys", "c2k", n_keys), ("c3", "c3", 100_000), ("c4", "c4", 1_000_000)):
                if name not in names:
                    continue
                try:
                    k2, v2 = build_keys(wl, nk)
                    cfgs[name] = other_config(torch, dev, acx, name, wl, "iter", k2, v2, None, None, None, args, nk, 512, 100_000 if cpu_on else 0)
                    del k2, v2
                except (SystemExit, Exception) as ex:        # noqa: BLE001
                    cfgs[name] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                torch.cuda.empty_cache()
            out["configs"] = cfgs
        if args.full_json:
            with open(args.full_json, "w") as f:
                f.write(json.dumps(out) + "\n")
        os.write(json_fd, (json.dumps(out if args.verbose else compact_line(out), separators=(",", ":")) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
