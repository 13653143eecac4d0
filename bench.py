#!/usr/bin/env python3
"""bench.py — headline benchmark of the batch Aho-Corasick scan on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
100,000 unique ACGT keys of length U[8,32] (seed 0) -> one flattened automaton;
1,000,000 x 150 B DNA-style reads (seed 1 + rank; every even read has a planted key),
resident in HBM before the timed region.  A "step" is one pass of the hot path (walk +
prefix-sum + expand kernels) over that batch, results left in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; rank 0 builds + flattens the automaton and the flat image is
replicated with ONE RCCL broadcast; every rank scans its own batch (weak scaling, no
data-path collective); time = max over ranks, value = all ranks' bytes / time.

Prints one JSON line (rank 0).  `roofline` and `cpu_baseline` are explained in DESIGN.md.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keys", type=int, default=100_000)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--mode", choices=["iter", "iter_long"], default="iter")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--pipeline", type=int, default=2,
                    help="result objects kept in flight per GPU (ACX_SCAN_ASYNC): the host queues step i+1 and reads "
                         "the counters of step i-1 while step i runs.  2 (default) = double-buffered results; on ONE "
                         "stream the kernels of consecutive steps still run strictly one after the other.  "
                         "1 = one synchronous call per step (the host's bookkeeping between two steps is then exposed)")
    ap.add_argument("--streams", type=int, default=1,
                    help="streams the in-flight steps are spread over.  1 (default): every kernel still runs alone, "
                         "one after the other, so kernel times are standalone durations.  2 lets consecutive steps "
                         "overlap on the GPU (needs --pipeline >= 2; +5-13 %% measured: 180-194 GB/s) but concurrent "
                         "kernels stretch each other and the per-kernel roofline is then not a standalone figure")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000,
                    help="reads timed on the CPU baseline leg (0 disables it)")
    ap.add_argument("--verify", action="store_true", help="check a sample of the GPU output against the oracle")
    return ap.parse_args()


def cpu_baseline(keys, reads, n_sample, mode):
    """Time the reference itself (oracle/_ref, kind 'reference') or, if the prebuilt module is
    absent, the plain-C port (oracle/ac_oracle.c, kind 'port') on ONE host core over the first
    n_sample reads of the same batch.  Baseline only — never on the product path."""
    from oracle import orc
    sample = [reads[i].tobytes() for i in range(n_sample)]
    nbytes = sum(len(s) for s in sample)
    ref = orc.load_reference()
    if ref is not None:
        A = ref.Automaton(ref.STORE_INTS)
        for i, k in enumerate(keys):
            A.add_word(k, i)
        A.make_automaton()
        scan = A.iter if mode == "iter" else A.iter_long
        for r in sample[:2000]:          # warm-up pass, discarded
            for _ in scan(r):
                pass
        n = 0
        t0 = time.perf_counter()
        for r in sample:
            for _ in scan(r):
                n += 1
        dt = time.perf_counter() - t0
        kind = "reference"
    else:
        O = orc.Oracle()
        for i, k in enumerate(keys):
            O.add_word(k, i)
        O.make_automaton()
        data = b"".join(sample)
        off = np.arange(n_sample + 1, dtype=np.int64) * len(sample[0])
        O.batch_count(data[: off[2000]], off[:2001], 0 if mode == "iter" else 1)
        t0 = time.perf_counter()
        n = O.batch_count(data, off, 0 if mode == "iter" else 1)
        dt = time.perf_counter() - t0
        kind = "port"
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": 1, "kind": kind,
            "matches_per_s": n / dt, "seconds": round(dt, 3),
            "host_cpus": os.cpu_count(),
            "sample": "first %d of the %d reads (%d B each, %.1f MB), Automaton.%s drained per read, 1 core"
                      % (n_sample, len(reads), len(sample[0]), nbytes / 1e6, mode)}


def main():
    args = parse()
    # stdout must carry exactly ONE JSON line: RCCL prints a version banner on fd 1 when its
    # communicator comes up, so everything until the final print goes to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # under torch.distributed.run (RANK/MASTER_ADDR set) the process group is created even for one
    # rank, so the N = 1 line and the N > 1 lines go through the same code (RCCL broadcast, barriers)
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    import pyahocorasick_amd as acx
    from pyahocorasick_amd import _lib
    from pyahocorasick_amd.device import Image, Scanner
    from pyahocorasick_amd.parallel import broadcast_image
    from pyahocorasick_amd.workloads import dna_keys, dna_reads
    _lib.check(_lib.lib().acx_device_set(local_rank))

    # ---- automaton: built on rank 0 (CPU), replicated by one RCCL broadcast -------------
    keys = dna_keys(args.keys, seed=0)
    t0 = time.perf_counter()
    blob = None
    if rank == 0:
        A = acx.Automaton(acx.STORE_INTS)
        for i, k in enumerate(keys):
            A.add_word(k, i)
        A.make_automaton()
        blob = A.flat_image_bytes()
    t_build = time.perf_counter() - t0
    t0 = time.perf_counter()
    image, image_tensor = broadcast_image(blob, src=0, device=dev)
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0

    # ---- this rank's batch, resident in HBM ----------------------------------------------
    reads = dna_reads(keys, args.reads, args.read_len, seed=1 + rank)
    n, L = reads.shape
    d_hay = torch.empty(n * L + 64, dtype=torch.uint8, device=dev)
    d_hay[: n * L].copy_(torch.from_numpy(reads.reshape(-1)))
    torch.cuda.synchronize()
    mode = acx.ACX_SCAN_ALL if args.mode == "iter" else acx.ACX_SCAN_LONG
    # every step is one complete batch scan (walk + prefix sum + expand -> match records in HBM) of
    # the same resident batch; with --pipeline P, P result objects on P streams are kept in flight
    P = max(1, args.pipeline)
    scs = [Scanner(image) for _ in range(P)]
    sc = scs[0]
    stream = torch.cuda.current_stream().cuda_stream
    S = max(1, min(args.streams, P))
    tstreams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else []
    streams = [t.cuda_stream for t in tstreams] if S > 1 else [stream]

    def step(timing=False, k=0):
        return scs[k % P].scan(d_hay.data_ptr(), n * L, n, stride=L, mode=mode, timing=timing, variant=args.variant,
                               stream=streams[k % P % S], asynchronous=P > 1)

    for k in range(max(args.warmup, P)):
        step(k=k)
    for x in scs:
        x.wait()
    # per-kernel times from HIP events on each scan's stream: on one stream they are taken again over the
    # timed steps below (these passes then only warm up); with several streams these separate passes, pipelined
    # like the timed loop (a walk shares the GPU with the previous step's expand, as it does there), are reported
    kt = {"walk": [], "scan": [], "expand": [], "total": []}
    for _ in range(3):
        for k in range(P):
            step(timing=True, k=k)
        for x in scs:
            x.wait()
            t = x.timing_ms()
            for k in kt:
                kt[k].append(t[k])
    matches = sc.num_matches()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    if S == 1:
        # the default: every kernel runs alone on ONE stream, each timed step with HIP events around its kernels
        # on that stream -> the per-kernel averages of THIS timed region (roofline.kernel_avg_ms).  A result
        # object's times are read when it is about to be reused, i.e. while a later step runs.
        # Only the dominant kernel is bracketed inside the timed steps (timing = 2): every event between two
        # kernels costs ~5 us of idle GPU; the scan and expand averages stay those of the passes above.
        pre = {k: float(np.mean(v)) for k, v in kt.items()}
        kt = {"walk": []}

        def collect(x):
            x.wait()
            kt["walk"].append(x.timing_ms()["walk"])

        for k in range(args.steps):
            if k >= P:
                collect(scs[k % P])
            step(timing=2, k=k)
        for k in range(max(0, args.steps - P), args.steps):
            collect(scs[k % P])                    # every step complete: totals read, records in HBM
    else:
        for k in range(args.steps):
            step(k=k)
        for x in scs:
            x.wait()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mm = torch.tensor([matches], dtype=torch.int64, device=dev)
        dist.all_reduce(mm, op=dist.ReduceOp.SUM)
        matches_all = int(mm.item())
    else:
        matches_all = matches

    if args.verify and rank == 0:
        from oracle import orc
        O = orc.Oracle()
        for i, k in enumerate(keys):
            O.add_word(k, i)
        O.make_automaton()
        off, e, v, _ = sc.fetch()
        for h in range(0, n, max(1, n // 2000)):
            oe, ov, _ = O.iter_arrays(reads[h].tobytes()) if args.mode == "iter" else (None, None, None)
            if oe is not None:
                assert np.array_equal(e[off[h]:off[h + 1]], oe) and np.array_equal(v[off[h]:off[h + 1]], ov), h

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        H = n * L                                     # haystack bytes per rank per step
        A_bytes = H + 8 * matches + 12 * n            # H + 8*M + 12*N
        med = {k: float(np.mean(v)) for k, v in kt.items()}      # averages (one stream: the walk over the timed steps themselves)
        if "scan" not in med:
            med.update({"scan": pre["scan"], "expand": pre["expand"], "total": med["walk"] + pre["scan"] + pre["expand"]})
        if args.mode != "iter":
            walk_kernel = "k_walk_long"
        elif image.itop_depth > 0 and not (args.variant >> 16) & 1:
            walk_kernel = "k_walk_itop"          # implicit top-of-trie walk
        else:
            walk_kernel = "k_walk_all"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                # measured offline with rocprofv3 --pmc (tools/gpu_round.sh stage pmc); bytes per launch
                # of the dominant kernel that crossed the L2 -> fabric boundary (FETCH_SIZE + WRITE_SIZE)
                ent = json.load(open(tpath)).get("%s_n%d_l%d_k%d" % (args.mode, n, L, args.keys))
                traffic = ent["hbm_bytes_dominant_kernel"] if ent else None
            except Exception:
                traffic = None
        out = {
            "metric": "GB/s haystack scanned, 100k-pattern automaton",
            "value": world * H / (dt / args.steps) / 1e9,
            "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "matches_per_s": matches_all / (dt / args.steps),
            "matches_per_step": matches_all,
            "config": {"workload": "config2: %d ACGT keys 8-32 B, %d x %d B reads per GPU, Automaton.%s"
                                   % (args.keys, n, L, args.mode),
                       "states": int(image.num_states), "classes": int(image.num_classes), "itop_depth": int(image.itop_depth),
                       "image_mb": round(image.nbytes / 1e6, 1), "variant": args.variant, "pipeline_depth": P, "streams": S,
                       "parallelism": "replicated automaton (1 RCCL broadcast), reads sharded x%d" % world},
            # dominant kernel = the walk: it alone reads the haystack (H) and the per-haystack
            # bookkeeping (12 B x N); the 8 B x M match records are written by k_expand.
            "roofline": {
                "bound": "hbm", "kernel": walk_kernel,
                "achieved": (H + 12 * n) / (med["walk"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (H + 12 * n) / (med["walk"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": H + 12 * n, "kernel_avg_ms": round(med["walk"], 4),
                "traffic": traffic,
                # the whole batch scan (walk + prefix sum + expand), A = H + 8*M + 12*N
                "pipeline": {"algorithmic_bytes": A_bytes, "gpu_ms": round(med["total"], 4),
                             "achieved": A_bytes / (med["total"] * 1e-3) / 1e9,
                             "frac": A_bytes / (med["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "kernel_ms": {k: round(v, 4) for k, v in med.items()}},
            },
            "setup": {"build_flatten_s": round(t_build, 3), "broadcast_upload_s": round(t_bcast, 3)},
        }
        if world == 1 and args.cpu_sample_reads > 0:
            out["cpu_baseline"] = cpu_baseline(keys, reads, min(args.cpu_sample_reads, n), args.mode)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
