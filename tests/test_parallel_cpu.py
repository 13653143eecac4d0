"""CPU suite: the N>1 path with world_size 2 over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np

from pyahocorasick_amd.parallel import shard_range, shard_range_by_bytes

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_and_balance():
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    off = np.concatenate([[0], np.cumsum([10] * 50 + [1000] * 5 + [10] * 45)])
    spans = [shard_range_by_bytes(off, r, 4) for r in range(4)]
    assert spans[0][0] == 0 and spans[-1][1] == 100 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    byts = [int(off[hi] - off[lo]) for lo, hi in spans]
    assert max(byts) <= 2 * (off[-1] // 4) + 1000


def test_world_size_2_gloo_broadcast_shard_gather():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "_parallel_worker.py")]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
    out = p.stdout.decode(errors="replace")
    assert p.returncode == 0 and "PARALLEL_CPU_OK" in out, out[-3000:]


def test_halo_shard_ranges():
    from pyahocorasick_amd.parallel import halo_shard
    for total in (0, 1, 100, 12345):
        for w in (1, 2, 8):
            cuts = [halo_shard(total, r, w, 32) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[0][1] == 0 and cuts[-1][2] == total
            assert all(a[2] == b[1] for a, b in zip(cuts, cuts[1:]))
            assert all(lo - s0 == min(lo, 31) for s0, lo, hi in cuts)


def test_bench_strong_scaling_shards_are_disjoint_and_cover_the_corpus():
    """`bench.py --scaling strong`: every rank generates the same corpus and keeps its share — packets (config 4) cut so
    that the shares are balanced by BYTES, reads by count, text with a halo.  Shares laid end to end are the corpus."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from pyahocorasick_amd import workloads as W
    sigs = W.snort_signatures(2000, seed=5)
    data, off = W.packet_payloads(sigs, 4 << 20, seed=7)
    for world in (1, 2, 3, 8):
        parts = [bench.strong_shard("c4", data, off, r, world, 128) for r in range(world)]
        assert all(o[0] == 0 and o[-1] == len(d) and np.all(np.diff(o) > 0) for d, o in parts)
        assert np.array_equal(np.concatenate([d for d, _ in parts]), data)                       # disjoint, in order, complete
        assert np.array_equal(np.concatenate([[0]] + [o[1:] + sum(len(x[0]) for x in parts[:i]) for i, (_, o) in enumerate(parts)]), off)
        sizes = [len(d) for d, _ in parts]
        assert max(sizes) - min(sizes) <= 2 * 1500, sizes                                          # balanced by bytes (a packet is at most 1500)
    reads = W.dna_reads(W.dna_keys(50, seed=0), 1001, 150, seed=1)
    for world in (2, 8):
        parts = [bench.strong_shard("c2", reads, None, r, world, 32)[0] for r in range(world)]
        assert np.array_equal(np.concatenate(parts), reads)
    text = np.frombuffer(bytes(range(256)) * 40, dtype=np.uint8)
    for world in (2, 8):
        parts = [bench.strong_shard("c3", text, None, r, world, 32) for r in range(world)]
        own = [d[(0 if r == 0 else 31):] for r, (d, _) in enumerate(parts)]                       # without its halo a share is the rank's own range
        assert np.array_equal(np.concatenate(own), text) and all(o[-1] == len(d) for d, o in parts)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start 2 ranks itself (the driver's 8-GPU
    run is launched exactly like that).  --launch-check stops after the ranks have met (gloo, no GPU)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    out = p.stdout.decode(errors="replace").strip().splitlines()
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    line = json.loads([x for x in out if x.startswith("{")][-1])
    assert line == {"n_gpus": 2, "launch_check": True}
    # a rank count that does not match --gpus is an error, not a silent N = 1 line
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p2 = subprocess.run([sys.executable, bench, "--gpus", "2", "--launch-check"], env=env2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p2.returncode != 0 and b"WORLD_SIZE=1" in p2.stderr


def _dry_run(*extra, expect_fail=False):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run", "--keys", "1500", *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    if expect_fail:
        assert p.returncode != 0
        return p.stderr.decode(errors="replace")
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    return json.loads(p.stdout.decode(errors="replace").strip().splitlines()[-1])


def test_bench_dry_run_strong_scaling_covers_one_corpus():
    """`bench.py --gpus 2 --scaling strong --dry-run` (gloo, no GPU): the blob reaches every rank by one broadcast and validates,
    the ranks' shards add up to ONE corpus — packets (config 4, iter_long: whole haystacks, no halo) and reads exactly, text
    plus the longest_word - 1 bytes of halo of every rank but the first."""
    d = _dry_run("--workload", "c4", "--batch-mb", "2", "--batches", "2", "--scaling", "strong", "--mode", "iter_long")
    assert d["dry_run"] and d["n_gpus"] == 2 and d["scaling"] == "strong" and d["blob_bytes_equal_on_all_ranks"]
    assert d["bytes_total"] == d["corpus_bytes_all_batches"] and len(d["per_rank"]) == 2
    sizes = [r["bytes"] for r in d["per_rank"]]
    assert abs(sizes[0] - sizes[1]) <= 2 * 2 * 1500                     # balanced by bytes, two batches
    d = _dry_run("--reads", "4001", "--batches", "2", "--scaling", "strong", "--mode", "iter_long")
    assert d["bytes_total"] == d["corpus_bytes_all_batches"] == 2 * 4001 * 150 and d["haystacks_total"] == 2 * 4001
    assert d["per_rank"][0]["shard_sha48"] != d["per_rank"][1]["shard_sha48"]
    d = _dry_run("--workload", "c3", "--batch-mb", "2", "--batches", "1", "--scaling", "strong")
    halo = d["bytes_total"] - d["corpus_bytes_all_batches"]
    assert 0 < halo < 256 and d["haystacks_total"] == 2                 # one rank's halo: longest_word - 1 bytes
    # weak scaling: every rank its own batches of the full size
    d = _dry_run("--reads", "3000", "--batches", "1")
    assert d["scaling"] == "weak" and d["bytes_total"] == 2 * 3000 * 150 and d["per_rank"][0]["shard_sha48"] != d["per_rank"][1]["shard_sha48"]
    # iter_long does not shard INSIDE a haystack
    err = _dry_run("--workload", "c3", "--batch-mb", "2", "--batches", "1", "--scaling", "strong", "--mode", "iter_long", expect_fail=True)
    assert "does not shard" in err


def test_bench_configs_takes_all_none_or_a_list_of_names():
    """`--configs` names which of the other single-GPU configurations the default line measures: all, none or a comma list; a name that
    is none of them is refused when the arguments are parsed (before any rank starts)."""
    bench = os.path.join(os.path.dirname(HERE), "bench.py")
    p = subprocess.run([sys.executable, bench, "--configs", "c5_iter_long,c9", "--dry-run"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"--configs" in p.stderr and b"c2_long_keys" in p.stderr
    p = subprocess.run([sys.executable, bench, "--configs", "c5_iter_long,c3", "--launch-check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
