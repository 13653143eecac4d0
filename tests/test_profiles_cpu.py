"""The committed measurement artefacts agree with each other (no GPU): every `roofline.frac` of the committed driver-format line lies
inside what the committed rocprofv3 traces support (`tools/roofline_check.py`), the line carries what the bench contract asks for, and
the PMC traffic entries belong to the kernel sources in the tree."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r6_bench.json")).read().strip().splitlines()[-1])


def test_union_of_spans():
    import roofline_check as rc
    assert rc.union_us([(0, 1000), (500, 2000), (3000, 4000)]) == 3.0      # ns in, us out; overlapping spans count once
    assert rc.union_us([]) == 0 and rc.union_us([(5, 5)]) == 0
    assert rc.union_us([(0, 4000), (1000, 2000)]) == 4.0                    # a span inside another


def test_committed_line_is_inside_what_the_committed_traces_support():
    spans = sorted(glob.glob(os.path.join(ROOT, "profiles", "r6_c*_spans.json")))
    assert len(spans) == 6
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_check.py"), "check", os.path.join(ROOT, "profiles", "r6_bench.json"), *spans],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    out = p.stdout.decode()
    assert p.returncode == 0, out
    assert out.count(" ok") == 6 and "outside" not in out


def test_committed_line_keeps_the_bench_contract():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["unit"] == "GB/s" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-5      # (the compact line rounds `achieved` to 0.1 GB/s and `frac` to five digits)
    assert r["traffic"] and r["traffic"] > r["algorithmic_bytes"]           # calibrated HBM bytes per launch: above the algorithmic ones
    assert abs(r["step"]["frac"] - r["step"]["algorithmic_bytes"] / (d["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 2e-3
    # value = bytes of one step / its time: config 2's 150 MB batches
    assert abs(d["value"] - 150e6 / (d["ms_per_step"] * 1e-3) / 1e9) / d["value"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] == 1 and c["value"] < 1.0 and "sample" in c
    assert set(d["configs"]) == {"c5_iter_long", "c2_offsets", "c2_long_keys", "c3", "c4"}
    for name, v in d["configs"].items():
        assert v["value"] > 100 and 0 < v["roofline"]["frac"] < 1 and "cpu_baseline" in v, name


def test_the_drivers_line_fits_the_tail_the_driver_keeps():
    """the driver keeps an 8 KB tail of the output and, in `parsed`, the headline's `config` / `roofline` / `cpu_baseline`:
    the committed line (the default command's, as the driver runs it) is below 7000 characters, `config.all` names all six configurations
    with [GB/s, roofline.frac], and compacting the FULL object of the same run gives a line of the same shape"""
    raw = open(os.path.join(ROOT, "profiles", "r6_bench.json")).read().strip().splitlines()[-1]
    assert len(raw) < 7000, len(raw)
    d = json.loads(raw)
    assert set(d["config"]["all"]) == {"headline", "c5_iter_long", "c2_offsets", "c2_long_keys", "c3", "c4"}
    for name, (gbps, frac) in d["config"]["all"].items():
        ent = d if name == "headline" else d["configs"][name]
        assert abs(gbps - ent["value"]) < 0.06 and abs(frac - ent["roofline"]["frac"]) < 6e-5, name
    q = d["config"]["queues"]
    assert q["scan_streams"] == 3 and q["results_in_flight"] == 3 and q["side_streams"] == 3 and q["GPU_MAX_HW_QUEUES"] == "8"
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r6_bench_full.json")).read().strip().splitlines()[-1])
    again = json.dumps(bench.compact_line(full), separators=(",", ":"))
    assert len(again) < 7000 and json.loads(again)["config"]["all"] == d["config"]["all"]
    assert len(json.dumps(full)) > 12000                      # (what the line was before: beyond the driver's tail)


def test_traffic_entries_belong_to_the_kernel_sources_in_the_tree():
    """bench.py refuses a PMC entry whose kernel_source_sha is not the hash of the kernel's files: the committed entries are current."""
    sys.path.insert(0, ROOT)
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    seen = 0
    for key, ent in t.items():
        if not isinstance(ent, dict) or "kernel_source_sha" not in ent or "kernel" not in ent:
            continue
        if key in ("c2_iter", "c2_iter_long", "c3_iter", "c4_iter"):
            assert ent["kernel_source_sha"] == bench.kernel_source_hash(ent["kernel"]), key
            seen += 1
    assert seen >= 3
