#!/usr/bin/env python3
"""Recompute every `roofline.frac` of a bench line from rocprofv3's own clock and fail when they disagree.

Two steps, because the rocprofv3 database stays on the GPU box and profiles/ holds small text files:

  export  (GPU box, after `rocprofv3 --kernel-trace --stats -- python bench.py <one config>`):
      python tools/roofline_check.py export <stats_results.db> <bench line of that run> <out.json> <name>
    (name: where the driver's line holds this configuration — headline, c5_iter_long, c2_offsets, c2_long_keys, c3, c4)
    writes, for the dominant kernel the bench line names: launches, the average launch span, and the UNION of its launch
    spans over the last `passes` launches (the timed region) per launch — with two scan streams the spans of consecutive
    launches overlap, the average span then counts shared time twice (339 us where the kernel alone takes 257) and only
    the union per launch is a duration the algorithmic bytes can be divided by; with one stream union == average.

  check   (anywhere):
      python tools/roofline_check.py check <bench line (driver format)> <spans.json> [<spans.json> ...]
    for the headline and every entry of `configs` that a spans file names:  algorithmic bytes per launch / duration / 8 TB/s
    from the trace's durations, compared with the line's `roofline.frac` (see check() for what "duration" is when launches
    overlap); exit status 1 when a line's figure is outside what the trace supports by more than 5 % (--tol).  The algorithmic bytes are the line's own `roofline.algorithmic_bytes` (H + 8 M + 4 N for
    the stream kernels on fixed-stride batches, H + 8 M + 12 N on offsets batches, H + 12 N for iter_long's scan).
"""
import json
import sqlite3
import sys

PEAK = 8000.0   # GB/s, /opt/skills/guides/MI355X_MICROARCH.md


def _last_json_line(path):
    lines = [x for x in open(path).read().strip().splitlines() if x.startswith("{")]
    return json.loads(lines[-1])


def _entry(line, config):
    if config in (None, "", "headline", "c2"):
        return line
    return line["configs"][config]


def union_us(spans):
    spans = sorted(spans)
    tot, cur_s, cur_e = 0, None, None
    for s, e in spans:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot / 1e3


def export(db, bench_path, out_path, config=None):
    line = _last_json_line(bench_path)
    # (the profiled command runs this configuration alone — its line's top level — or, round 6, behind the headline in ONE process as the
    #  driver's command does: `--configs <name>`, its entry of `configs`; its launches are the trace's last then)
    ent = line["configs"][config] if (config and config in (line.get("configs") or {})) else line
    kernel = ent["roofline"]["kernel"]
    passes = int(ent.get("passes") or ent["steps"] * ent.get("inner_repeats", 1))
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    src = "kernels" if "kernels" in names else next(n for n in names if "kernel_dispatch" in n)
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % src)]
    name_col = "name" if "name" in cols else next(x for x in cols if "name" in x)
    rows = c.execute("select %s, start, end from %s order by start" % (name_col, src)).fetchall()
    short_name = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].strip()
    mine = [(s, e) for n, s, e in rows if short_name(n) == kernel]
    if not mine:
        raise SystemExit("no launches of %s in %s" % (kernel, db))
    timed = mine[-passes:] if len(mine) >= passes else mine
    every = {}
    for n, s, e in rows:
        short = short_name(n)
        d = every.setdefault(short, [0, 0.0])
        d[0] += 1
        d[1] += (e - s) / 1e3
    out = {
        "config": config or "headline", "kernel": kernel, "launches_in_trace": len(mine), "timed_launches": len(timed),
        "avg_span_us_all": sum(e - s for s, e in mine) / len(mine) / 1e3,
        "avg_span_us_timed": sum(e - s for s, e in timed) / len(timed) / 1e3,
        "union_us_timed": union_us(timed), "union_per_launch_us": union_us(timed) / len(timed),
        "timed_region_us": (max(e for _, e in timed) - min(s for s, _ in timed)) / 1e3,
        "overlapping": any(timed[k + 1][0] < timed[k][1] for k in range(len(timed) - 1)),
        "kernels": {k: {"calls": v[0], "total_us": round(v[1], 1), "avg_us": round(v[1] / v[0], 2)} for k, v in sorted(every.items(), key=lambda kv: -kv[1][1])[:12]},
        "bench_line_of_the_profiled_run": {"value": ent["value"], "ms_per_step": ent["ms_per_step"], "frac": ent["roofline"]["frac"],
                                           "algorithmic_bytes": ent["roofline"]["algorithmic_bytes"], "scan_streams": ent.get("scan_streams")},
        "source": "rocprofv3 --kernel-trace --stats of `python bench.py` for this configuration alone (tools/r5_profiles.sh)",
    }
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("config", "kernel", "timed_launches", "avg_span_us_timed", "union_per_launch_us", "overlapping")}))


# round 6: ONE trace of the driver's own command holds every configuration — their dominant kernels are different instantiations
SIGNATURES = {
    "headline": "k_ppm_stream4<false, false>", "c5_iter_long": "k_ppm_stream4<true, false>",
    "c2_offsets": "k_ppm_stream4<false, true>", "c2_long_keys": "k_ppm_stream<2, 8, true, false, false, true, true, 6>",
    "c3": "k_ppm_stream<8, 8, false, true, false, false, false, 4>", "c4": "k_ppm_stream<8, 4, true, true, true, false, false, 4>",
}
UNION_LEG_LAUNCHES = 36          # bench.py measure(): 12 x P launches with event pairs BEHIND every timed region (P = 3 results in flight)


def export_all(db, bench_path, out_prefix):
    """spans of every configuration of the line from a rocprofv3 --kernel-trace of `python bench.py --no-e2e --cpu-sample-reads 0` (the
    driver's command without its CPU and host-to-host legs): the timed launches of a configuration are the last `passes` launches of ITS
    instantiation in front of the union leg's."""
    line = _last_json_line(bench_path)
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    src = "kernels" if "kernels" in names else next(n for n in names if "kernel_dispatch" in n)
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % src)]
    name_col = "name" if "name" in cols else next(x for x in cols if "name" in x)
    rows = c.execute("select %s, start, end from %s order by start" % (name_col, src)).fetchall()
    for config, sig in SIGNATURES.items():
        ent = line if config == "headline" else (line.get("configs") or {}).get(config)
        if not ent or "roofline" not in ent:
            continue
        mine = [(s, e) for n, s, e in rows if sig in n]
        if not mine:
            print("%s: no launches of %s" % (config, sig)); continue
        passes = int(ent.get("passes") or ent["steps"] * ent.get("inner_repeats", 1))
        leg = 12 * int(ent.get("results_in_flight") or (ent.get("config") or {}).get("queues", {}).get("results_in_flight") or 3)      # (iter_long: --long-depth results in flight)
        tail = leg if len(mine) >= passes + leg else 0
        timed = mine[-(passes + tail):len(mine) - tail] if len(mine) >= passes + tail else mine
        out = {
            "config": config, "kernel": ent["roofline"]["kernel"], "instantiation": sig, "launches_in_trace": len(mine), "timed_launches": len(timed),
            "avg_span_us_timed": sum(e - s for s, e in timed) / len(timed) / 1e3,
            "union_us_timed": union_us(timed), "union_per_launch_us": union_us(timed) / len(timed),
            "timed_region_us": (max(e for _, e in timed) - min(s for s, _ in timed)) / 1e3,
            "overlapping": any(timed[k + 1][0] < timed[k][1] for k in range(len(timed) - 1)),
            "bench_line_of_the_profiled_run": {"value": ent["value"], "ms_per_step": ent["ms_per_step"], "frac": ent["roofline"]["frac"],
                                               "algorithmic_bytes": ent["roofline"]["algorithmic_bytes"]},
            "source": "rocprofv3 --kernel-trace --stats of `python bench.py --no-e2e --cpu-sample-reads 0`: ALL configurations in one process, as the driver runs them (tools/r6_trace_all.sh)",
        }
        json.dump(out, open("%s_%s_spans.json" % (out_prefix, {"headline": "c2", "c5_iter_long": "c5", "c2_offsets": "c2o", "c2_long_keys": "c2k"}.get(config, config)), "w"), indent=1)
        print(json.dumps({k: out[k] for k in ("config", "timed_launches", "avg_span_us_timed", "union_per_launch_us", "timed_region_us")}))


def check(bench_path, span_paths, tol=0.05):
    """With launches that do not overlap (one scan stream) the trace gives ONE duration per launch and the line's frac must be within
    `tol` of algorithmic bytes / that / peak.  With overlapping launches it gives two bounds of the time the kernel occupies the chip per
    launch: from below the union of its launch spans (rocprofv3 starts a span when the first wave runs), from above timed region /
    launches (which includes the time only gathers run).  bench.py measures in between — its events mark when a launch reaches the head
    of its stream, not when its first wave starts —, and its frac must lie between the two (each widened by `tol`)."""
    line = _last_json_line(bench_path)
    bad = 0
    print("%-14s %-16s %10s %10s %10s | %9s %9s %9s  %s" % ("config", "kernel", "alg. MB", "union us", "region us", "frac lo", "line frac", "frac hi", ""))
    for sp in span_paths:
        s = json.load(open(sp))
        try:
            ent = _entry(line, s["config"])
        except KeyError:
            print("%-14s not in the bench line" % s["config"])
            bad += 1
            continue
        r = ent["roofline"]
        if r["kernel"] != s["kernel"]:
            print("%-14s the line names %s, the trace %s" % (s["config"], r["kernel"], s["kernel"]))
            bad += 1
            continue
        region = s["timed_region_us"] / s["timed_launches"]
        hi = r["algorithmic_bytes"] / (s["union_per_launch_us"] * 1e-6) / 1e9 / PEAK
        lo = r["algorithmic_bytes"] / (max(region, s["union_per_launch_us"]) * 1e-6) / 1e9 / PEAK
        if not s["overlapping"]:
            lo = hi
        ok = lo * (1 - tol) <= r["frac"] <= hi * (1 + tol)
        print("%-14s %-16s %10.1f %10.2f %10.2f | %9.4f %9.4f %9.4f  %s" % (s["config"], s["kernel"], r["algorithmic_bytes"] / 1e6, s["union_per_launch_us"], region,
                                                                         lo, r["frac"], hi, "ok" if ok else "<-- outside what the trace supports"))
        bad += not ok
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "export":
        export(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    elif len(sys.argv) >= 5 and sys.argv[1] == "export_all":
        export_all(sys.argv[2], sys.argv[3], sys.argv[4])
    elif len(sys.argv) >= 4 and sys.argv[1] == "check":
        args = [a for a in sys.argv[2:] if not a.startswith("--tol")]
        tol = next((float(a.split("=")[1]) for a in sys.argv[2:] if a.startswith("--tol=")), 0.05)
        sys.exit(check(args[0], args[1:], tol))
    else:
        print(__doc__)
        sys.exit(2)
