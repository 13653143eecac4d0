#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per counter pass) into per-kernel
per-dispatch averages.

    python tools/pmc_summary.py gpurun_out <TAG>  > gpurun_out/<TAG>_pmc_summary.json

HBM bytes per launch follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE reads exactly half of a wide coalesced
stream (128-B requests tallied at 64 B), so the corrected read traffic is 2 x FETCH_SIZE.
Both the raw and the corrected figure are kept — for this kernel most requests are 4-B
gathers (64-B sectors), where the x2 correction is an upper bound.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(out_dir, tag):
    acc = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [values per dispatch]
    for path in glob.glob(os.path.join(out_dir, "%s_pmc_*" % tag, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                c = row.get("Counter_Name") or row.get("Counter Name")
                v = row.get("Counter_Value") or row.get("Counter Value")
                if c is None or v is None:
                    continue
                acc[k][c].append(float(v))
    out = {}
    for k, cs in acc.items():
        short = k.replace("void ", "").replace("(anonymous namespace)::", "")
        for stop in ("<", "("):
            if stop in short:
                short = short.split(stop)[0]
        short = short.strip() or k
        d = {"dispatches": max(len(v) for v in cs.values())}
        for c, vals in cs.items():
            d[c] = sum(vals) / len(vals)
        if "FETCH_SIZE" in d:
            d["fetch_bytes_raw"] = d["FETCH_SIZE"] * 1024
            d["fetch_bytes_x2_gfx950"] = 2 * d["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in d:
            d["write_bytes"] = d["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
            d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
            d["l2_requests"] = d["TCC_HIT_sum"] + d["TCC_MISS_sum"]
        out[short] = d
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
