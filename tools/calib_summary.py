#!/usr/bin/env python3
"""tools/fetch_calib.bin under rocprofv3 --pmc -> what one unit of each memory-side counter is worth for the access patterns of
the scan kernels (profiles/r5_fetch_calibration.json; tools/make_traffic.py applies it).

    python tools/calib_summary.py gpurun_out <TAG>_calib gpurun_out/<TAG>_calib_known.jsonl
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main(out_dir, tag, known_path):
    known = {}
    for ln in open(known_path):
        if ln.startswith("{"):
            d = json.loads(ln)
            known[d["kernel"]] = d                              # (the second repetition of every launch overwrites the first)
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(out_dir, "%s_pmc_*" % tag, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = (row.get("Kernel_Name") or row.get("Kernel Name") or "").replace("void ", "").split("(")[0].strip()
                c = row.get("Counter_Name") or row.get("Counter Name")
                v = row.get("Counter_Value") or row.get("Counter Value")
                if c and v is not None:
                    acc[k][c].append(float(v))
    out = {"kernels": {}, "source": "tools/fetch_calib.hip under rocprofv3 --pmc (separate passes), means over its two launches of each kernel"}
    for k, cs in sorted(acc.items()):
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        d.update({"known": known.get(k)})
        out["kernels"][k] = d
    K = out["kernels"]
    s = K.get("calib_stream16", {})
    if s.get("known") and "FETCH_SIZE" in s:
        n = s["known"]["bytes"]
        out["stream16"] = {"bytes": n, "bytes_per_FETCH_SIZE_KiB": n / (s["FETCH_SIZE"] * 1024.0),
                           "bytes_per_RDREQ": n / s["TCC_EA0_RDREQ_sum"] if s.get("TCC_EA0_RDREQ_sum") else None,
                           "RDREQ_32B_share": s.get("TCC_EA0_RDREQ_32B_sum", 0) / s["TCC_EA0_RDREQ_sum"] if s.get("TCC_EA0_RDREQ_sum") else None}
    for t in (0, 1, 2):
        g = K.get("calib_gather8<%d>" % t, {})
        if g.get("known") and "FETCH_SIZE" in g:
            n = g["known"]["gathers"]
            out["gather8_table_%d_MiB" % (g["known"]["table_bytes"] >> 20)] = {
                "gathers": n, "FETCH_SIZE_bytes_per_gather": g["FETCH_SIZE"] * 1024.0 / n,
                "RDREQ_per_gather": g.get("TCC_EA0_RDREQ_sum", 0) / n, "RDREQ_32B_per_gather": g.get("TCC_EA0_RDREQ_32B_sum", 0) / n,
                "l2_miss_per_gather": g.get("TCC_MISS_sum", 0) / n, "l2_requests_per_gather": (g.get("TCC_MISS_sum", 0) + g.get("TCC_HIT_sum", 0)) / n}
    w = K.get("calib_store8", {})
    if w.get("known") and "WRITE_SIZE" in w:
        n = w["known"]["bytes"]
        out["store8"] = {"bytes": n, "bytes_per_WRITE_SIZE_KiB": n / (w["WRITE_SIZE"] * 1024.0),
                         "bytes_per_WRREQ": n / w["TCC_EA0_WRREQ_sum"] if w.get("TCC_EA0_WRREQ_sum") else None,
                         "WRREQ_64B_share": w.get("TCC_EA0_WRREQ_64B_sum", 0) / w["TCC_EA0_WRREQ_sum"] if w.get("TCC_EA0_WRREQ_sum") else None}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(*sys.argv[1:4])
