#!/usr/bin/env python3
"""profiles/traffic.json from a rocprofv3 --pmc summary (tools/pmc_summary.py output).

    python tools/make_traffic.py gpurun_out/<tag>_pmc_summary.json <workload>_<mode> <kernel> [note] [haystack bytes per launch]

The entry is tied to the kernel sources by their hash (bench.py:kernel_source_hash): bench.py puts
`roofline.traffic` into its JSON line only when the hash of the sources it runs matches — a PMC
figure measured on other kernels is refused, not silently reused.
Bytes = FETCH_SIZE + WRITE_SIZE of the dominant kernel per launch (separate --pmc passes).
FETCH_SIZE is kept raw (TCC_EA0_RDREQ x 64 B; /opt/skills/guides/MI355X_MICROARCH.md says gfx950
tallies the 128-B requests of a wide coalesced stream at 64 B: the x2 figure is kept beside it as
the upper bound; the stream kernel's haystack loads are 4 B per lane, its gathers 16-32 B).

Round 5: `traffic_calibrated` — ONE figure, from what tools/fetch_calib.hip measured on known byte counts on this chip
(profiles/r5_fetch_calibration.json): FETCH_SIZE tallies 64 B per fabric read request; a request of a coalesced 16-byte
streaming read is 128 B (stream16: 2.00 bytes per counted byte, 128.0 B per TCC_EA0_RDREQ), a request of a missing 8-byte
gather is one 64-B sector (gather8 over 1 GiB: 0.98 requests and 63 B per gather); WRITE_SIZE is exact for the record
stores (store8: 1.00).  The kernels read the haystack's H bytes once with 16-byte non-temporal loads and nothing else in that
pattern, so   traffic = FETCH_SIZE + H / 2 + WRITE_SIZE   (the H / 128 streaming requests counted once more)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def main(summary, key, kernel, note="", hay_bytes=None):
    d = json.load(open(summary))[kernel]
    path = os.path.join(ROOT, "profiles", "traffic.json")
    out = json.load(open(path)) if os.path.exists(path) else {}
    out = {k: v for k, v in out.items() if isinstance(v, dict) and "kernel_source_sha" in v}     # drop entries of older formats
    out[key] = {
        "kernel": kernel, "kernel_source_sha": kernel_source_hash(kernel),
        "hbm_bytes_dominant_kernel": d["fetch_bytes_raw"] + d["write_bytes"],
        "fetch_bytes_raw": d["fetch_bytes_raw"], "fetch_bytes_x2_gfx950": d["fetch_bytes_x2_gfx950"], "write_bytes": d["write_bytes"],
        "l2_requests": d.get("l2_requests"), "l2_hit_rate": d.get("l2_hit_rate"), "ea_rdreq": d.get("TCC_EA0_RDREQ_sum"),
        "source": os.path.basename(summary), "note": note or "rocprofv3 --pmc on tools/microbench.py (same kernels, one batch), per launch",
    }
    if hay_bytes:
        out[key]["haystack_bytes"] = float(hay_bytes)
        out[key]["traffic_calibrated"] = d["fetch_bytes_raw"] + float(hay_bytes) / 2 + d["write_bytes"]
        out[key]["calibration"] = "profiles/r5_fetch_calibration.json: FETCH_SIZE + H / 2 + WRITE_SIZE (streaming requests are 128 B tallied at 64, gather misses 64-B sectors, writes exact)"
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[key], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
