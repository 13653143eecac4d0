#!/usr/bin/env python3
"""profiles/traffic.json from a rocprofv3 --pmc summary (tools/pmc_summary.py output).

    python tools/make_traffic.py gpurun_out/<tag>_pmc_summary.json <workload>_<mode> <kernel> [note]

The entry is tied to the kernel sources by their hash (bench.py:kernel_source_hash): bench.py puts
`roofline.traffic` into its JSON line only when the hash of the sources it runs matches — a PMC
figure measured on other kernels is refused, not silently reused.
Bytes = FETCH_SIZE + WRITE_SIZE of the dominant kernel per launch (separate --pmc passes).
FETCH_SIZE is kept raw (TCC_EA0_RDREQ x 64 B; /opt/skills/guides/MI355X_MICROARCH.md says gfx950
tallies the 128-B requests of a wide coalesced stream at 64 B: the x2 figure is kept beside it as
the upper bound; the stream kernel's haystack loads are 4 B per lane, its gathers 16-32 B)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402


def main(summary, key, kernel, note=""):
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
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[key], indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
