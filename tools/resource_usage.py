#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel of libacx as the compiler reports them (no GPU needed):
    python tools/resource_usage.py > profiles/r5_resource_usage.txt
(hipcc -Rpass-analysis=kernel-resource-usage on every .hip source with the flags of pyahocorasick_amd/build.py)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pyahocorasick_amd", "csrc")
FIELDS = [("TotalSGPRs", "SGPR"), ("VGPRs", "VGPR"), ("SGPRs Spill", "s-spill"), ("VGPRs Spill", "v-spill"), ("ScratchSize [bytes/lane]", "scratch"),
          ("Occupancy [waves/SIMD]", "occ"), ("LDS Size [bytes/block]", "LDS")]


def demangle(n):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            return subprocess.check_output([tool, n], text=True).strip()
        except Exception:
            continue
    return n


def main(files):
    print("%-92s %5s %5s %8s %8s %8s %4s %7s" % (("kernel",) + tuple(f[1] for f in FIELDS)))
    for f in files:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-mcode-object-version=5",
               "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", "/dev/null"] + sys.argv[2:]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        rows = {}
        for ln in err.splitlines():
            m = re.search(r"remark: [^:]*:\d+:\d+: (.*) \[-Rpass-analysis", ln) or re.search(r"remark: (.*) \[-Rpass-analysis", ln)
            if not m:
                continue
            t = m.group(1).strip()
            if t.startswith("Function Name:"):
                cur = t.split(":", 1)[1].strip()
                rows[cur] = {}
            elif cur and ":" in t:
                k, v = t.rsplit(":", 1)
                rows[cur][k.strip()] = v.strip()
        print("# " + f)
        for name, r in rows.items():
            d = demangle(name).replace("(anonymous namespace)::", "").replace("void ", "")
            d = re.sub(r"\(.*\)$", "", d)
            print("%-92s %5s %5s %8s %8s %8s %4s %7s" % ((d[:92],) + tuple(r.get(k, "?") for k, _ in FIELDS)))


if __name__ == "__main__":
    which = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else sorted(x for x in os.listdir(CSRC) if x.endswith(".hip"))
    main(which)
