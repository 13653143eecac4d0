"""Timing experiment (results may be WRONG on rare paths): drop chosen `s_waitcnt vmcnt` lines from k_ppm_stream4<false>'s loop.
   python tools/s4_patch_waits.py in.s out.s      with S4_DROP="M_PUSH:1,M_FILTER:2"  (mark name : which occurrence of the mark, 1-based,
   inside the <false> kernel; the FIRST vmcnt wait behind that mark is dropped) or "M_REC:1:-1" (the LAST vmcnt wait of that region)"""
import os, re, sys
src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if l.startswith("_ZN12_GLOBAL__N_113k_ppm_stream4ILb0ELb0EEEv12acx_ppm_args:"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
marks = []   # (line, name)
for i in range(start, end):
    m = re.search(r"; MARK (\w+)", src[i])
    if m: marks.append((i, m.group(1)))
marks.append((end, "END"))
for spec in os.environ.get("S4_DROP", "").split(","):
    if not spec: continue
    parts = spec.split(":")
    name, occ = parts[0], int(parts[1]); which = int(parts[2]) if len(parts) > 2 else 0
    idx = [k for k, (_, n) in enumerate(marks) if n == name][occ - 1]
    lo, hi = marks[idx][0], marks[idx + 1][0]
    waits = [i for i in range(lo, hi) if "s_waitcnt" in src[i] and "vmcnt" in src[i] and not src[i].lstrip().startswith(";")]
    i = waits[which]
    print("dropping line %d (%s): %s" % (i + 1, spec, src[i].strip()), file=sys.stderr)
    rest = re.sub(r"vmcnt\(\d+\)\s*", "", src[i].strip())
    src[i] = ("\t" + rest if "cnt(" in rest else "\t; dropped: " + src[i].strip())
open(sys.argv[2], "w").write("\n".join(src))
