#!/usr/bin/env python3
"""The vector-memory waits of one kernel in an assembly listing (no GPU): every `s_waitcnt vmcnt(n)` with the instruction behind it and
the vector-memory instructions in front of it since the wait before.   python tools/isa_waits.py file.s <substring of the kernel's label>"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end + 1]
pend = []
for i, l in enumerate(body):
    t = l.strip()
    if re.match(r"(global|buffer|flat)_(load|store|atomic)", t): pend.append(t.split()[0].replace("global_", "") + ("@%d" % (i + 1)))
    if t.startswith("s_waitcnt") and "vmcnt" in t:
        nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
        print("%6d  %-26s then: %-50s since last wait: %s" % (i + 1, t, nxt[:50], " ".join(pend[-8:])))
        pend = []
for l in body[-80:] + src[end:end + 60]:
    if re.search(r"NumVgprs|ScratchSize|sgpr_spill_count|vgpr_spill_count|Occupancy", l): print(l.strip())
