#!/usr/bin/env python3
"""Basic blocks of k_ppm_stream4<false>'s loop by phase mark, with instruction counts by class (no GPU).
   tools/s4_waits.sh -DACX_S4_MARK && python tools/s4_isa_blocks.py [build/isa/loop0.s]  -> one line per block:
   phase, label, VALU / SALU / LDS / VMEM / branch / wait+nop counts, and where its terminators go."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "build/isa/loop0.s"
lines = open(path).read().split("\n")
def cls(op):
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep")): return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")): return "br"
    if op.startswith(("ds_",)): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith(("s_load", "s_memtime", "s_memrealtime", "s_store")): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
phase, nmark = "prologue", 0
blocks = []     # (phase, label, counts, targets, first_line)
cur = None
def start(label, i):
    global cur
    cur = {"phase": phase, "label": label, "n": {}, "to": [], "line": i + 1}
    blocks.append(cur)
start("entry", 0)
for i, l in enumerate(lines):
    t = l.strip()
    m = re.search(r"; MARK (\w+)", t)
    if m:
        nmark += 1; phase = "%s#%d" % (m.group(1), nmark)
        start("(mark)", i); continue
    if re.match(r"^\.LBB\d+_\d+:", t):
        start(t.split(":")[0], i); continue
    if t.startswith("; %bb."):
        start(t.split()[1], i); continue
    if not t or t.startswith((";", ".", "_Z")): continue
    op = t.split()[0]
    c = cls(op)
    cur["n"][c] = cur["n"].get(c, 0) + 1
    if c == "br" and len(t.split()) > 1: cur["to"].append(op.replace("s_cbranch_", "").replace("s_branch", "jmp") + ">" + t.split()[1])
for b in blocks:
    n = b["n"]
    tot = sum(n.values())
    if tot == 0 and not b["to"]: continue
    print("%-12s %-10s L%-5d tot %3d  valu %3d salu %3d lds %2d vmem %2d smem %2d br %2d wait %2d  %s" % (
        b["phase"], b["label"], b["line"], tot, n.get("valu", 0), n.get("salu", 0), n.get("lds", 0), n.get("vmem", 0), n.get("smem", 0), n.get("br", 0), n.get("wait", 0), " ".join(b["to"])))
