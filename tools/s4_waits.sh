#!/bin/bash
# The vector-memory waits of k_ppm_stream4<false>'s main loop, by phase mark (no GPU): which s_waitcnt vmcnt() the compiler put where.
#   tools/s4_waits.sh [extra hipcc flags]      -> build/isa/s4_mark.s, build/isa/loop0.s
cd "$(dirname "$0")/.."
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -mcode-object-version=5 -Iinclude -Ipyahocorasick_amd/csrc -DACX_S4_MARK "$@" \
    --cuda-device-only -S pyahocorasick_amd/csrc/acx_ppm_stream4.hip -o build/isa/s4_mark.s || exit 1
python3 - <<'PY'
import os, re
src = open("build/isa/s4_mark.s").read().split("\n")
# the kernel <false>: from its label to its s_endpgm
start = next(i for i, l in enumerate(src) if l.startswith("_ZN12_GLOBAL__N_113k_ppm_stream4ILb" + os.environ.get("KERN", "0") + "ELb" + os.environ.get("OFFS", "0") + "EEEv12acx_ppm_args:"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end + 1]
open("build/isa/loop0.s", "w").write("\n".join(body))
phase = "prologue"; nmark = 0
for i, l in enumerate(body):
    m = re.search(r"; MARK (\w+)", l)
    if m:
        nmark += 1; phase = m.group(1) + "#%d" % nmark
    if "s_waitcnt" in l and "vmcnt" in l:
        nxt = next(b.strip() for b in body[i + 1:] if b.strip() and not b.strip().startswith(";"))
        print("%5d  %-14s %-28s then: %s" % (i + 1, phase, l.strip(), nxt))
for l in body:
    if re.search(r"\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count)", l): print(l.strip())
PY
grep -A30 "k_ppm_stream4ILb${KERN:-0}ELb${OFFS:-0}EEEv12acx_ppm_args$" build/isa/s4_mark.s | grep -E "NumVgprs|NumSgprs|ScratchSize|Occupancy|spill" | head
