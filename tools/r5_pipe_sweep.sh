#!/bin/bash
# round 5: results in flight x scan streams for the headline configuration (the union of the scan kernel's spans covers 90 % of the
# timed region with 2 x 2: the host launches scan k + 2 only when gather k has completed).  usage: tools/r5_pipe_sweep.sh TAG "P,S" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; TAG=$1; shift
for PS in "$@"; do
  P=${PS%,*}; S=${PS#*,}
  python bench.py --configs none --no-e2e --cpu-sample-reads 0 --pipeline $P --scan-streams $S $EXTRA > $OUT/${TAG}_p${P}_s${S}.json 2> $OUT/${TAG}_p${P}_s${S}.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_p${P}_s${S}.json").read().strip().splitlines()[-1])
    print("P=$P S=$S value %.1f GB/s  ms_per_step %.4f  kernel events %.4f  alone %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_events"]["avg_ms"], d["roofline"]["kernel_alone_ms"]))
except Exception as e:
    print("P=$P S=$S failed:", e); print(open("$OUT/${TAG}_p${P}_s${S}.err").read()[-800:])
PY
done
