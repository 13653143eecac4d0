#!/bin/bash
# round 5: iter_long (config 5) — the bench line for a list of "P,S" (results in flight, scan streams)   tools/r5_long.sh TAG "P,S" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; TAG=$1; shift
for PS in "$@"; do
  P=${PS%,*}; S=${PS#*,}
  python bench.py --configs none --mode iter_long --no-e2e --cpu-sample-reads 0 --pipeline $P --scan-streams $S $EXTRA > $OUT/${TAG}_p${P}_s${S}.json 2> $OUT/${TAG}_p${P}_s${S}.err
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_p${P}_s${S}.json").read().strip().splitlines()[-1])
    print("iter_long P=$P S=$S value %.1f GB/s  ms_per_step %.4f  sync %.4f  kernels %s" % (d["value"], d["ms_per_step"], d["ms_per_step_synchronous"], d["roofline"]["pipeline"]["kernel_ms"]))
except Exception as e:
    print("P=$P S=$S failed:", e); print(open("$OUT/${TAG}_p${P}_s${S}.err").read()[-1500:])
PY
done
