#!/bin/bash
# tuning build (-DACX_TUNING): where the gather runs (side stream at low / default / high priority, or the caller's stream) x results in flight
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo -n "$1 pipeline=$2: "; env $1 python bench.py --steps 30 --warmup 4 --cpu-sample-reads 0 --pipeline $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'GB/s', round(d['ms_per_step'],4), 'ms; kernel', d['roofline']['kernel_avg_ms'])"; }
for P in 2 3; do
  run "ACX_X=0" $P
  run "ACX_SIDE_DEFAULT_PRIORITY=1" $P
  run "ACX_SIDE_HIGH_PRIORITY=1" $P
done
