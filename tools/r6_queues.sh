#!/bin/bash
# How much of the headline depends on which hardware queues the scan and side streams get: the default command's
# headline alone, then with N streams made and used BEFORE the library's first scan (what RCCL's and a caller's own streams do), with the
# HIP runtime's default of four hardware queues, and with both.   tools/r6_queues.sh [TAG]  -> gpurun_out/<TAG>_queues.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=${1:-r6}; OUT=gpurun_out/${TAG}_queues.txt; mkdir -p gpurun_out; : > $OUT
run() { # label, env, args
  local L=$1; shift; local E=$1; shift
  env $E python bench.py --configs none --cpu-sample-reads 0 --no-e2e "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s %7.1f GB/s  step %.4f ms  kernel alone %.4f  queues %s' % ('$L', d['value'], d['ms_per_step'], d['roofline']['kernel_alone_ms'], json.dumps(d['config']['queues'])))" >> $OUT
}
run "default (GPU_MAX_HW_QUEUES=8)" "A=1"
run "8 streams made first" "A=1" --dummy-streams 8
run "16 streams made first" "A=1" --dummy-streams 16
run "GPU_MAX_HW_QUEUES=4 (the runtime's default)" "GPU_MAX_HW_QUEUES=4"
run "GPU_MAX_HW_QUEUES=4, 8 streams made first" "GPU_MAX_HW_QUEUES=4" --dummy-streams 8
run "default again" "A=1"
cat $OUT
