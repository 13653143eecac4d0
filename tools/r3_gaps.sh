#!/bin/bash
# kernel timeline of the bench loop (config 2): where a step's time goes besides the scan kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; T=${1:-r3u}
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OUT/${T}_trace -o tr -- python $R/bench.py --steps 20 --warmup 4 --cpu-sample-reads 0 > $OUT/${T}_trace_bench.json 2> $OUT/${T}_trace.err; echo "rc=$?"
cd $R
python tools/gap_report.py $(find $OUT/${T}_trace -name "*.db" | head -1) 70 > $OUT/${T}_gaps.txt 2>&1
rm -rf $OUT/${T}_trace
tail -75 $OUT/${T}_gaps.txt
