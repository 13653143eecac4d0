#!/bin/bash
# per-kernel durations of ONE batch scan at a time (tools/microbench.py: synchronous scans, nothing shares the chip): tools/r5_micro_stats.sh TAG [microbench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; TAG=$1; shift
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o stats -- python $R/tools/microbench.py --reps 5 --variants 0 "$@" > $OUT/${TAG}_micro.json 2> $OUT/${TAG}_micro.err; echo "rocprof rc=$?")
python tools/rocpd_summary.py $(find $OUT/${TAG}_prof -name "*.db" | head -1) $OUT/${TAG}_micro_kernel_stats "rocprofv3 --kernel-trace --stats -- python tools/microbench.py --reps 5 --variants 0 $*" > /dev/null 2>&1
rm -rf $OUT/${TAG}_prof
tail -1 $OUT/${TAG}_micro.json | cut -c1-400; head -16 $OUT/${TAG}_micro_kernel_stats.md | tail -11
