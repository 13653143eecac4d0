#!/bin/bash
# the last dispatches of a bench run as rocprofv3 saw them (start, duration, idle time before, queue): tools/r5_timeline.sh TAG [bench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; TAG=$1; shift
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o stats -- python $R/bench.py --configs none --cpu-sample-reads 0 --no-e2e "$@" > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err; echo "rocprof rc=$?")
python tools/gap_report.py $(find $OUT/${TAG}_prof -name "*.db" | head -1) ${NLAST:-60} > $OUT/${TAG}_timeline.txt
rm -rf $OUT/${TAG}_prof
tail -${NLAST:-60} $OUT/${TAG}_timeline.txt
