#!/bin/bash
# rocprofv3 kernel stats of the config-2 bench command (tools/measure_all.sh does this for every config)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; T=${1:-r3H}_c2
BARGS="--steps 20 --warmup 4"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${T}_prof -o stats -- python $R/bench.py $BARGS --cpu-sample-reads 0 --no-e2e > $OUT/${T}_prof_bench.json 2> $OUT/${T}_prof.err; cd $R
python tools/rocpd_summary.py $(find $OUT/${T}_prof -name "*.db" | head -1) $OUT/${T}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS --cpu-sample-reads 0 --no-e2e" > /dev/null 2>&1; rm -rf $OUT/${T}_prof
head -9 $OUT/${T}_kernel_stats.md | tail -3
