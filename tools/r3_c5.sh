#!/bin/bash
# the iter_long bench line and its rocprofv3 kernel stats (what tools/measure_all.sh does for c5, without the PMC passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; T=${1:-r3C}_c5
BARGS="--mode iter_long --steps 20 --warmup 4 --cpu-sample-reads 200000"
python bench.py $BARGS > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; echo "bench rc=$?"
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${T}_prof -o stats -- python $R/bench.py $BARGS --cpu-sample-reads 0 > $OUT/${T}_prof_bench.json 2> $OUT/${T}_prof.err; cd $R
python tools/rocpd_summary.py $(find $OUT/${T}_prof -name "*.db" | head -1) $OUT/${T}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS --cpu-sample-reads 0" > /dev/null 2>&1; rm -rf $OUT/${T}_prof
tail -c 600 $OUT/${T}_bench.json
