#!/bin/bash
# One gpurun call: correctness, smoke, bench, microbench variants, rocprof stats.
# Everything is logged under gpurun_out/ (merged back by gpurun).  Never aborts early.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
TAG=${1:-r01a}
echo "== hw" | tee $OUT/${TAG}_hw.txt
(rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx|Cacheline|L2:|L3:" | head -20; nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | head -8) >> $OUT/${TAG}_hw.txt 2>&1
echo "== pytest gpu"
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
tail -5 $OUT/${TAG}_pytest.log
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/${TAG}_smoke.log
tail -3 $OUT/${TAG}_smoke.log
echo "== microbench"
timeout 600 python tools/microbench.py --variants ${VARIANTS:-0,1,256,257,64,128,16,32} > $OUT/${TAG}_micro.log 2>&1; echo "micro rc=$?" | tee -a $OUT/${TAG}_micro.log
cat $OUT/${TAG}_micro.log | tail -12
timeout 300 python tools/microbench.py --mode iter_long --variants 0 --reps 5 > $OUT/${TAG}_micro_long.log 2>&1; tail -2 $OUT/${TAG}_micro_long.log
timeout 300 python tools/microbench.py --alphabet alnum --variants 0,1 --reps 5 > $OUT/${TAG}_micro_alnum.log 2>&1; tail -3 $OUT/${TAG}_micro_alnum.log
echo "== bench (torch path)"
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" | tee -a $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench.json; tail -5 $OUT/${TAG}_bench.err
echo "== rocprof kernel stats"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/${TAG}_prof -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-sample-reads 0 > $GRAFT_REPO_ROOT/$OUT/${TAG}_prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/${TAG}_prof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find $OUT/${TAG}_prof -name "*stats*" | head; for f in $(find $OUT/${TAG}_prof -name "*kernel_stats*.csv" | head -2); do head -12 $f; done
# keep the merge-back small: drop the raw per-dispatch trace if it is huge
find $OUT/${TAG}_prof -name "*.csv" -size +8M -delete
echo "== done"
