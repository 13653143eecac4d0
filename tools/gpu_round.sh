#!/bin/bash
# One gpurun call: any of  pytest smoke micro bench prof pmc  (env STAGES, default all).
# Everything is logged under gpurun_out/<TAG>_* (merged back by gpurun).  Never aborts early.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r01}
STAGES=${STAGES:-"hw pytest smoke micro bench prof pmc"}
has() { [[ " $STAGES " == *" $1 "* ]]; }

if has hw; then
  (rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock|gfx|Cacheline|L2:|L3:" | head -20; nproc; free -g | head -2) > $OUT/${TAG}_hw.txt 2>&1
fi
if has pytest; then
  echo "== pytest gpu"
  timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest.log
  tail -4 $OUT/${TAG}_pytest.log
fi
if has smoke; then
  echo "== smoke"
  timeout 300 python __graft_entry__.py --smoke > $OUT/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/${TAG}_smoke.log
  tail -2 $OUT/${TAG}_smoke.log
fi
if has micro; then
  echo "== microbench"
  timeout 600 python tools/microbench.py --variants ${VARIANTS:-0} ${MICRO_ARGS} > $OUT/${TAG}_micro.log 2>&1; echo "micro rc=$?" | tee -a $OUT/${TAG}_micro.log
  tail -16 $OUT/${TAG}_micro.log
fi
if has micro2; then
  timeout 300 python tools/microbench.py --mode iter_long --variants ${VARIANTS_LONG:-0} --reps 5 > $OUT/${TAG}_micro_long.log 2>&1; tail -4 $OUT/${TAG}_micro_long.log
  timeout 300 python tools/microbench.py --alphabet alnum --variants ${VARIANTS_ALNUM:-0} --reps 5 > $OUT/${TAG}_micro_alnum.log 2>&1; tail -4 $OUT/${TAG}_micro_alnum.log
fi
if has bench; then
  echo "== bench (torch path)"
  timeout 900 python bench.py ${BENCH_ARGS:---steps 20 --warmup 3} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench rc=$?" | tee -a $OUT/${TAG}_bench.err
  cat $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
fi
if has prof; then
  echo "== rocprofv3 --kernel-trace --stats (same command as bench)"
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o stats -- python $R/bench.py ${BENCH_ARGS:---steps 20 --warmup 3} --cpu-sample-reads 0 > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err; echo "rocprof rc=$?"
  cd $R
  python tools/rocpd_summary.py $(find $OUT/${TAG}_prof -name "*.db" | head -1) $OUT/${TAG}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py ${BENCH_ARGS:---steps 20 --warmup 3} --cpu-sample-reads 0" && cat $OUT/${TAG}_kernel_stats.md | tail -8
  cat $OUT/${TAG}_prof_bench.json
fi
if has pmc; then
  echo "== rocprofv3 --pmc passes (counters only; separate runs)"
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    N=$(echo $C | tr ' ' '_')
    cd /tmp && timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$N -o pmc -- python $R/tools/microbench.py --variants ${PMC_VARIANT:-0} --reps 3 ${MICRO_ARGS} > $OUT/${TAG}_pmc_$N.log 2>&1; echo "pmc $N rc=$?"
    cd $R
  done
  python tools/pmc_summary.py $OUT ${TAG} > $OUT/${TAG}_pmc_summary.json 2> $OUT/${TAG}_pmc_summary.err; cat $OUT/${TAG}_pmc_summary.json; tail -3 $OUT/${TAG}_pmc_summary.err
  find $OUT -name "*.csv" -size +4M -delete
fi
echo "== done"
