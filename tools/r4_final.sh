#!/bin/bash
# round 4, last call: the bench line, kernel stats of the same command, PMC passes for every bench line's dominant kernel on the
# final sources, smoke, the suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
C="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum|TCC_EA0_RDREQ_sum|SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
IFS='|' read -ra GRP <<< "$C"
(time python bench.py) > $OUT/r4_bench.json 2> $OUT/r4_bench.err; echo "bench rc=$?"
BARGS="--configs none --cpu-sample-reads 0 --no-e2e"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/r4_c2_prof -o stats -- python $R/bench.py $BARGS > $OUT/r4_c2_prof_bench.json 2> $OUT/r4_c2_prof.err; cd $R
python tools/rocpd_summary.py $(find $OUT/r4_c2_prof -name "*.db" | head -1) $OUT/r4_c2_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS" > /dev/null 2>&1; rm -rf $OUT/r4_c2_prof
head -9 $OUT/r4_c2_kernel_stats.md | tail -3 | cut -c1-160
VARIANTS=0 bash tools/r4_pmc.sh r4_c2 "${GRP[@]}" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" | tail -1
MICRO_ARGS="--alphabet text --keys 100000 --bytes 536870912" VARIANTS=0 bash tools/r4_pmc.sh r4_c3 "${GRP[@]}" | tail -1
MICRO_ARGS="--mode iter_long" VARIANTS=0 bash tools/r4_pmc.sh r4_c5 "${GRP[@]}" | tail -2
MICRO_ARGS="--alphabet snort --keys 1000000 --bytes 536870912" VARIANTS=0 bash tools/r4_pmc.sh r4_c4 "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" | tail -1
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q > $OUT/r4_pytest.log 2>&1; grep -E "passed|failed" $OUT/r4_pytest.log | tail -1
