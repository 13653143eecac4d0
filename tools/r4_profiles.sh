#!/bin/bash
# round 4: the artefacts behind the config-2 roofline entry — rocprofv3 kernel stats of the bench command, PMC passes
# (separate runs, counters only) on the scan kernel alone.   tools/r4_profiles.sh TAG
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; T=${1:-r4}_c2
BARGS="--configs none --cpu-sample-reads 0 --no-e2e"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${T}_prof -o stats -- python $R/bench.py $BARGS > $OUT/${T}_prof_bench.json 2> $OUT/${T}_prof.err; cd $R
python tools/rocpd_summary.py $(find $OUT/${T}_prof -name "*.db" | head -1) $OUT/${T}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS" > /dev/null 2>&1; rm -rf $OUT/${T}_prof
head -9 $OUT/${T}_kernel_stats.md | tail -4
VARIANTS=0 bash tools/r4_pmc.sh ${T} "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
   "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
   "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" | tail -3
