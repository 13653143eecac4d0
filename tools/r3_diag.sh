#!/bin/bash
# round-3 diagnostics: instruction rates, gather rates, and where the waves of k_ppm_stream spend their cycles
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r3a}
timeout 300 tools/valu_rate.bin > $OUT/${TAG}_valu_rate.txt 2>&1; echo "valu rc=$?"
timeout 300 python tools/microbench.py --variants 0 --reps 7 > $OUT/${TAG}_micro.log 2>&1; tail -2 $OUT/${TAG}_micro.log
bash tools/pmc_small.sh ${TAG} "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
   "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
   "SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE SQ_INSTS_LDS"
cat $OUT/${TAG}_pmc_summary.json | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if 'ppm_stream' in k: print(json.dumps(v,indent=1))
"
