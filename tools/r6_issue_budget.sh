#!/bin/bash
# The issue budget of k_ppm_stream4: dynamic wave-instruction counts by class for the release kernel and for
# timing-only builds with phases switched off (-DACX_S4_EXP: 2 no deeper walks, 6 no walks + no record stores, 8 no rounds, 24 no push),
# from the SQ_INSTS_* counters — the phases' counts are the differences.    tools/r6_issue_budget.sh build | run [TAG]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
EXPS="0 2 6 8 24"
if [ "$1" == "build" ]; then
  for E in $EXPS; do tools/build_variant.sh bud$E -DACX_S4_EXP=$E > /dev/null & done; wait; ls build/variants/libacx_bud*.so
else
  TAG=${2:-r6}; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
  (cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ') > $OUT/${TAG}_sq_counter_names.txt
  for E in $EXPS; do
    tools/pmc_small.sh ${TAG}bud$E "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES" -- --lib $R/build/variants/libacx_bud$E.so > /dev/null 2>&1
    python - <<PY
import json
d = json.load(open("$OUT/${TAG}bud${E}_pmc_summary.json"))
k = d.get("k_ppm_stream4", {})
print("EXP=%-3s" % "$E", json.dumps({a: int(b) for a, b in sorted(k.items()) if a.startswith("SQ_")}))
PY
  done | tee $OUT/${TAG}_issue_budget_counters.txt
  rm -rf $OUT/${TAG}bud*_pmc_s*
fi
