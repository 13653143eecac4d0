#!/bin/bash
# Config 4 (a million signatures, 512 MiB of packets): which structure the fabric reads above the haystack's bytes belong to.
# FETCH_SIZE / TCC hits of k_ppm_stream<8,4,..> in a development build with parts switched off (ACX_PPM_DBG): 32 no deeper walks, 4 no rounds
# (no hot cells either), 16 no filter probes.     tools/r6_c4_traffic.sh [TAG]   (needs build/variants/libacx_c4dev.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; TAG=${1:-r6}; mkdir -p $OUT
MARGS="--alphabet snort --keys 1000000 --bytes 536870912 --lib $R/build/variants/libacx_c4dev.so"
for DBG in 0 32 4; do
  for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    N=$(echo $C | tr ' ' '_')
    (cd /tmp && ACX_PPM_DBG=$DBG timeout -k 5 300 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}c4d${DBG}_pmc_$N -o pmc -- python $R/tools/microbench.py --variants 0 --reps 3 $MARGS > $OUT/${TAG}c4d${DBG}_$N.log 2>&1; echo "dbg $DBG pmc [$C] rc=$?")
  done
  python tools/pmc_summary.py $OUT ${TAG}c4d${DBG} > $OUT/${TAG}_c4_dbg${DBG}_pmc_summary.json 2>/dev/null
  ACX_PPM_DBG=$DBG python tools/microbench.py --variants 0 --reps 5 $MARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg $DBG', d['ms'], d['matches'])"
  python - <<PY
import json
d = json.load(open("$OUT/${TAG}_c4_dbg${DBG}_pmc_summary.json")).get("k_ppm_stream", {})
print("dbg $DBG", {k: (round(v / 1e6, 1) if isinstance(v, float) and v > 1e4 else v) for k, v in d.items()})
PY
  rm -rf $OUT/${TAG}c4d${DBG}_pmc_* 
done 2>&1 | tee $OUT/${TAG}_c4_traffic_attribution.txt
