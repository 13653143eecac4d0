#!/bin/bash
# round 4: SQ counters of k_ppm_stream4 and k_ppm_stream (variant bit 19) from ONE process per pass
# usage: tools/r4_pmc.sh TAG "CTR1 CTR2" "CTR3" ...   (env VARIANTS, MICRO_ARGS)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
i=0
for C in "$@"; do
  i=$((i+1))
  cd /tmp && timeout -k 5 60 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_s$i -o pmc -- python $R/tools/microbench.py --reps 3 --variants ${VARIANTS:-0,524288} ${MICRO_ARGS} > $OUT/${TAG}_pmc_s$i.log 2>&1; echo "pmc [$C] rc=$?"
  cd $R
done
python tools/pmc_summary.py $OUT ${TAG} > $OUT/${TAG}_pmc_summary.json 2> $OUT/${TAG}_pmc_summary.err
find $OUT -name "*.csv" -size +4M -delete
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_pmc_summary.json"))
for k,v in d.items():
    if 'ppm_stream' in k or 'walk' in k: print(k, json.dumps({a:round(b,1) for a,b in v.items()}))
PY
