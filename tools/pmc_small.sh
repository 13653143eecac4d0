#!/bin/bash
# A few small rocprofv3 --pmc passes (<= 3 counters each, 45 s cap per pass).
# usage: tools/pmc_small.sh TAG "CTR1 CTR2" "CTR3" ... [-- microbench args]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=$1; shift
GROUPS_=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
[ "$1" == "--" ] && shift
i=0
for C in "${GROUPS_[@]}"; do
  i=$((i+1))
  cd /tmp && timeout -k 5 45 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_s$i -o pmc -- python $R/tools/microbench.py --reps 3 --variants 0 "$@" > $OUT/${TAG}_pmc_s$i.log 2>&1; echo "pmc [$C] rc=$?"
  cd $R
done
python tools/pmc_summary.py $OUT ${TAG} > $OUT/${TAG}_pmc_summary.json 2> $OUT/${TAG}_pmc_summary.err
find $OUT -name "*.csv" -size +4M -delete
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_pmc_summary.json"))
for k,v in d.items():
    if 'walk' in k: print(k, {a:b for a,b in v.items()})
PY
