#!/bin/bash
# dev build only: time and count the instructions of k_ppm_stream with phases switched off (ACX_PPM_DBG)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r3d}
for D in ${DBGS:-0 1 2 3 4 12 28}; do
  export ACX_PPM_DBG=$D
  python tools/microbench.py --variants 0 --reps 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$D', 'walk_ms', d['ms']['walk'], 'matches', d['matches'])"
  cd /tmp && timeout -k 5 60 rocprofv3 --pmc ${PMCS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} --output-format csv -d $OUT/${TAG}_pmc_d$D -o pmc -- python $R/tools/microbench.py --reps 2 --variants 0 > /dev/null 2>&1
  cd $R
  python - <<PY
import csv,glob
acc={}
for p in glob.glob("$OUT/${TAG}_pmc_d$D/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if 'k_ppm_stream' in r['Kernel_Name']:
            acc.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
tiles=150000000/2048
print('   per tile:', {k: round(sum(v)/len(v)/tiles,1) for k,v in acc.items()})
PY
done
