#!/bin/bash
# One gpurun call: for every named BASELINE config a bench line (with cpu_baseline), the rocprofv3 kernel stats of the
# same command and the PMC passes that profiles/traffic.json is made of.  Results under gpurun_out/<TAG>_<cfg>_*;
# copy what is to be judged into profiles/.   usage: tools/measure_all.sh TAG [cfg ...]   cfg in: c2 c5 c3 c4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r3}; shift
CFGS=${@:-c2 c5 c3 c4}
for CFG in $CFGS; do
  case $CFG in
    c2) BARGS="--steps 20 --warmup 4"; MARGS=""; KEY=c2_iter; KERNEL=k_ppm_stream;;
    c5) BARGS="--mode iter_long --steps 20 --warmup 4 --cpu-sample-reads 200000"; MARGS="--mode iter_long"; KEY=c2_iter_long; KERNEL=k_walk_long_sel;;
    c3) BARGS="--workload c3 --steps 12 --warmup 4"; MARGS="--alphabet text --bytes 536870912"; KEY=c3_iter; KERNEL=k_ppm_stream;;
    c4) BARGS="--workload c4 --steps 12 --warmup 4"; MARGS="--alphabet snort --keys 1000000 --bytes 536870912"; KEY=c4_iter; KERNEL=k_ppm_stream;;
  esac
  T=${TAG}_${CFG}
  echo "== $CFG: pmc"
  for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    N=$(echo $C | tr ' ' '_')
    cd /tmp && timeout -k 5 240 rocprofv3 --pmc $C --output-format csv -d $OUT/${T}_pmc_$N -o pmc -- python $R/tools/microbench.py --variants 0 --reps 3 $MARGS > $OUT/${T}_pmc_$N.log 2>&1; echo "pmc $N rc=$?"
    cd $R
  done
  python tools/pmc_summary.py $OUT ${T} > $OUT/${T}_pmc_summary.json 2> $OUT/${T}_pmc_summary.err
  python tools/make_traffic.py $OUT/${T}_pmc_summary.json $KEY $KERNEL > $OUT/${T}_traffic_entry.json 2>&1; tail -3 $OUT/${T}_traffic_entry.json
  find $OUT -name "*.csv" -size +4M -delete
  echo "== $CFG: bench"
  timeout 1200 python bench.py $BARGS > $OUT/${T}_bench.json 2> $OUT/${T}_bench.err; echo "bench rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("$OUT/${T}_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["kernel"], round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], "cpu", d.get("cpu_baseline", {}).get("value"), d.get("setup"))
except Exception as e:
    print("no bench line:", e); print(open("$OUT/${T}_bench.err").read()[-1500:])
PY
  echo "== $CFG: rocprofv3 --kernel-trace --stats"
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${T}_prof -o stats -- python $R/bench.py $BARGS --cpu-sample-reads 0 --no-e2e > $OUT/${T}_prof_bench.json 2> $OUT/${T}_prof.err; echo "rocprof rc=$?"
  cd $R
  python tools/rocpd_summary.py $(find $OUT/${T}_prof -name "*.db" | head -1) $OUT/${T}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS --cpu-sample-reads 0 --no-e2e" > /dev/null 2>&1 && head -8 $OUT/${T}_kernel_stats.md
  rm -rf $OUT/${T}_prof
done
cp profiles/traffic.json $OUT/${TAG}_traffic.json
echo "== done"
