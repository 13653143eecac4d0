#!/bin/bash
# round 5: what every roofline entry of the driver's bench line is recomputed from (tools/roofline_check.py), in ONE gpurun call:
#   * for each single-GPU configuration of the default line, `rocprofv3 --kernel-trace --stats` of bench.py running THAT
#     configuration alone -> <TAG>_<cfg>_kernel_stats.{md,csv} (the per-kernel averages) and <TAG>_<cfg>_spans.json (the union of
#     the dominant kernel's launch spans per launch over the timed region: with two scan streams launches overlap);
#   * optionally (PMC=1) four counter passes per configuration on tools/microbench.py (counters only, separate runs) ->
#     <TAG>_<cfg>_pmc_summary.json; CALIB=1: the same counters on tools/fetch_calib.bin (known byte counts).
# usage: [PMC=1] [CALIB=1] [NOSTATS=1] tools/r5_profiles.sh TAG [cfg ...]     cfg in: c2 c5 c2o c2k c3 c4   (NOSTATS: the counter passes only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
TAG=${1:-r5}; shift
CFGS=${@:-c2 c5 c2o c2k c3 c4}
CTRS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum")
if [ -n "$CALIB" ]; then
  [ -x tools/fetch_calib.bin ] || hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib.bin
  tools/fetch_calib.bin > $OUT/${TAG}_calib_known.jsonl; cat $OUT/${TAG}_calib_known.jsonl
  i=0
  for C in "${CTRS[@]}" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    (cd /tmp && timeout -k 5 120 rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_calib_pmc_s$i -o pmc -- $R/tools/fetch_calib.bin > $OUT/${TAG}_calib_pmc_s$i.log 2>&1; echo "calib pmc [$C] rc=$?")
  done
  python tools/calib_summary.py $OUT ${TAG}_calib $OUT/${TAG}_calib_known.jsonl > $OUT/${TAG}_fetch_calibration.json; cat $OUT/${TAG}_fetch_calibration.json
fi
for CFG in $CFGS; do
  case $CFG in
    c2)  BARGS="--configs none"; MARGS=""; NAME=headline; KEY=c2_iter; KERNEL=k_ppm_stream4; HB=150000000;;
    c5)  BARGS="--configs none --mode iter_long"; MARGS="--mode iter_long"; NAME=c5_iter_long; KEY=c2_iter_long; KERNEL=k_ppm_stream4; HB=150000000;;
    c2o) BARGS="--configs none --workload c2o"; MARGS=""; NAME=c2_offsets;;
    c2k) BARGS="--configs none --workload c2k"; MARGS=""; NAME=c2_long_keys;;
    c3)  BARGS="--configs none --workload c3 --steps 12"; MARGS="--alphabet text --bytes 536870912"; NAME=c3; KEY=c3_iter; KERNEL=k_ppm_stream; HB=536870912;;
    c4)  BARGS="--configs none --workload c4 --steps 12"; MARGS="--alphabet snort --keys 1000000 --bytes 536870912"; NAME=c4; KEY=c4_iter; KERNEL=k_ppm_stream; HB=536870912;;
  esac
  T=${TAG}_${CFG}
  if [ -z "$NOSTATS" ]; then
  echo "== $CFG: rocprofv3 --kernel-trace --stats -- python bench.py $BARGS --cpu-sample-reads 0 --no-e2e"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/${T}_prof -o stats -- python $R/bench.py $BARGS --cpu-sample-reads 0 --no-e2e > $OUT/${T}_prof_bench.json 2> $OUT/${T}_prof.err; echo "rocprof rc=$?")
  DB=$(find $OUT/${T}_prof -name "*.db" | head -1)
  python tools/rocpd_summary.py $DB $OUT/${T}_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py $BARGS --cpu-sample-reads 0 --no-e2e" > /dev/null 2>&1 && head -9 $OUT/${T}_kernel_stats.md | tail -4
  python tools/roofline_check.py export $DB $OUT/${T}_prof_bench.json $OUT/${T}_spans.json $NAME
  rm -rf $OUT/${T}_prof
  fi
  if [ -n "$PMC" ] && [ "$CFG" != "c2o" ] && [ "$CFG" != "c2k" ]; then
    i=0
    for C in "${CTRS[@]}"; do
      i=$((i+1))
      (cd /tmp && timeout -k 5 240 rocprofv3 --pmc $C --output-format csv -d $OUT/${T}_pmc_s$i -o pmc -- python $R/tools/microbench.py --variants 0 --reps 3 $MARGS > $OUT/${T}_pmc_s$i.log 2>&1; echo "pmc [$C] rc=$?")
    done
    python tools/pmc_summary.py $OUT ${T} > $OUT/${T}_pmc_summary.json 2> $OUT/${T}_pmc_summary.err
    python tools/make_traffic.py $OUT/${T}_pmc_summary.json $KEY $KERNEL "" $HB > $OUT/${T}_traffic_entry.json 2>&1; tail -4 $OUT/${T}_traffic_entry.json
    find $OUT -name "*.csv" -size +4M -delete
  fi
done
cp profiles/traffic.json $OUT/${TAG}_traffic.json
echo "== done"
