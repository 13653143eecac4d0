#!/bin/bash
# round 4: bench.py (config 2, the driver's command) for a list of library builds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
TAG=$1; shift
for V in "$@"; do
  if [ "$V" == "release" ]; then L=""; else L="--lib build/variants/libacx_$V.so"; fi
  timeout 300 python bench.py --steps 40 --warmup 4 --cpu-sample-reads 0 --no-e2e $L ${BENCH_ARGS} > $OUT/${TAG}_bench_$V.json 2> $OUT/${TAG}_bench_$V.err
  echo "$V rc=$? $(python -c 'import sys,json; d=json.load(open(sys.argv[1])); print("GB/s %.1f ms/step %.4f kernel %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_avg_ms"]))' $OUT/${TAG}_bench_$V.json 2>/dev/null)"
done
echo "== done"
