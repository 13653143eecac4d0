#!/bin/bash
# round 4: the scan kernel alone for a list of development variants (build/variants/libacx_<NAME>.so; "release" = the package's library)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
TAG=$1; shift
for V in "$@"; do
  if [ "$V" == "release" ]; then L=""; else L="--lib build/variants/libacx_$V.so"; fi
  timeout 120 python tools/microbench.py --variants ${VARIANTS:-0} --reps ${REPS:-7} $L ${MICRO_ARGS} > $OUT/${TAG}_$V.log 2>&1
  echo "$V rc=$? $(tail -1 $OUT/${TAG}_$V.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms"]["walk"], d["min_walk_ms"], d["matches"], d.get("sample_ok_vs_oracle"))' 2>/dev/null)"
done
echo "== done"
