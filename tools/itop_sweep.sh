#!/bin/bash
# Occupancy / shape sweeps of the itop walk in ONE gpurun call (profiles/r02_experiments.md).
# RUNS = lines of "lib D threads blocks_per_CU variants": lib = std (the built libacx.so) or the suffix of an
# alternative build pyahocorasick_amd/libacx_<lib>.so (e.g. ACX_EXTRA_CFLAGS=-DACX_ITOP_WPE=8); D caps the
# implicit depth (ACX_ITOP_MAX_D); threads / blocks per CU go to ACX_ITOP_THREADS / ACX_ITOP_BPC.
# example: RUNS='std 9 1024 1 0,131072' TAG=x bash tools/itop_sweep.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; LOG=$OUT/${TAG:-wpe}.log; : > $LOG
cp pyahocorasick_amd/libacx.so /tmp/libacx_std.so
run() {  # lib D threads bpc variants
  if [ "$1" = std ]; then cp /tmp/libacx_std.so pyahocorasick_amd/libacx.so; else cp pyahocorasick_amd/libacx_$1.so pyahocorasick_amd/libacx.so; fi
  echo "== lib=$1 D=$2 threads=$3 bpc=$4" | tee -a $LOG
  ACX_ITOP_MAX_D=$2 ACX_ITOP_THREADS=$3 ACX_ITOP_BPC=$4 timeout 120 python tools/microbench.py --variants $5 --reps 5 2>&1 | grep -o '"variant": [0-9]*\|"matches": [0-9]*\|"walk": [0-9.]*\|"expand": [0-9.]*' | paste - - - - | tee -a $LOG
}
while read -r line; do [ -n "$line" ] && run $line; done <<< "$RUNS"
cp /tmp/libacx_std.so pyahocorasick_amd/libacx.so
