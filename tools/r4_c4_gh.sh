#!/bin/bash
# round 4, config 4: a larger hashed copy of the global filter beside tiles of 1024 positions (122 KiB, 58 % of random codes
# pass) against the release (96 KiB beside tiles of 2048: 67 %), and tiles of 1024 with the 96 KiB copy (what the tile size alone does)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=$(pwd)/gpurun_out; mkdir -p $OUT; TAG=${1:-r4c4gh}
M="tools/microbench.py --alphabet snort --keys 1000000 --bytes 536870912 --variants 0,0 --check 400 --reps 5"
timeout 300 python $M > $OUT/${TAG}_release.log 2>&1; echo "release rc=$?"; tail -3 $OUT/${TAG}_release.log
timeout 300 python $M --lib build/variants/libacx_gh122.so > $OUT/${TAG}_gh122.log 2>&1; echo "gh122 rc=$?"; tail -3 $OUT/${TAG}_gh122.log
ACX_PPM_NSUB=4 timeout 300 python $M --lib build/variants/libacx_tuning.so > $OUT/${TAG}_nsub4.log 2>&1; echo "nsub4 rc=$?"; tail -3 $OUT/${TAG}_nsub4.log
