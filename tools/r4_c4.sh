#!/bin/bash
# round 4, config 4 (a million signatures, ragged packets): one line of the pair table per two positions (default) against
# one probe of the 3-gram bitmap per position (variant bit 18); parity on the 256-symbol three-way tests first
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=$(pwd)/gpurun_out; mkdir -p $OUT; TAG=${1:-r4c4}
timeout 600 python -m pytest tests/test_gpu_ppm.py -x -q -k "bytes256 or high4 or text27 or hex16" > $OUT/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/${TAG}_pytest.log
timeout 900 python tools/microbench.py --alphabet snort --keys 1000000 --bytes 536870912 --variants ${VARIANTS:-0,262144,0,262144} --check 400 --reps 5 > $OUT/${TAG}_micro.log 2>&1; echo "micro rc=$?"
tail -5 $OUT/${TAG}_micro.log
