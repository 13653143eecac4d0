#!/bin/bash
# round 4: k_ppm_stream4 against k_ppm_stream (variant bit 19) — parity on the three-way tests, then the scan kernel alone
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out; mkdir -p $OUT
TAG=${1:-r4a}
timeout 600 python -m pytest tests/test_gpu_ppm.py -x -q -k "alphabets or fixed_stride or runs_of_tiles or he_her" > $OUT/${TAG}_pytest_ppm.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/${TAG}_pytest_ppm.log
tail -5 $OUT/${TAG}_pytest_ppm.log
timeout 600 python tools/microbench.py --variants ${VARIANTS:-0,524288,0,524288} --check 3000 --reps 9 > $OUT/${TAG}_micro.log 2>&1; echo "micro rc=$?" | tee -a $OUT/${TAG}_micro.log
tail -8 $OUT/${TAG}_micro.log
echo "== done"
