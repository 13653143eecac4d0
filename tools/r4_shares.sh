#!/bin/bash
# round 4: unequal runs for the waves of a block (acx_ppm_slot_first_tile): parity, then the scan kernel alone with the end
# times of its waves for several splits (per mille of tiles-per-wave: a,b -> +a +b -b -a for the four ages of a SIMD's waves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; OUT=$(pwd)/gpurun_out; mkdir -p $OUT; TAG=${1:-r4sh}
timeout 300 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_fuzz.py -m gpu -x -q -k "three_way or stream4 or fixed" 2>&1 | tail -2
for S in ${SHARES:-0,0 110,40 160,55 220,75}; do
  ACX_S4_SHARE=$S timeout 100 python tools/microbench.py --reps 5 --lib build/variants/libacx_wavetime.so > $OUT/${TAG}_$S.log 2>&1
  echo "share $S: $(tail -1 $OUT/${TAG}_$S.log | cut -c1-200)"; grep -A1 "wave times" $OUT/${TAG}_$S.log | tail -2 | cut -c1-330
done
