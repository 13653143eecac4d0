#!/bin/bash
# Where a wave of k_ppm_stream4 spends a trip of its loop: one library build per segment between two clock points (-DACX_S4_TSEG=16a+b,
# acx_ppm_stream4.hip), ONE pair of s_memtime reads per trip each, so that every build runs at about the kernel's own speed.
#   tools/r6_phase_profile.sh build            (here: build/variants/libacx_tseg_<a>_<b>.so)
#   tools/r6_phase_profile.sh run [TAG]        (GPU box: gpurun_out/<TAG>_phase_profile.txt)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
SEGS=${SEGS:-"0_1 1_2 2_3 3_4 4_5 5_6 6_7 7_8 8_9 9_10 0_10"}
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -w -mcode-object-version=5 -Iinclude -Ipyahocorasick_amd/csrc"
if [ "$1" == "build" ]; then
  set -e
  python -c "from pyahocorasick_amd.build import build_libacx; build_libacx(verbose=False)"
  mkdir -p build/variants
  /opt/rocm/bin/hipcc $FL -DACX_TUNING -c pyahocorasick_amd/csrc/acx_capi.hip -o build/variants/tuning_acx_capi.hip.o &
  for S in $SEGS; do
    a=${S%_*}; b=${S#*_}
    /opt/rocm/bin/hipcc $FL -DACX_TUNING -DACX_S4_TSEG=$((16 * a + b)) $EXTRA -c pyahocorasick_amd/csrc/acx_ppm_stream4.hip -o build/variants/tseg_${S}_acx_ppm_stream4.hip.o &
    if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
  done
  wait
  for S in $SEGS; do
    OBJS=""
    for o in build/obj/*.o; do
      b=$(basename $o .o)
      if [[ "$b" == "acx_ppm_stream4.hip" ]]; then OBJS="$OBJS build/variants/tseg_${S}_acx_ppm_stream4.hip.o";
      elif [[ "$b" == "acx_capi.hip" ]]; then OBJS="$OBJS build/variants/tuning_acx_capi.hip.o"; else OBJS="$OBJS $o"; fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=pyahocorasick_amd/csrc/libacx.map -o build/variants/libacx_tseg_$S.so $OBJS
  done
  ls build/variants/libacx_tseg_*.so
else
  TAG=${2:-r6}; OUT=gpurun_out/${TAG}_phase_profile.txt; mkdir -p gpurun_out; : > $OUT
  for S in $SEGS; do
    ACX_PPM_PHASES=1 python tools/microbench.py --reps 5 --variants 0 $MICRO_ARGS --lib build/variants/libacx_tseg_$S.so 2> /tmp/ph_$S.err | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('seg $S kernel_ms', d['ms']['walk'])" >> $OUT
    grep "phases of the previous scan" /tmp/ph_$S.err | tail -1 | awk -v s=$S '{ for (i = 1; i <= NF; i++) if ($i ~ /^[0-9]+$/) { v[++n] = $i } ; printf("seg %s ticks %s trips %s waves %s  -> %.3f us per trip\n", s, v[1], v[2], v[8], v[1] / v[2] / 100.0) }' >> $OUT
  done
  cat $OUT
fi
