#!/bin/bash
# A development variant of libacx.so: the files named by SRC (default acx_ppm_stream4.hip) compiled again with extra flags,
# everything else from the release objects.    [SRC="a.hip b.hip"] tools/build_variant.sh NAME -DACX_S4_EXP=2 ...
#   ->  build/variants/libacx_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=${SRC:-acx_ppm_stream4.hip}
python -c "from pyahocorasick_amd.build import build_libacx; build_libacx(verbose=False)"
mkdir -p build/variants
for S in $SRC; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -w -mcode-object-version=5 -Iinclude -Ipyahocorasick_amd/csrc "$@" \
      -c pyahocorasick_amd/csrc/$S -o build/variants/${NAME}_$S.o
done
OBJS=""
for o in build/obj/*.o; do
  b=$(basename $o .o)
  if [[ " $SRC " == *" $b "* ]]; then OBJS="$OBJS build/variants/${NAME}_$b.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=pyahocorasick_amd/csrc/libacx.map -o build/variants/libacx_$NAME.so $OBJS
echo build/variants/libacx_$NAME.so
