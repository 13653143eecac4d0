#!/bin/bash
# A development variant of libacx.so: acx_ppm_stream4.hip (or the file named by SRC) compiled again with extra flags, everything
# else from the release objects.    tools/build_variant.sh NAME -DACX_S4_EXP=2 ...    ->  build/variants/libacx_NAME.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
SRC=${SRC:-acx_ppm_stream4.hip}
python -c "from pyahocorasick_amd.build import build_libacx; build_libacx(verbose=False)"
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mcode-object-version=5 -Iinclude -Ipyahocorasick_amd/csrc "$@" \
    -c pyahocorasick_amd/csrc/$SRC -o build/variants/${NAME}_$SRC.o
OBJS=""
for o in build/obj/*.o; do
  case $o in *"/$SRC.o") OBJS="$OBJS build/variants/${NAME}_$SRC.o";; *) OBJS="$OBJS $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build/variants/libacx_$NAME.so $OBJS
echo build/variants/libacx_$NAME.so
