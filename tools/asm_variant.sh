#!/bin/bash
# A development variant of libacx.so whose k_ppm_stream4 device code comes from PATCHED assembly (timing experiments only):
#   tools/asm_variant.sh NAME patch.py [extra hipcc flags]   -> build/variants/libacx_NAME.so
# patch.py is run as: python patch.py in.s out.s
set -e
cd "$(dirname "$0")/.."
NAME=$1; PATCH=$2; shift; shift
SRC=pyahocorasick_amd/csrc/acx_ppm_stream4.hip
LLVM=/opt/rocm/lib/llvm/bin
W=build/variants/asm_$NAME; mkdir -p $W
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fvisibility-inlines-hidden -w -mcode-object-version=5 -Iinclude -Ipyahocorasick_amd/csrc"
python -c "from pyahocorasick_amd.build import build_libacx; build_libacx(verbose=False)"
/opt/rocm/bin/hipcc $FL "$@" --cuda-device-only -S $SRC -o $W/dev.s
python $PATCH $W/dev.s $W/dev_patched.s
$LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -mcode-object-version=5 -c $W/dev_patched.s -o $W/dev.o
$LLVM/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $W/dev.out $W/dev.o
$LLVM/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.out -output=$W/dev.hipfb
/opt/rocm/bin/hipcc $FL "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $SRC -o $W/acx_ppm_stream4.hip.o
OBJS=""
for o in build/obj/*.o; do
  b=$(basename $o .o)
  if [[ "$b" == "acx_ppm_stream4.hip" ]]; then OBJS="$OBJS $W/acx_ppm_stream4.hip.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -Wl,--version-script=pyahocorasick_amd/csrc/libacx.map -o build/variants/libacx_$NAME.so $OBJS
echo build/variants/libacx_$NAME.so
