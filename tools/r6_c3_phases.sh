#!/bin/bash
# Config 3 (100 k text keys, 512 MiB as one haystack): k_ppm_stream<8,8,false,true,..>'s time with parts switched off (a -DACX_PPM_DEV -DACX_PPM_DEV_C3 -DACX_TUNING
# build of acx_ppm_kernels.hip + acx_capi.hip: build/variants/libacx_c3dev.so; ACX_PPM_DBG 32: no deeper walks, 4: no rounds at all, 16: no filter probes)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; R=$(pwd)
for DBG in ${DBGS:-0 32 4 16 20}; do
  ACX_PPM_DBG=$DBG python tools/microbench.py --variants 0 --reps 5 --alphabet text --bytes 536870912 --lib $R/build/variants/libacx_c3dev.so 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg $DBG', d['ms'], d['matches'])"
done
