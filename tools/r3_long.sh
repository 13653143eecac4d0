#!/bin/bash
# iter_long A/B in one gpurun call (variant bits: 22 the branchy form, 21 no table rows in LDS, 12 non-temporal haystack loads and event stores)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/microbench.py --mode iter_long --variants ${2:-0,4194304,2097152,4096} --reps 9 --check 20000 > gpurun_out/${1:-r3t}_long_ab.jsonl 2> gpurun_out/${1:-r3t}_long_ab.err
tail -5 gpurun_out/${1:-r3t}_long_ab.jsonl | cut -c1-400
