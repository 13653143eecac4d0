#!/bin/bash
# iter_long A/B in one gpurun call: select form (0), branchy form (bit 22), each with and without table rows in LDS (bit 21)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python tools/microbench.py --mode iter_long --variants 0,4194304,2097152,6291456 --reps 9 --check 20000 > gpurun_out/${1:-r3t}_long_ab.jsonl 2> gpurun_out/${1:-r3t}_long_ab.err
tail -5 gpurun_out/${1:-r3t}_long_ab.jsonl | cut -c1-600
