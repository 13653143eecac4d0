#!/bin/bash
# the scan kernel alone on the other shapes (text = config 3, snort = config 4, offsets = c2_offsets), for a list of library builds
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in "$@"; do
  for MA in "--alphabet text" "--alphabet snort --keys 1000000" "--layout offsets"; do
    python tools/microbench.py --reps 5 --variants 0 --check 200 $MA --lib $L 2>/dev/null | tail -1 | LIBNAME="$L $MA" python -c '
import json, os, sys
d = json.loads(sys.stdin.read()); print(os.environ["LIBNAME"], "ms", d.get("ms"), "min", d.get("min_walk_ms"), "matches", d.get("matches"), "sample_ok", d.get("sample_ok_vs_oracle"))'
  done
done
