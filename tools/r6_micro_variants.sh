#!/bin/bash
# tools/microbench.py (config 2 shape, scan kernel alone) for a list of library builds, with a sample check against the oracle:
#   tools/r6_micro_variants.sh "microbench args" lib.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; MA=$1; shift
for L in "$@"; do
  python tools/microbench.py --reps 9 --variants 0 --check 2000 $MA --lib $L 2>/dev/null | tail -1 | LIBNAME="$L $MA" python -c '
import json, os, sys
d = json.loads(sys.stdin.read()); print(os.environ["LIBNAME"], "ms", d.get("ms"), "min", d.get("min_walk_ms"), "matches", d.get("matches"), "sample_ok", d.get("sample_ok_vs_oracle"))'
done
