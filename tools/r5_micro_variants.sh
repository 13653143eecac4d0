#!/bin/bash
# tools/microbench.py for a list of library builds: tools/r5_micro_variants.sh "microbench args" lib.so ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; MA=$1; shift
for L in "$@"; do
  python tools/microbench.py --reps 7 --variants 0 $MA --lib $L 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['ms'], 'matches', d['matches'])"
done
