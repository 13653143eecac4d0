#!/bin/bash
# dev build: microbench walk time for a list of ACX_PPM_DBG values
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for D in "$@"; do
  ACX_PPM_DBG=$D python tools/microbench.py --variants 0 --reps 7 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dbg=$D', 'walk_ms', d['ms']['walk'], 'min', d['min_walk_ms'], 'matches', d['matches'])"
done
