#!/bin/bash
# the default command N times on one box: config.all of every line (how much the six figures move from run to run)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for i in $(seq 1 ${1:-3}); do python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: v[0] for k, v in d['config']['all'].items()})"; done
