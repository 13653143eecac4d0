#!/bin/bash
# results in flight x scan streams (bench.py --pipeline P --scan-streams S): the headline, c2_offsets and config 5 alone in their processes
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for PS in "3 3" "4 4" "5 5" "6 6"; do set -- $PS
  for W in "" "--workload c2o" "--mode iter_long"; do
    python bench.py --configs none $W --cpu-sample-reads 0 --no-e2e --pipeline $1 --scan-streams $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('P $1 S $2 [$W]', round(d['value'],1), d['ms_per_step'])"
  done
done
