#!/bin/bash
# config 5 measured the two ways: alone in its process, and behind the headline in the default command's process (what the driver's line carries)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for i in 1 2; do
python bench.py --configs none --mode iter_long --cpu-sample-reads 0 --no-e2e $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('alone  ', d['value'], d['ms_per_step'])"
python bench.py --configs c5_iter_long --cpu-sample-reads 0 --no-e2e $EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['configs']['c5_iter_long']; print('in line', c['value'], c['ms_per_step'], 'headline', d['value'])"
done
