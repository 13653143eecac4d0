#!/bin/bash
# config 5 behind the headline (the driver's command without its CPU legs), by --long-depth: two runs each
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for D in 3 4 5 6; do for i in 1 2; do
python bench.py --configs c5_iter_long --cpu-sample-reads 0 --long-depth $D 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['configs']['c5_iter_long']; print('long-depth $D: c5', c['value'], c['ms_per_step'], 'headline', round(d['value'],1))"
done; done
