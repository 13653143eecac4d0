#!/bin/bash
# c2_offsets: k_ppm_stream4's offsets form against the general stream kernel's (variant bit 19), the scan kernel's and the whole step's times
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for V in 0 524288; do
python bench.py --configs none --workload c2o --cpu-sample-reads 0 --no-e2e --verbose --variant $V 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('variant $V', 'GB/s', round(d['value'],1), 'ms/step', d['ms_per_step'], 'kernel', r.get('kernel'), 'kernel_ms', r.get('kernel_ms'), {k: v for k, v in d.items() if 'ms' in k and k != 'ms_per_step'})"
done
