for i in 1 2 3; do
for V in 0 536870912; do
python bench.py --configs none --mode iter_long --cpu-sample-reads 0 --no-e2e --variant $V 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('variant $V', d['value'], d['ms_per_step'])"
done; done
