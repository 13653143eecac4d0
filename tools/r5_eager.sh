#!/bin/bash
# the pool's streams made and used at the first request (-DACX_SIDE_EAGER variant) against made on first use (release), alternating
mkdir -p gpurun_out; O=gpurun_out/${1:-eager}_eager.txt; : > $O
pick='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
print("headline %.1f  c5 in-line %.1f" % (d["value"], d["configs"]["c5_iter_long"]["value"]))'
for i in 1 2 3; do
  echo -n "release: " >> $O; python bench.py --configs c5_iter_long --cpu-sample-reads 0 2>/dev/null | python -c "$pick" >> $O
  echo -n "eager:   " >> $O; python bench.py --configs c5_iter_long --cpu-sample-reads 0 --lib build/variants/libacx_eager.so 2>/dev/null | python -c "$pick" >> $O
done
cat $O
