#!/bin/bash
# third leg: how many side streams should the pool hold?  variants build/variants/libacx_side{1,2,3}.so (-DACX_SIDE_STREAMS=n), release = 4
TAG=${1:-gap3}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_inline_gap.txt
: > $O
pick='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
cs=d.get("configs",{})
print("   headline(%s) %.1f GB/s step %.4f ms   %s" % (d["roofline"]["kernel"], d["value"], d["ms_per_step"], "  ".join("%s %.1f" % (k, v.get("value", -1)) for k, v in cs.items())))'
run() { echo "== $*" >> $O; "$@" 2>/dev/null | python -c "$pick" >> $O 2>&1; }
for n in 1 2 3; do
  run python bench.py --configs c5_iter_long,c2_offsets --cpu-sample-reads 0 --lib build/variants/libacx_side$n.so
done
run python bench.py --configs c5_iter_long,c2_offsets --cpu-sample-reads 0
run python bench.py --mode iter_long --configs none --cpu-sample-reads 0 --lib build/variants/libacx_side1.so
run python bench.py --mode iter_long --configs none --cpu-sample-reads 0 --lib build/variants/libacx_side2.so
cat $O
