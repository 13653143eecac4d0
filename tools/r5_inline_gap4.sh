#!/bin/bash
# fourth leg: scan streams x side streams (pool size n = build/variants/libacx_side{n}.so)
TAG=${1:-gap4}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_inline_gap.txt
: > $O
pick='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
print("   %.1f GB/s step %.4f ms  frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))'
run() { echo "== $*" >> $O; "$@" 2>/dev/null | python -c "$pick" >> $O 2>&1; }
H="python bench.py --configs none --cpu-sample-reads 0"
L="python bench.py --mode iter_long --configs none --cpu-sample-reads 0"
run $H --lib build/variants/libacx_side1.so --scan-streams 3 --pipeline 3
run $H --lib build/variants/libacx_side1.so --scan-streams 2 --pipeline 3
run $H --lib build/variants/libacx_side1.so --scan-streams 3 --pipeline 4
run $H --lib build/variants/libacx_side1.so --scan-streams 4 --pipeline 4
run $H --lib build/variants/libacx_side2.so --scan-streams 2 --pipeline 3
run $H --lib build/variants/libacx_side1.so --scan-streams 3 --pipeline 3
run $L --lib build/variants/libacx_side2.so --scan-streams 2 --pipeline 3
run $L --lib build/variants/libacx_side2.so --scan-streams 2 --pipeline 4
run $L --lib build/variants/libacx_side3.so --scan-streams 2 --pipeline 3
run $L --lib build/variants/libacx_side3.so --scan-streams 3 --pipeline 3
run $L --lib build/variants/libacx_side1.so --scan-streams 2 --pipeline 3
run $L --lib build/variants/libacx_side3.so --scan-streams 1 --pipeline 3
cat $O
