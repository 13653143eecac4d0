#!/bin/bash
# Why is c5 (iter_long) 5 % slower inside the default line than alone, and the headline 1 % slower with the pack installed?
# usage (GPU box): tools/r5_inline_gap.sh TAG      -> gpurun_out/TAG_inline_gap.txt
TAG=${1:-gap}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_inline_gap.txt
: > $O
pick='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
c=d.get("configs",{}).get("c5_iter_long")
print("   headline(%s) %.1f GB/s step %.4f ms%s" % (d["roofline"]["kernel"], d["value"], d["ms_per_step"], "   c5 in-line %.1f GB/s" % c["value"] if c and "value" in c else (" c5: %s" % c if c else "")))'
run() { echo "== $*" >> $O; "$@" 2>/dev/null | python -c "$pick" >> $O 2>&1; }
run python bench.py --mode iter_long --configs none --cpu-sample-reads 0
run python bench.py --configs c5_iter_long --cpu-sample-reads 0
run env ACX_BENCH_KEEP_HEADLINE=1 python bench.py --configs c5_iter_long --cpu-sample-reads 0
run env ACX_BENCH_KEEP_HEADLINE=1 python bench.py --configs c5_iter_long
run env GPU_MAX_HW_QUEUES=8 python bench.py --configs c5_iter_long --cpu-sample-reads 0
run env GPU_MAX_HW_QUEUES=8 python bench.py --mode iter_long --configs none --cpu-sample-reads 0
run python bench.py --configs none --cpu-sample-reads 0
run env GPU_MAX_HW_QUEUES=8 python bench.py --configs none --cpu-sample-reads 0
cat $O
