#!/bin/bash
# second leg of tools/r5_inline_gap.sh: the side streams come from a pool now — does keeping the headline's scanners alive still cost c5?
TAG=${1:-gap2}
mkdir -p gpurun_out
O=gpurun_out/${TAG}_inline_gap.txt
: > $O
pick='import json,sys
l=[x for x in sys.stdin.read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
c=d.get("configs",{}).get("c5_iter_long")
print("   headline(%s) %.1f GB/s step %.4f ms%s" % (d["roofline"]["kernel"], d["value"], d["ms_per_step"], "   c5 in-line %.1f GB/s" % c["value"] if c and "value" in c else (" c5: %s" % c if c else "")))'
run() { echo "== $*" >> $O; "$@" 2>/dev/null | python -c "$pick" >> $O 2>&1; }
run env ACX_BENCH_KEEP_HEADLINE=1 python bench.py --configs c5_iter_long --cpu-sample-reads 0
run python bench.py --configs c5_iter_long --cpu-sample-reads 0
run env ACX_BENCH_KEEP_HEADLINE=1 python bench.py --configs c5_iter_long --cpu-sample-reads 0
run python bench.py --mode iter_long --configs none --cpu-sample-reads 0
timeout 600 python -m pytest tests/test_gpu_long.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 >> $O
cat $O
