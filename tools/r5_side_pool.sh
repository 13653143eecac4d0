#!/bin/bash
# the side streams as a pool of the device (iter: stream 0; iter_long: round robin over three): the default line + the GPU tests that
# exercise asynchronous scans            usage (GPU box): tools/r5_side_pool.sh TAG
TAG=${1:-pool}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'P'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
print("headline %.1f GB/s step %.4f frac %.4f step.frac %.4f | %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step"]["frac"],
      "  ".join("%s %.1f" % (k, v.get("value", -1)) for k, v in d.get("configs", {}).items())))
P
timeout 900 python -m pytest tests/test_gpu_long.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
tail -2 gpurun_out/${TAG}_tests.txt
