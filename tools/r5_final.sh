#!/bin/bash
# the final default line of the round (and the smoke entry)       usage (GPU box): tools/r5_final.sh TAG
TAG=${1:-final}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - gpurun_out/${TAG}_bench.json <<'P'
import json,sys
l=[x for x in open(sys.argv[1]).read().splitlines() if x.startswith("{")][-1]; d=json.loads(l)
print("headline %.1f GB/s step %.4f frac %.4f step.frac %.4f | %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step"]["frac"],
      "  ".join("%s %.1f" % (k, v.get("value", -1)) for k, v in d.get("configs", {}).items())))
P
timeout 100 python -m pytest tests/test_gpu_ppm.py tests/test_gpu_ws.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
