#!/bin/bash
# round 4: do bench.py's HIP events and rocprofv3 agree on the scan kernel's launch duration — on one scan stream, on two?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
for S in 1 2; do
  cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/r4_clk$S -o st -- python $R/bench.py --configs none --cpu-sample-reads 0 --no-e2e --scan-streams $S > $OUT/r4_clk${S}_bench.json 2> $OUT/r4_clk$S.err; cd $R
  python tools/rocpd_summary.py $(find $OUT/r4_clk$S -name "*.db" | head -1) $OUT/r4_clk${S}_stats "rocprofv3 --kernel-trace --stats -- python bench.py --configs none --cpu-sample-reads 0 --no-e2e --scan-streams $S" > /dev/null 2>&1; rm -rf $OUT/r4_clk$S
  python3 - <<PY
import json
d=json.loads(open("$OUT/r4_clk${S}_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
row=[l for l in open("$OUT/r4_clk${S}_stats.md") if "k_ppm_stream4" in l][0].split("|")
print("scan streams $S: value %.1f GB/s, step %.4f ms | HIP events in the timed region %.4f ms, pre-pass (alone) %.4f ms | rocprofv3: %s launches, average %s us" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"], r["kernel_alone_ms"], row[2].strip(), row[4].strip()))
PY
done
