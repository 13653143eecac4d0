#!/bin/bash
# ONE rocprofv3 --kernel-trace --stats of the driver's own command (without its CPU and host-to-host legs): kernel stats of the whole run and,
# per configuration, the union of its dominant kernel's launch spans over its timed region (tools/roofline_check.py export_all) — every
# configuration in the process, and behind the neighbours, in which the driver measures it.   tools/r6_trace_all.sh [TAG]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT; TAG=${1:-r6}
(cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_all_prof -o stats -- python $R/bench.py --no-e2e --cpu-sample-reads 0 --full-json $OUT/${TAG}_all_bench_full.json > $OUT/${TAG}_all_bench.json 2> $OUT/${TAG}_all_prof.err; echo "rocprof rc=$?")
DB=$(find $OUT/${TAG}_all_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB $OUT/${TAG}_all_kernel_stats "rocprofv3 --kernel-trace --stats -- python bench.py --no-e2e --cpu-sample-reads 0" > /dev/null 2>&1; head -16 $OUT/${TAG}_all_kernel_stats.md
python tools/roofline_check.py export_all $DB $OUT/${TAG}_all_bench.json $OUT/${TAG}
rm -rf $OUT/${TAG}_all_prof
python tools/roofline_check.py check $OUT/${TAG}_all_bench.json $OUT/${TAG}_c*_spans.json
