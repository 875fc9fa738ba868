#!/bin/bash
# GPU box: per-site GEMM profile of the current build + timing ablations of the fused-RMSNorm epilogue pieces.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-abl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32"
for v in 0 1 2 3; do
  RPR_DEBUG_FUSED=$v timeout 600 $B > $OUT/bench_dbg$v.json 2> $OUT/bench_dbg$v.log
  python - <<PY
import json
d = json.loads(open("$OUT/bench_dbg$v.json").read().strip().splitlines()[-1])
print("RPR_DEBUG_FUSED=$v", round(d["value"], 1), "q/s", d["kernel_breakdown_ms"])
PY
done
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B --no-roofline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/bench_results.db $OUT/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/gemm_sites.py $OUT/trace/bench_results.db $OUT/gemm_sites.json | tee $OUT/gemm_sites.txt
rm -rf $OUT/trace
