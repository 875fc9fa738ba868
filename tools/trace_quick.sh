#!/bin/bash
# rocprofv3 kernel trace of the headline's timed region only (via gpurun, from the repo root): per-kernel averages.
# Usage: tools/trace_quick.sh TAG
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/bench_results.db $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-150
rm -rf $OUT/trace
