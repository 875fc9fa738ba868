#!/bin/bash
# Kernel sequences of one search at 1 / 8 / 64 queries in flight (beam 10) and beam 1000 at batch 1 (via gpurun):
# tools/small_batch_traces.sh TAG   -> gpurun_out/TAG_q{1,8,64}, TAG_b1000
TAG=${1:-sb}
for cfg in "1 10" "8 10" "64 10" "1 1000"; do
  set -- $cfg
  bash $GRAFT_REPO_ROOT/tools/latency_trace.sh ${TAG}_q$1_b$2 $1 $2 > /dev/null 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_q$1_b$2/last_search.txt
  head -c 400 $GRAFT_REPO_ROOT/gpurun_out/${TAG}_q$1_b$2/plain.json; echo
done
