#!/bin/bash
# Same-box A/B of the tail / step attention generations on the headline configuration (via gpurun, from the repo root).
# Usage: tools/attn_ab.sh TAG
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --secondary ''"
for cfg in "1 0" "2 0" "2 1" "1 0" "2 1"; do
  set -- $cfg
  RPR_TAIL_ATTN_GEN=$1 RPR_STEP_CROSS_MFMA=$2 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --secondary "" \
    > $OUT/ab_gen$1_step$2.json 2>> $OUT/ab.log
  python - <<PY
import json
d = json.loads(open("$OUT/ab_gen$1_step$2.json").read().strip().splitlines()[-1])
print("gen $1 step-cross-mfma $2:", round(d["value"], 1), "q/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 1) for k, v in d.get("kernel_breakdown_lanes_ms", {}).items()})
PY
done
