#!/bin/bash
# Same-box A/B of the encoder attention kernels on the headline configuration, then single-query latency with the step
# cross-attention on either kernel (via gpurun, from the repo root). Usage: tools/attn_ab2.sh TAG
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cfg in "0" "2" "0" "2"; do
  RPR_STEP_CROSS_MFMA=$cfg timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --secondary "" \
    > $OUT/ab_enc$cfg.json 2>> $OUT/ab2.log
  python - <<PY
import json
d = json.loads(open("$OUT/ab_enc$cfg.json").read().strip().splitlines()[-1])
print("step-cross-mfma $cfg:", round(d["value"], 1), "q/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 1) for k, v in d.get("kernel_breakdown_lanes_ms", {}).items()})
PY
done
for cfg in "0" "2"; do
  RPR_STEP_CROSS_MFMA=$cfg timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "latency" \
    > $OUT/ab_lat$cfg.json 2>> $OUT/ab2.log
  python - <<PY
import json
d = json.loads(open("$OUT/ab_lat$cfg.json").read().strip().splitlines()[-1])
print("step-cross-mfma $cfg latency:", {k: (round(v["value"], 3) if isinstance(v, dict) and "value" in v else v) for k, v in d["secondary"]["latency"].items()})
PY
done
