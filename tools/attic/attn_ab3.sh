#!/bin/bash
# Same-box A/B of the step cross-attention kernels on the legs with more than 16 beams (config 4: t5-large beam 100; beam 1000
# at batch 1) via gpurun, from the repo root. Usage: tools/attn_ab3.sh TAG
set -u
TAG=${1:-rXX}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cfg in "0" "2" "0" "2"; do
  RPR_STEP_CROSS_MFMA=$cfg timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "latency,config4,f2" \
    > $OUT/ab3_step$cfg.json 2>> $OUT/ab3.log
  python - <<PY
import json
d = json.loads(open("$OUT/ab3_step$cfg.json").read().strip().splitlines()[-1])
s = d["secondary"]
print("step-cross-mfma $cfg:", "config4", round(s["config4"]["value"], 2), "q/s; latency", round(s["latency"]["beams10"]["value"], 3), round(s["latency"]["beams1000"]["value"], 3), "ms; f2", {k: round(v["value"], 1) for k, v in s["f2"].items() if isinstance(v, dict)})
PY
done
