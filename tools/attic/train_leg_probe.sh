#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-trainleg}; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { local tag=$1; shift; timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>$O/$tag.log | tail -1 > $O/$tag.json; python - <<PY
import json
d = json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
s = d["secondary"]
print("$tag", round(d["value"]), {k: round(v["ms_per_step"], 2) for k, v in s.items() if isinstance(v, dict) and "ms_per_step" in v})
PY
}
run full_a --secondary train
run full_b --secondary train
run noexact --no-exact-fp32 --secondary train
run noroof --no-exact-fp32 --no-roofline --secondary train
run nolanes --no-exact-fp32 --no-roofline --no-lanes --secondary train
