#!/bin/bash
# via gpurun: the forced tail at 128-dim heads (t5-3b): tests, fuzz, then the opt-in bench leg. Usage: tools/r06_3b.sh TAG
TAG=${1:-r06g}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_edges.py tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -q --maxfail=5 -k "128_dim or g7_3b or randomised_parity" > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR|^E  " $O/pytest.log | tail -20
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary t5_3b > $O/bench_3b.json 2> $O/bench_3b.log
echo "bench rc=$?"; python - <<PY
import json
d = json.loads(open("$O/bench_3b.json").read().strip().splitlines()[-1])
t = d["secondary"]["t5_3b"]
print({k: t[k] for k in ("value", "ms_per_step", "valid_leaves", "kernel_ms_by_class") if k in t}); print("roof", t.get("roofline", {}).get("frac"))
PY
