#!/bin/bash
# One GPU-box round trip (via gpurun, from the repo root): GPU test-suite, smoke, GEMM micro-bench, short bench line.
# Usage: tools/gpu_round.sh TAG [pytest-args...]
set -u
TAG=${1:-rXX}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 -s "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^\[parity\]|^\[select\]|passed|failed|error" $OUT/pytest.log | tail -40
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"
timeout 300 python tools/gemm_bench.py 21760 > $OUT/gemm_bench.log 2>&1; cat $OUT/gemm_bench.log | tail -8
timeout 900 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc=$?"
tail -5 $OUT/bench.log; python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step", "value_pcie_inclusive", "saturated") if k in d})
    print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us", "launches_per_step", "traffic")})
    print("breakdown", d.get("kernel_breakdown_ms"))
    print("exact_fp32", d.get("exact_fp32")); print("cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
PY
