#!/bin/bash
# via gpurun: new-kernel parity, then the GPU suite, then the small-batch bench legs
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05f}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wave_split or linear_kernel or repeatable" > $O/pytest_wsplit.log 2>&1; echo "wsplit rc=$?"; tail -3 $O/pytest_wsplit.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --secondary latency,small_batch > $O/bench_small.json 2> $O/bench_small.log; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench_small.json").read().strip().splitlines()[-1])
print("headline", round(d["value"], 1), "q/s", round(d["ms_per_step"], 1), "ms")
s = d["secondary"]
print("latency", {k: round(v["value"], 2) for k, v in s["latency"].items() if isinstance(v, dict)})
for k, v in s["small_batch"].items():
    if isinstance(v, dict):
        print(k, round(v["value"], 1), "q/s", round(v["ms_per_search"], 2), "ms", "launches", v["launches_per_search"], "hbm frac", round(v["roofline"]["frac"], 3), "ft", round(v["roofline"]["frac_forced_tail"], 3), "mfma", round(v["roofline"]["mfma_frac_forced_tail"], 3))
PY
