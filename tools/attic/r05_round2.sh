#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05g}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_fullsize.py::test_mid_size_trie_against_the_kv_cached_oracle tests/test_gpu_attn_generations.py -m gpu -q -x -s --durations=8 > $O/pytest_new.log 2>&1; echo "new rc=$?"; grep -E "^\[heavy|^\[mid|passed|failed|Error" $O/pytest_new.log | tail -12
timeout 2000 python -m pytest tests/test_gpu_cli.py -m gpu -q -x -s -k "eight_rank" --durations=3 > $O/pytest_8rank.log 2>&1; echo "8rank rc=$?"; grep -E "^\[bench x8|passed|failed|Error|slowest|s call" $O/pytest_8rank.log | tail -8
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 --deselect tests/test_gpu_cli.py::test_eight_rank_rehearsal_of_the_scaling_bench_and_the_cli > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary heavy_tail,latency > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("headline", round(d["value"], 1), "q/s", round(d["ms_per_step"], 1), "ms", "saturated", d["saturated"], "exact_fp32", d.get("exact_fp32", {}).get("value"))
print("roofline frac", round(d["roofline"]["frac"], 4), "avg_launch_us", round(d["roofline"]["avg_launch_us"], 1))
print("heavy_tail", d["secondary"]["heavy_tail"])
print("latency", {k: round(v["value"], 2) for k, v in d["secondary"]["latency"].items() if isinstance(v, dict)})
PY
