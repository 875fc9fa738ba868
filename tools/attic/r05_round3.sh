#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05k}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_edges.py::test_trie_level_tables_change_nothing tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_levels.log 2>&1; echo "levels rc=$?"; tail -3 $O/pytest_levels.log
for lv in 1 0; do
RPR_SELECT_LEVELS=$lv timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary latency,small_batch > $O/bench_lv$lv.json 2> $O/bench_lv$lv.log
python - <<PY
import json
d = json.loads(open("$O/bench_lv$lv.json").read().strip().splitlines()[-1])
s = d["secondary"]
print("levels=$lv headline", round(d["value"], 1), "latency", {k: round(v["value"], 3) for k, v in s["latency"].items() if isinstance(v, dict)},
      "small", {k: round(v["ms_per_search"], 3) for k, v in s["small_batch"].items() if isinstance(v, dict)})
PY
done
