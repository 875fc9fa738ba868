#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05l}; mkdir -p $O
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary ''"
one() { local tag=$1; shift; env "$@" bash -c "$B $EXTRA" 2>$O/$tag.log | tail -1 > $O/$tag.json; python - <<PY
import json
d = json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag", round(d["value"], 1), "q/s", round(d["ms_per_step"], 2), "ms", d["config"]["queries_per_step_per_gpu"])
PY
}
EXTRA=""
one band4_a RPR_GEMM_BAND=4
one band0_a RPR_GEMM_BAND=0
one band4_b RPR_GEMM_BAND=4
one band0_b RPR_GEMM_BAND=0
one band2 RPR_GEMM_BAND=2
one band8 RPR_GEMM_BAND=8
EXTRA="--batch 4300"
one q4300 RPR_GEMM_BAND=4
EXTRA="--batch 6450 --steps 2"
one q6450 RPR_GEMM_BAND=4
RPR_SELECT_CLOCK=1 timeout 300 python bench.py --batch 1 --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary '' 2>&1 | grep "select t=" | tail -8 > $O/select_clock_q1.txt
cat $O/select_clock_q1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -x > $O/pytest_band.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_band.log
