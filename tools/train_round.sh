#!/bin/bash
# GPU: training-step tests + the training bench in both GEMM precisions. Usage: bash tools/train_round.sh <tag>
tag=${1:-train}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x -s > $out/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "train-bwd|train\]|passed|failed|Error|assert" $out/pytest.log | tail -30
for p in f16x2 f32; do
  timeout 600 python tools/train_bench.py --bz 128 --precision $p > $out/train_bench_$p.json 2> $out/train_bench_$p.log; echo "bench $p rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$out/train_bench_$p.json")); print("$p", round(d["ms_per_step"],2), "ms", round(d["examples_per_s"],1), "ex/s", d["loss_first"], d["backward_kernel_ms"])
except Exception as e: print("no result", e); print(open("$out/train_bench_$p.log").read()[-2000:])
PY
done
