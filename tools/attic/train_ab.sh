#!/bin/bash
# GPU: the bf16 step time under a few switches (same box). Usage: tools/train_ab.sh "ENV=.. ENV=.." "..."   (TESTS=1: training tests first)
[ "${TESTS:-0}" = 1 ] && python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -3
for envs in "$@"; do
  for rep in 1 2; do
    v=$(env $envs python tools/train_bench.py --bz 128 --steps 10 --precision bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  loss', d['loss_last'][:2])")
    echo "bf16 [$envs] -> $v"
  done
done
