#!/bin/bash
# GPU: training tests, then the step time in every precision (and the earlier split-K weight-gradient route beside it)
export RPR_DEV_LIB=1   # the switches below are development switches: libripor_hip_dev.so (same sources, -DRPR_DEV_SWITCHES)
python -m pytest tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -4
for prec in bf16 f16x2; do
  for legacy in 0 1; do
    v=$(RPR_TRAIN_DW_SPLITK=$legacy python tools/train_bench.py --bz 128 --steps 8 --precision $prec 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d.get('ms_per_step', d.get('value',0)),2), d.get('unit',''))")
    echo "precision=$prec splitk_route=$legacy -> $v"
  done
done
