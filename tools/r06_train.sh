#!/bin/bash
# via gpurun: the fused feed-forward epilogue of the bf16 training step: bit-identity test, training tests, step time with the
# fusion on / off (development build), then the product build. Usage: tools/r06_train.sh TAG
TAG=${1:-r06f}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_train.py -m gpu -q --maxfail=5 -x -s > $O/pytest_train.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest_train.log
grep -E "passed|failed|error|^FAILED|^ERROR|^E  " $O/pytest_train.log | tail -20
for rep in 1 2; do
  for fuse in 1 0; do
    v=$(RPR_DEV_LIB=1 RPR_TRAIN_FUSE_FF=$fuse python tools/train_bench.py --bz 128 --steps 8 --precision bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', d['backward_kernel_ms'])")
    echo "fuse=$fuse rep$rep: $v"
  done
done | tee $O/train_fuse_ab.txt
python tools/train_bench.py --bz 128 --steps 8 --precision bf16 2>/dev/null | tail -1 > $O/train_bench_product.json; python -c "import json; d=json.load(open('$O/train_bench_product.json')); print('product', d['ms_per_step'])"
