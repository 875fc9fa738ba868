#!/bin/bash
# via gpurun: the bf16 step's products one by one under different kernel routes (development build). Usage: tools/r06_gemm_bf16.sh TAG
TAG=${1:-r06_gemm_bf16}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
{
for v in "RPR_NOP=1" "RPR_GEMM_DEEP=400" "RPR_BF16_PP=90" "RPR_BF16_PP=0" $EXTRA; do
  echo "== $v"
  env RPR_DEV_LIB=1 $v timeout 300 python tools/gemm_bf16_bench.py 2>&1 | tail -12
done
} | tee $O/gemm_bf16.txt
