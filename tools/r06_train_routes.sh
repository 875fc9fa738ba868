#!/bin/bash
# via gpurun: which kernel should carry the bf16 step's N = 768 products (8192 / 4096 rows, K = 768 ... 3072)? Development build,
# alternating environments on one box. Usage: tools/r06_train_routes.sh TAG
TAG=${1:-r06_train_routes}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {  # label, env...
  local label=$1; shift
  v=$(env RPR_DEV_LIB=1 "$@" python tools/train_bench.py --bz 128 --steps 8 --precision bf16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', d['backward_kernel_ms'])")
  echo "$label: $v"
}
for rep in 1 2; do
  run "default rep$rep" RPR_NOP=1
  run "deep 4-stage to 400 tiles rep$rep" RPR_GEMM_DEEP=400
  run "pp from 96 tiles rep$rep" RPR_BF16_PP=96
  run "pp never rep$rep" RPR_BF16_PP=0
  for t in $EXTRA_TILES; do run "bf16 mid tile $t rep$rep" RPR_BF16_MID_TILE=$t; done
done 2>&1 | tee $O/train_routes.txt
