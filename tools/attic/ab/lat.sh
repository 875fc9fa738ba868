#!/bin/bash
# same-box A/B at small batches: tools/ab/lat.sh
cd $GRAFT_REPO_ROOT
run() { python bench.py --batch $1 --beams $2 --steps 8 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 Q=$1 B=$2', round(d['value'],1), 'q/s', round(d['ms_per_step'],2), 'ms')"; }
for v in old new; do
  cp tools/ab/lib_$v.so ripor_amd/libripor_hip.so
  run 1 10 $v; run 8 10 $v; run 32 10 $v; run 64 10 $v; run 1 1000 $v
done
cp tools/ab/lib_new.so ripor_amd/libripor_hip.so
RPR_GEMM_SKINNY=704 run 64 10 new_skinny704
RPR_GEMM_SKINNY=1408 run 128 10 new_skinny1408
run 128 10 new
