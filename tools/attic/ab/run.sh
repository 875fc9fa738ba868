#!/bin/bash
# same-box A/B of two builds of the library: tools/ab/run.sh [rounds] [extra bench args]
N=${1:-3}; shift || true
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary \"\" $*"
for i in $(seq $N); do
  for v in old new; do
    cp tools/ab/lib_$v.so ripor_amd/libripor_hip.so
    eval $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
cp tools/ab/lib_new.so ripor_amd/libripor_hip.so
