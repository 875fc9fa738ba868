#!/bin/bash
# same-box A/B of the exact-fp32 mode: tools/ab/f32.sh
cd $GRAFT_REPO_ROOT
for v in old new old new; do
  cp tools/ab/lib_$v.so ripor_amd/libripor_hip.so
  python bench.py --precision f32 --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v f32', round(d['value'],1), 'q/s', round(d['ms_per_step'],1), 'ms')"
done
cp tools/ab/lib_new.so ripor_amd/libripor_hip.so
