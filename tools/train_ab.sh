#!/bin/bash
# via gpurun: bf16 / f16x2 fine-tune step, this build against the round-4 library (ab_old/libripor_hip_r04.so), alternating
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-train_ab}; mkdir -p $O
cd $GRAFT_REPO_ROOT
cp ripor_amd/libripor_hip.so /tmp/new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp ab_old/libripor_hip_r04.so ripor_amd/libripor_hip.so; else cp /tmp/new.so ripor_amd/libripor_hip.so; fi
    for p in bf16 f16x2; do
      echo -n "$which $p rep$rep: "; timeout 300 python tools/train_bench.py --precision $p --steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d.get('ms_per_step', 0), 2), 'ms', round(d.get('value', 0), 1))"
    done
  done
done 2>&1 | tee $O/train_ab.txt
cp /tmp/new.so ripor_amd/libripor_hip.so
