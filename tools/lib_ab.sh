#!/bin/bash
# via gpurun: this build against an older library (ab_old/<name>.so, git-ignored but it travels), alternating on the same box:
# the GEMM micro-bench and a short headline bench. Usage: tools/lib_ab.sh TAG OLD_SO_NAME
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-lib_ab}; mkdir -p $O
OLD=ab_old/${2:-libripor_hip_r05q.so}
cd $GRAFT_REPO_ROOT
cp ripor_amd/libripor_hip.so /tmp/new.so
for rep in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp $OLD ripor_amd/libripor_hip.so; else cp /tmp/new.so ripor_amd/libripor_hip.so; fi
    echo "== $which rep$rep: $(timeout 300 python tools/gemm_bench.py 21760 2>/dev/null | tail -1)"
    timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --secondary "" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   bench', round(d['value'],1), 'q/s', round(d['roofline']['avg_launch_us'],1), 'us/launch')"
  done
done 2>&1 | tee $O/lib_ab.txt
cp /tmp/new.so ripor_amd/libripor_hip.so
