#!/bin/bash
# via gpurun: row split of ragged 256-tile launches (launch_gemm_h2, RPR_GEMM_ROWSPLIT) on / off at the row counts where the
# last round is badly filled, the two parts alone, and the single-query / 64-query searches of the bench
export RPR_DEV_LIB=1   # the switches below are development switches: libripor_hip_dev.so (same sources, -DRPR_DEV_SWITCHES)
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-rowsplit}; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rs in 0 1; do
  echo "== RPR_GEMM_ROWSPLIT=$rs"
  RPR_GEMM_ROWSPLIT=$rs timeout 300 python tools/gemm_bench.py 27000 28000 17920 37120 2>/dev/null | grep -v "N=256"
done 2>&1 | tee $O/gemm.txt
echo "== parts alone (split off)"; RPR_GEMM_ROWSPLIT=0 timeout 300 python tools/gemm_bench.py 21760 5240 2>/dev/null | grep -v "N=256" | tee -a $O/gemm.txt
for rs in 0 1 0 1; do
  echo -n "RPR_GEMM_ROWSPLIT=$rs: "
  RPR_GEMM_ROWSPLIT=$rs timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary latency,small_batch 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']
print('headline', round(d['value'],1), 'b1000', round(s['latency']['beams1000']['value'],2), 'ms  b10', round(s['latency']['beams10']['value'],2), 'ms  q8', round(s['small_batch']['q8']['value'],1), ' q64', round(s['small_batch']['q64']['value'],1))"
done 2>&1 | tee $O/bench.txt
