#!/bin/bash
# via gpurun: the part of tools/r06_final.sh that did not run (bench as the driver runs it), the lane-split test, and 64 queries in
# flight with and without the lane split. Usage: tools/r06_final2.sh TAG
TAG=${1:-r06_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_edges.py -m gpu -q -k lane_split 2>&1 | tail -3
T0=$(date +%s); timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.log; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
for lanes in default 600; do
  if [ $lanes != default ]; then export RPR_LANE_MIN_ROWS=$lanes; fi
  for rep in 1 2; do
    timeout 600 python bench.py --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('q64 lane_min_rows=$lanes rep$rep', round(d['value'],1), 'q/s', round(d['ms_per_step'],2), 'ms', d['config']['lanes'][:12])"
  done
done | tee $O/q64_lanes_ab.txt
