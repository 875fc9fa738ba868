#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-launch_probe2}; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 120 tools/attic/launch_probe2 > $O/probe2.txt 2>&1
timeout 300 python tools/attic/launch_probe3.py > $O/probe3.txt 2>&1
B="python bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary ''"
( timeout 300 bash -c "$B --no-graph" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager', round(d['ms_per_step'],3), 'ms')" ) > $O/bench_eager.txt 2>&1
cat $O/probe2.txt $O/probe3.txt $O/bench_eager.txt
