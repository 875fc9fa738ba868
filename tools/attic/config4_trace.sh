#!/bin/bash
# GPU: kernel trace of BASELINE config 4 (t5-large dims, beam 100, len 32) through bench.py's main leg
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/config4; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --model t5-large --beams 100 --batch ${1:-162} --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline"
$B --secondary "" > $O/plain.json 2> $O/plain.log
timeout 900 rocprofv3 --kernel-trace -d $O/trace -o b -- $B --secondary "" > $O/under.json 2> $O/trace.log
python $GRAFT_REPO_ROOT/tools/trace_dump.py $O/trace/b_results.db --seq 30 > $O/summary.txt
rm -rf $O/trace
head -c 700 $O/plain.json; echo; head -32 $O/summary.txt
