#!/bin/bash
# via gpurun: the round's closing run — whole GPU suite, smoke, the bench as the driver runs it (+ the opt-in t5_3b leg), the kernel
# sequence of one beam-1000 search. Usage: tools/r06_final.sh TAG
TAG=${1:-r06_final}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR" $O/pytest.log | tail -20
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
T0=$(date +%s); timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.log; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary t5_3b > $O/bench_3b.json 2> $O/bench_3b.log; echo "bench 3b rc=$?"
bash tools/latency_trace.sh ${TAG}_q1_b1000 1 1000 > /dev/null 2>&1; tail -1 gpurun_out/${TAG}_q1_b1000/last_search.txt
