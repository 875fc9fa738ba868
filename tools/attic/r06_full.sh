#!/bin/bash
# via gpurun: the whole GPU suite, smoke, then the kernel sequence of one beam-1000 search. Usage: tools/r06_full.sh TAG
TAG=${1:-r06b}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 -x -s > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR" $O/pytest.log | tail -30
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"
bash tools/latency_trace.sh ${TAG}_q1_b1000 1 1000
