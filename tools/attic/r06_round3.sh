#!/bin/bash
# via gpurun: whole GPU suite on the pruned build, GEMM routes beyond 1400 rows, the new bench legs. Usage: tools/r06_round3.sh TAG
TAG=${1:-r06d}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 -x -s > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR" $O/pytest.log | tail -30
export RPR_DEV_LIB=1
for mx in 1400 2600; do
  RPR_GEMM_WSPLIT_MAX=$mx python tools/gemm_bench.py 1700 2100 2560 2>&1 | grep weighted | sed "s/^/wsplit_max=$mx: /"
done | tee $O/gemm_wsplit_max.txt
unset RPR_DEV_LIB
timeout 1500 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-exact-fp32 --secondary latency,beam1000,rankdata_ref_flags,small_batch,f2 > $O/bench.json 2> $O/bench.log
echo "bench rc=$?"; tail -12 $O/bench.log
