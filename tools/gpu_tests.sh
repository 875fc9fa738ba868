#!/bin/bash
# GPU test-suite only (via gpurun, from the repo root). Usage: tools/gpu_tests.sh TAG [pytest-args...]
set -u
TAG=${1:-rXX}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q --maxfail=10 -s "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR" $OUT/pytest.log | tail -30
