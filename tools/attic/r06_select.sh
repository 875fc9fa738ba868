#!/bin/bash
# via gpurun: the radix selection's tests, then the kernel sequence of one beam-1000 search. Usage: tools/r06_select.sh TAG
TAG=${1:-r06a}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_select_radix.py "tests/test_gpu_forced_tail.py::test_grouped_selection_large_beams_few_queries" \
  "tests/test_gpu_forced_tail.py::test_exact_score_ties_resolve_identically_on_every_path" \
  "tests/test_gpu_edges.py::test_beam_1000_like_the_reference_retrieval_script" "tests/test_gpu_edges.py::test_trie_level_tables_change_nothing" \
  -m gpu -q --maxfail=8 -s > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR|Error" $O/pytest.log | tail -30
bash tools/latency_trace.sh ${TAG}_q1_b1000 1 1000
grep -E "select|rs_" gpurun_out/${TAG}_q1_b1000/summary.txt | head
