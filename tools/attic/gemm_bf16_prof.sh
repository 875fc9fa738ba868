#!/bin/bash
# GPU: kernel times of the bf16 GEMM routes per shape (rocprofv3 kernel trace of tools/gemm_bf16_bench.py). Usage: tools/gemm_bf16_prof.sh TAG
TAG=${1:-bf16gemm}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in 24 0; do
  RPR_BF16_W128=$w timeout 300 rocprofv3 --kernel-trace -d $O/t$w -o b -- python $GRAFT_REPO_ROOT/tools/gemm_bf16_bench.py > $O/bench$w.log 2>&1
  python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("$O/t$w/*_results.db")[0])
tab = [r[0] for r in db.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")]
q = "select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%gemm%' group by name, grid_x order by min(start)"
print("RPR_BF16_W128=$w")
for name, gx, n, avg, mn in db.execute(q):
    print(f"   {name.split('(')[0][-58:]:58s} grid {gx:>8} n={n:4d} avg {avg/1e3:8.1f} us min {mn/1e3:8.1f}")
PY
  rm -rf $O/t$w
done
