#!/bin/bash
# GPU: rocprofv3 kernel trace of the training bench. Usage: bash tools/train_profile.sh <tag> [precision]
tag=${1:-trainprof}; prec=${2:-f16x2}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python tools/train_bench.py --bz 128 --steps 3 --precision $prec > $out/bench.json 2> $out/bench.log
f=$(find $out/prof -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$f" $out/kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:28]:
    print(f'{r["Name"][:70]:70s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["AverageNs"])/1e3:9.1f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%')
print("total", tot/1e6, "ms over 5 steps (1 warm + 3 timed + 1 profiled)")
PY
python - <<PY
import sqlite3
db = sqlite3.connect("$f")
for pat in ("absmax2", "split_dyn_T", "gemm_h2_dma_kernel<128, 64", "gemm_h2_dma_kernel<128, 128", "gemm_h2_pp", "colsum"):
    rows = list(db.execute("select grid_x, grid_y, count(*), avg(duration), sum(duration) from kernels where name like ? group by grid_x, grid_y order by sum(duration) desc limit 12", (f"%{pat}%",)))
    print(pat)
    for gx, gy, n, avg, tot in rows:
        print(f"   grid {gx:>8} x {gy:<4} n={n:5d} avg {avg/1e3:8.1f} us total {tot/1e6:8.2f} ms")
PY
python tools/train_streams.py "$f"
rm -rf $out/prof
