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
rm -rf $out/prof
