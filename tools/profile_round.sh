#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line, rocprofv3 kernel stats, HBM traffic PMC passes.
# Usage: tools/profile_round.sh rNN [extra bench args]
set -u
TAG=${1:-rXX}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py $*"
# 1. plain bench (with cpu_baseline)
timeout 600 $B > $OUT/bench.json 2> $OUT/bench.log
# 2. same command under rocprofv3 --kernel-trace --stats
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# 3. HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4), kernel-trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/pmc_write.log
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/bench_results.db $OUT/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/gemm_sites.py $OUT/trace/bench_results.db $OUT/gemm_sites.json | tee $OUT/gemm_sites.txt
python - <<PY
import csv, collections, json
out = {}
for name, f in (("FETCH_SIZE", "$OUT/pmc_fetch/p_counter_collection.csv"), ("WRITE_SIZE", "$OUT/pmc_write/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "rpr::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[name] = {k: {"launches": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for k, v in agg.items()}
json.dump(out, open("$OUT/hbm_pmc.json", "w"), indent=1)
print(json.dumps({k: {kk: round(vv["mean"], 1) for kk, vv in v.items()} for k, v in out.items()}, indent=1)[:3000])
PY
rm -rf $OUT/trace $OUT/pmc_fetch/*.db
tail -3 $OUT/bench.log; head -c 400 $OUT/bench.json
