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
# 2. same command under rocprofv3 --kernel-trace --stats; --no-roofline --secondary "": the trace holds the graph-replayed
#    timed region only, so the CSV average of gemm_h2_pp_kernel is that of the timed configuration's lane launches
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $B --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
# 3. HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 > 4), kernel-trace only
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-exact-fp32 --secondary "" > /dev/null 2> $OUT/pmc_fetch.log
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-exact-fp32 --secondary "" > /dev/null 2> $OUT/pmc_write.log
# 4. matrix-pipe utilisation and effective clock: SQ_VALU_MFMA_BUSY_CYCLES (cycles, = 32 per 32x32x16 f16 MFMA) and
#    GRBM_GUI_ACTIVE (shader-clock cycles the GPU was busy; / kernel duration = effective clock under DVFS)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_mfma -o p -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-exact-fp32 --secondary "" > /dev/null 2> $OUT/pmc_mfma.log
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $OUT/trace/bench_results.db $OUT/kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/trace_dump.py $OUT/trace/bench_results.db --json $OUT/trace_summary.json | tee $OUT/trace_summary.txt | head -30
# per-site times of the tail-pass GEMMs on whole-chip launches (one more traced run without lanes)
timeout 600 rocprofv3 --kernel-trace -d $OUT/trace_nolanes -o bench -- $B --no-lanes --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary "" > $OUT/bench_nolanes_under_rocprof.json 2> $OUT/trace_nolanes.log
python $GRAFT_REPO_ROOT/tools/tail_sites.py $OUT/trace_nolanes/bench_results.db $((2176 * 10 * 28)) $OUT/tail_gemm_sites.json | tee $OUT/tail_gemm_sites.txt
rm -rf $OUT/trace_nolanes
python - <<PY
import csv, collections, json
out = {}
for name, f in (("FETCH_SIZE", "$OUT/pmc_fetch/p_counter_collection.csv"), ("WRITE_SIZE", "$OUT/pmc_write/p_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "rpr::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[name] = {k: {"launches": len(v), "sum": sum(v), "mean": sum(v) / len(v)} for k, v in agg.items()}
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import __graft_entry__ as ge
build = {"tag": "$TAG", "source_hash": ge.source_hash()[:16]}   # which build of libripor_hip.so the counters / the trace belong to
out["_meta"] = {"steps_in_pmc_pass": 2, "build": build, "note": "bench.py --steps 1 --warmup 0 runs the step twice (resident-input loop + PCIe-inclusive loop); launches = dispatches counted over both"}
json.dump(out, open("$OUT/hbm_pmc.json", "w"), indent=1)
json.dump({"build": build, "command": "$B --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary ''"}, open("$OUT/kernel_stats.meta.json", "w"), indent=1)
# MFMA pass: per kernel mean counter values per launch + mean duration of the same dispatches (kernel trace of that pass)
try:
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open("$OUT/pmc_mfma/p_kernel_trace.csv")):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"].split("(")[0], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$OUT/pmc_mfma/p_counter_collection.csv")):
        if "rpr::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                agg[k]["duration_ns"].append(dur[r["Dispatch_Id"]][1])
    mf = {}
    for k, c in agg.items():
        e = {n: sum(v) / len(v) for n, v in c.items()}
        e["launches"] = len(c.get("GRBM_GUI_ACTIVE", []))
        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (memory-bound kernels read 19.0e9 / s = 8 x 2.38 GHz, the
        # chip's maximum clock); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (it equals 32 cycles x the number
        # of 32x32x16 MFMAs the launch issues)
        if e.get("duration_ns") and e.get("GRBM_GUI_ACTIVE"):
            e["effective_clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / e["duration_ns"]
        if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy_frac_of_active_cycles"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            e["mfma_frac_of_nominal_peak"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["duration_ns"] * 2.4 * 1024.0)
        mf[k] = e
    json.dump(mf, open("$OUT/mfma_pmc.json", "w"), indent=1)
    for k, e in sorted(mf.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:4]:
        print("MFMA", k[:60], {n: round(v, 4) if v < 100 else round(v) for n, v in e.items()})
except Exception as ex:
    print("mfma pmc summary failed:", ex)
print(json.dumps({k: {kk: round(vv["mean"], 1) for kk, vv in v.items()} for k, v in out.items() if k != "_meta"}, indent=1)[:3000])
PY
rm -rf $OUT/trace $OUT/pmc_fetch/*.db $OUT/pmc_mfma/*.db
tail -3 $OUT/bench.log; head -c 400 $OUT/bench.json
