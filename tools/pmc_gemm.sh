#!/bin/bash
# SQ counters of the ping-pong GEMM alone (via gpurun): what fills the issue slots of gemm_h2_pp_kernel.
# Usage: tools/pmc_gemm.sh TAG   -> gpurun_out/TAG/sq_pmc.txt
TAG=${1:-pmc_gemm}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$n -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py 65536 > $O/$n.log 2>&1
done
python - <<PY
import csv, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_h2_pp_kernel" in r["Kernel_Name"]:
            agg[(r["Grid_Size"], r["Kernel_Name"].split("(")[0][-40:])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/sq_pmc.txt", "w") as out:
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1
        out.write(f"grid {k[0]} {k[1]}: launches {len(c.get('SQ_WAVE_CYCLES', []))}\n")
        for n in sorted(m):
            out.write(f"   {n:28s} {m[n]:16.0f}   {m[n] / wc:8.4f} of SQ_WAVE_CYCLES\n")
print(open("$O/sq_pmc.txt").read())
PY
