#!/bin/bash
# GPU: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the training step's GEMM kernels. Usage: tools/train_pmc.sh TAG
TAG=${1:-trainpmc}; O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/train_bench.py --bz 128 --steps 1 --precision bf16 > /dev/null 2> $O/$c.log
done
python - <<PY
import csv, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open("$O/pmc_%s/p_counter_collection.csv" % c)):
        if r["Counter_Name"] == c and "gemm" in r["Kernel_Name"]:
            key = r["Kernel_Name"].split("(")[0][-60:] + " grid " + r.get("Grid_Size", "?")
            agg[key].append(float(r["Counter_Value"]))
    print(c, "(KiB per launch; bytes = x1024, FETCH_SIZE x2 on gfx950 per the guide's correction)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print(f"   {k:90s} n={len(v):4d} mean {sum(v)/len(v)/1024:10.1f} MiB-units")
PY
rm -rf $O/pmc_*/*.db
