#!/bin/bash
# via gpurun: super-tile order of the persistent 256 x 256 kernel (GemmH2Args::tile_cw) against the row-major order, development
# build, alternating on one box: GEMM micro-bench, headline bench, fabric traffic (FETCH_SIZE / WRITE_SIZE) of the wide products.
# Usage: tools/r06_supertile.sh TAG
TAG=${1:-r06_supertile}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_parity.log 2>&1; echo "parity rc=$?"; tail -1 $O/pytest_parity.log
{
for rep in 1 2; do
  for v in 0 1; do
    echo "== supertile=$v rep$rep"
    RPR_DEV_LIB=1 RPR_PP_SUPERTILE=$v timeout 300 python tools/gemm_bench.py 53760 2>/dev/null | tail -6
    RPR_DEV_LIB=1 RPR_PP_SUPERTILE=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-exact-fp32 --secondary "" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   bench', round(d['value'],1), 'q/s', round(d['roofline']['avg_launch_us'],1), 'us/launch', 'board', d.get('board_power'))"
  done
done
} 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    RPR_DEV_LIB=1 RPR_PP_SUPERTILE=$v timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_${v}_$c -o p -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py 53760 > $O/pmc_${v}_$c.log 2>&1
  done
done
python - <<PY | tee $O/pmc.txt
import csv, collections, glob
for v in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("$O/pmc_%d_%s/**/p_counter_collection.csv" % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_h2_pp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    agg[r["Dispatch_Id"]][c].append(float(r["Counter_Value"]))
    # launches in dispatch order: gemm_bench runs 3 + 20 + 1 launches per shape, five shapes
    ids = sorted(agg, key=int)
    print("supertile=%d: %d pp launches" % (v, len(ids)))
    per = collections.defaultdict(list)
    for i, d in enumerate(ids):
        per[i // 24].append((sum(agg[d].get("FETCH_SIZE", [0])), sum(agg[d].get("WRITE_SIZE", [0]))))
    for k, rows in per.items():
        f = sum(r[0] for r in rows) / len(rows); w = sum(r[1] for r in rows) / len(rows)
        print("   shape #%d: FETCH %.1f MB (x2 = %.1f), WRITE %.1f MB  -> (2F + W) %.1f MB per launch" % (k, f / 1024, 2 * f / 1024, w / 1024, (2 * f + w) / 1024))
PY
