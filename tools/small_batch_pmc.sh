#!/bin/bash
# via gpurun: HBM traffic of ONE search at 1 / 8 / 64 queries in flight (beam 10): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# separate passes (TCC slots), summed over every rpr:: kernel -> gpurun_out/TAG/small_batch_pmc.json (copy to
# profiles/latest_small_batch_pmc.json: bench.py's small_batch leg reads it for roofline.traffic). Usage: tools/small_batch_pmc.sh TAG
TAG=${1:-sbpmc}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "1 10" "8 10" "64 10"; do
  set -- $cfg
  B="python $GRAFT_REPO_ROOT/bench.py --batch $1 --beams $2 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-exact-fp32 --secondary ''"
  for c in FETCH_SIZE WRITE_SIZE; do
    eval timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/q$1_b$2_$c -o p -- $B > /dev/null 2> $O/q$1_b$2_$c.log
  done
done
python - <<PY
import csv, json, sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
import __graft_entry__ as ge
out = {"_meta": {"build": {"tag": "$TAG", "source_hash": ge.source_hash()[:16]},
                 "note": "bench.py --steps 1 --warmup 0 runs the search twice (resident-input loop + PCIe-inclusive loop): sums are halved; "
                         "KB as rocprofv3 reports them (FETCH_SIZE counts half of a wide coalesced read on gfx950: x2 when compared with bytes)"}}
for q, b in ((1, 10), (8, 10), (64, 10)):
    e = {}
    for c, key in (("FETCH_SIZE", "fetch_kb_per_search"), ("WRITE_SIZE", "write_kb_per_search")):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(f"$O/q{q}_b{b}_{c}/p_counter_collection.csv")):
            if r["Counter_Name"] == c and "rpr::" in r["Kernel_Name"] and "split_planes" not in r["Kernel_Name"] and "max_row_norm" not in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
        e[key] = tot / 2.0
        e["dispatches_per_search"] = n / 2.0
    out[f"q{q}_b{b}"] = e
json.dump(out, open("$O/small_batch_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/q*_SIZE
