cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/b1000; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --batch 1 --beams 1000 --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline"
$B --secondary "" > $O/plain.json 2> $O/plain.log
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o b -- $B --secondary "" > $O/under.json 2> $O/trace.log
python $GRAFT_REPO_ROOT/tools/trace_dump.py $O/trace/b_results.db --seq 40 > $O/summary.txt
python - <<PY
import sqlite3
db=sqlite3.connect("$O/trace/b_results.db");c=db.cursor()
rows=list(c.execute("select name,start,duration,grid_x,workgroup_x from kernels order by start"))
# last search: find last init_beams_kernel
idx=[i for i,r in enumerate(rows) if "init_beams" in r[0]]
# the encoder precedes init_beams; take from the previous tail_rank/finalize end
i0=idx[-1]
# walk back to first kernel after a gap > 1 ms
j=i0
while j>0 and rows[j][1]-(rows[j-1][1]+rows[j-1][2])<500000: j-=1
t0=rows[j][1]
with open("$O/last_search.txt","w") as f:
    for n,s,d,g,w in rows[j:]:
        f.write(f"{(s-t0)/1e3:10.1f} {d/1e3:8.1f} {g//max(w,1):6d} {n.split('(')[0][:70]}\n")
PY
rm -rf $O/trace
cat $O/plain.json | head -c 600
