#!/bin/bash
# Kernel sequence of ONE search at a small in-flight batch (via gpurun): tools/latency_trace.sh TAG BATCH BEAMS [extra bench args]
TAG=${1:-lat}; Q=${2:-1}; BEAMS=${3:-10}; shift 3 || true
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
B="python $GRAFT_REPO_ROOT/bench.py --batch $Q --beams $BEAMS --steps 5 --warmup 2 --no-cpu-baseline --no-exact-fp32 --no-roofline $*"
$B --secondary "" > $O/plain.json 2> $O/plain.log
timeout 600 rocprofv3 --kernel-trace -d $O/trace -o b -- $B --secondary "" > $O/under.json 2> $O/trace.log
python $GRAFT_REPO_ROOT/tools/trace_dump.py $O/trace/b_results.db --seq 40 > $O/summary.txt
python - <<PY
import sqlite3
db=sqlite3.connect("$O/trace/b_results.db");c=db.cursor()
rows=list(c.execute("select name,start,duration,grid_x,workgroup_x from kernels order by start"))
idx=[i for i,r in enumerate(rows) if "init_beams" in r[0]]
i0=idx[-1]
j=i0
while j>0 and rows[j][1]-(rows[j-1][1]+rows[j-1][2])<500000: j-=1
t0=rows[j][1]
busy=0; n=0
with open("$O/last_search.txt","w") as f:
    for n_,s,d,g,w in rows[j:]:
        f.write(f"{(s-t0)/1e3:10.1f} {d/1e3:8.1f} {g//max(w,1):6d} {n_.replace('(anonymous namespace)::', '').split('(')[0][:70]}\n")
        busy+=d; n+=1
    end=rows[-1][1]+rows[-1][2]
    f.write(f"# {n} launches, wall {(end-t0)/1e3:.1f} us, kernel busy {busy/1e3:.1f} us\n")
print(open("$O/last_search.txt").read()[-200:])
PY
rm -rf $O/trace
head -c 300 $O/plain.json; echo; head -12 $O/summary.txt
