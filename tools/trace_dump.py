#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace rocpd database of bench.py: per kernel name (template arguments kept) calls,
total and average duration; the time covered by launches shorter than 6 us (the near-empty launches of compacted
stages); idle gaps between consecutive kernels of the busiest window; and, with --seq N, the N longest launches in time
order. Usage: trace_dump.py results.db [--seq N] [--json out.json]"""
import json
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
scol = "start" if "start" in cols else "start_time"
ecol = "end" if "end" in cols else ("end_time" if "end_time" in cols else None)
q = f"select name, {scol}, duration from kernels order by {scol}"
rows = [(n, s, d) for n, s, d in cur.execute(q)]
t0, t1 = rows[0][1], rows[-1][1] + rows[-1][2]
agg = defaultdict(lambda: [0, 0.0])
short_n, short_t = 0, 0.0
for n, s, d in rows:
    k = n.replace("(anonymous namespace)::", "").split("(")[0].replace("rpr::", "")
    agg[k][0] += 1
    agg[k][1] += d
    if d < 6000:
        short_n += 1
        short_t += d
tot = sum(v[1] for v in agg.values())
print(f"{len(rows)} launches, wall {(t1 - t0) / 1e6:.1f} ms, kernel time {tot / 1e6:.1f} ms; "
      f"launches < 6 us: {short_n} = {short_t / 1e6:.2f} ms")
out = {}
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{t / 1e6:9.2f} ms {c:7d} x {t / c / 1e3:9.1f} us  {k[:110]}")
    out[k] = {"calls": c, "total_ms": t / 1e6, "avg_us": t / c / 1e3}
# idle gaps (single-stream runs): sum of max(0, next.start - cur.end)
gap = 0.0
end = rows[0][1] + rows[0][2]
for n, s, d in rows[1:]:
    if s > end:
        if s - end < 5e6:   # ignore host-side pauses between steps
            gap += s - end
    end = max(end, s + d)
print(f"idle gaps between kernels (< 5 ms each): {gap / 1e6:.2f} ms")
if "--seq" in sys.argv:
    N = int(sys.argv[sys.argv.index("--seq") + 1])
    big = sorted(rows, key=lambda r: -r[2])[:N]
    for n, s, d in sorted(big, key=lambda r: r[1]):
        print(f"  t={(s - t0) / 1e6:9.3f} ms  {d / 1e3:9.1f} us  {n.replace('(anonymous namespace)::', '').split('(')[0][:90]}")
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
