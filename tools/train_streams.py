#!/usr/bin/env python3
"""Per-stream view of one training step from a rocprofv3 --kernel-trace rocpd database of tools/train_bench.py: for the
last complete step (delimited by the adamw launches), per stream: launches, busy time, idle gaps, and the kernels of the
main stream ranked by total time. Usage: train_streams.py results.db"""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, duration, stream_id from kernels order by start"))
# steps end with a run of adamw_kernel launches; take the span between the ends of the last two runs
ad = [i for i, r in enumerate(rows) if "adamw" in r[0]]
ends = [ad[k] for k in range(len(ad)) if k + 1 == len(ad) or ad[k + 1] != ad[k] + 1]
# the final profiled backward has no adamw: use the last two adamw runs
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
t0, t1 = step[0][1], step[-1][1] + step[-1][2]
print(f"step wall {(t1 - t0) / 1e6:.2f} ms, {len(step)} launches")
per = defaultdict(list)
for n, s, d, sid in step:
    per[sid].append((n, s, d))
for sid, ks in sorted(per.items(), key=lambda kv: -sum(k[2] for k in kv[1])):
    busy = sum(k[2] for k in ks)
    gaps = 0
    end = ks[0][1] + ks[0][2]
    for n, s, d in ks[1:]:
        if s > end: gaps += s - end
        end = max(end, s + d)
    print(f"stream {sid}: {len(ks)} launches, busy {busy / 1e6:.2f} ms, gaps {gaps / 1e6:.2f} ms, span {(end - ks[0][1]) / 1e6:.2f} ms")
main = max(per.items(), key=lambda kv: len(kv[1]))[1]
agg = defaultdict(lambda: [0, 0.0])
for n, s, d in main:
    k = n.split("(")[0].replace("rpr::", "")[:60]
    agg[k][0] += 1; agg[k][1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"   main: {t / 1e6:7.2f} ms {c:5d} x {t / c / 1e3:7.1f} us  {k}")
