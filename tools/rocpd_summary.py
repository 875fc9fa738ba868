#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result into the per-kernel stats summary committed under
profiles/: name, calls, total/avg/min/max duration. Usage: rocpd_summary.py results.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
try:
    cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
except Exception:
    cols = []
if "duration" in cols and "name" in cols:
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
        "group by name order by sum(duration) desc"))
    hdr = ["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"]
else:
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    hdr = ["Name", "Calls", "TotalDuration(us)", "Average(us)", "Percentage"]
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(hdr)
    for r in rows:
        w.writerow(list(r))
print(f"{len(rows)} kernels -> {sys.argv[2]}")
