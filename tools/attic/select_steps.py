"""Per-call durations of select_kernel from a rocprofv3 rocpd database (diagnostic)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
rows = list(cur.execute("select name, start, duration from kernels where name like '%select_kernel%' order by start"))
d = [r[2] / 1e3 for r in rows]
n = 32
print("calls", len(d))
for s in range(0, len(d), n):
    print(" ".join(f"{x:7.0f}" for x in d[s:s + n]))
