#!/usr/bin/env python3
"""Per-site durations of the dominant GEMM kernel inside the search step, from a rocprofv3 --kernel-trace rocpd
database of `bench.py` (t5-base, B=10, L=32). The 2281 launches of gemm_h2_pp_kernel per step come in a fixed order:
12 encoder layers x [qkv, o, wi, wo], the cross-K/V projection, then (steps 1..31) x 12 decoder layers x
[qkv, o, xq, xo, wi, wo] (step 0 and the logits use the small-tile kernels).
Usage: gemm_sites.py results.db [out.json]"""
import json
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
tcol = "start" if "start" in cols else ("start_time" if "start_time" in cols else cols[0])
rows = list(cur.execute(f"select name, duration from kernels where name like '%gemm_h2_pp_kernel%' order by {tcol}"))
dur = [r[1] for r in rows]
PER = 2281
names = []
for _ in range(12):
    names += ["enc_qkv", "enc_o", "enc_wi", "enc_wo"]
names.append("xkv")
for _ in range(31 * 12):
    names += ["dec_qkv", "dec_o", "dec_xq", "dec_xo", "dec_wi", "dec_wo"]
assert len(names) == PER
agg = defaultdict(list)
nsteps = len(dur) // PER
for s in range(nsteps):
    for n, d in zip(names, dur[s * PER:(s + 1) * PER]):
        agg[n].append(d)
out = {n: {"launches_per_step": len(v) // max(1, nsteps), "avg_us": sum(v) / len(v) / 1e3,
           "ms_per_step": sum(v) / 1e6 / max(1, nsteps)} for n, v in agg.items()}
print(f"{len(dur)} launches = {nsteps} steps x {PER} (+{len(dur) - nsteps * PER} left over); columns: {cols}")
for n, v in sorted(out.items(), key=lambda kv: -kv[1]["ms_per_step"]):
    print(f"{n:8s} {v['launches_per_step']:5d} x {v['avg_us']:8.1f} us = {v['ms_per_step']:7.2f} ms/step")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
