#!/usr/bin/env python3
"""Per-site counter values of the tail-pass projection GEMMs from a `rocprofv3 --pmc X --kernel-trace --output-format csv`
run of bench.py (lanes on: rows per launch = 1075 x 10 x 28). The tail launches are the 6-periodic run of the largest
gemm_h2_pp_kernel dispatches (qkv, o, xq, xo, wi, wo per decoder layer).
Usage: pmc_sites.py p_counter_collection.csv COUNTER [rows_per_launch]"""
import collections
import csv
import sys

path, name = sys.argv[1], sys.argv[2]
M = int(sys.argv[3]) if len(sys.argv) > 3 else 1075 * 280
rows = []
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == name and "gemm_h2_pp_kernel" in r["Kernel_Name"]:
        rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
rows.sort()
vals = [v for _, v in rows]
# the tail launches: maximal runs of 72 consecutive pp dispatches whose values repeat with period 6
best = []
for i in range(len(vals) - 71):
    seg = vals[i:i + 72]
    if all(abs(seg[k] - seg[k % 6]) <= 0.08 * max(seg[k % 6], 1.0) for k in range(72)) and min(seg[:6]) > 0:
        best.append((sum(seg), i))
best.sort(reverse=True)
names = ["qkv", "o", "xq", "xo", "wi", "wo"]
shape = {"qkv": (2304, 768), "o": (768, 768), "xq": (768, 768), "xo": (768, 768), "wi": (3072, 768), "wo": (768, 3072)}
if not best:
    sys.exit("no 6-periodic run of 72 launches found")
i = best[0][1]
seg = vals[i:i + 72]
agg = collections.defaultdict(list)
for k, v in enumerate(seg):
    agg[names[k % 6]].append(v)
print(f"{name} of the tail-pass GEMMs (KB per launch, mean of 12 layers), rows per launch {M}")
for n in names:
    v = sum(agg[n]) / len(agg[n])
    N, K = shape[n]
    a_bytes = M * K * 4
    resid = M * N * 4 if n in ("o", "xo", "wo") else 0
    out_bytes = M * N * 4
    if name == "FETCH_SIZE":
        print(f"  {n:4s} {v * 1024 / 1e9:7.2f} GB raw, x2 (gfx950 wide-read correction) {2 * v * 1024 / 1e9:7.2f} GB; algorithmic reads "
              f"{(a_bytes + resid + N * K * 4) / 1e9:5.2f} GB; A panel x column tiles = {a_bytes * (N / 256) / 1e9:6.2f} GB")
    else:
        print(f"  {n:4s} {v * 1024 / 1e9:7.2f} GB; algorithmic writes {out_bytes / 1e9:5.2f} GB")
