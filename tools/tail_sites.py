#!/usr/bin/env python3
"""Per-site durations of the projection GEMMs of the forced-tail pass, from a rocprofv3 --kernel-trace rocpd database of
`bench.py` (t5-base, B=10, L=32; run with --no-lanes for whole-chip launches): the tail pass of the first fork issues
12 decoder layers x [qkv, o, xq, xo, wi, wo] = 72 launches of gemm_h2_pp_kernel over all forced rows — by far the
longest launches of a search, so the 72 longest-per-search pp launches in time order are those sites.
Usage: tail_sites.py results.db rows_per_launch [out.json]"""
import json
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
M = int(sys.argv[2])
cur = db.cursor()
cols = [d[0] for d in cur.execute("select * from kernels limit 1").description]
tcol = "start" if "start" in cols else "start_time"
rows = list(cur.execute(f"select name, {tcol}, duration from kernels order by {tcol}"))
# a tail pass is bracketed by tail_embed_kernel and tail_gold_kernel; the ping-pong launches between them are its
# projection GEMMs in site order. Keep the passes of the first fork (the longest ones; the second fork forces a few rows).
passes, cur_pass = [], None
for n, t, d in rows:
    if "tail_embed_kernel" in n:
        cur_pass = []
    elif "tail_gold_kernel" in n:
        if cur_pass is not None and len(cur_pass) == 72:
            passes.append(cur_pass)
        cur_pass = None
    elif cur_pass is not None and "gemm_h2_pp_kernel" in n:
        cur_pass.append((n, t, d))
if not passes:
    sys.exit("no complete tail pass (72 ping-pong launches between tail_embed and tail_gold) in the trace")
longest = max(sum(r[2] for r in p) for p in passes)
passes = [p for p in passes if sum(r[2] for r in p) > 0.5 * longest]
big = [r for p in passes for r in p]
n_search = len(passes)
names = ["qkv", "o", "xq", "xo", "wi", "wo"]
shape = {"qkv": (2304, 768), "o": (768, 768), "xq": (768, 768), "xo": (768, 768), "wi": (3072, 768), "wo": (768, 3072)}
agg = defaultdict(list)
for s in range(n_search):
    for i, r in enumerate(big[s * 72:(s + 1) * 72]):
        agg[names[i % 6]].append(r[2] / 1e3)
out = {}
tot = 0.0
for n in names:
    us = sum(agg[n]) / max(1, len(agg[n]))
    N, K = shape[n]
    tiles = (M / 256.0) * (N / 256.0)
    out[n] = {"avg_us": us, "tflops": 2.0 * M * N * K / us / 1e6, "us_per_round_of_256_tiles": us / (tiles / 256.0)}
    tot += us * 12
    print(f"{n:4s} {us:9.1f} us  {out[n]['tflops']:6.1f} TF/s  {out[n]['us_per_round_of_256_tiles']:6.1f} us per round of 256 tiles")
out["_meta"] = {"searches": n_search, "rows_per_launch": M, "tail_gemm_ms_per_search": tot / 1e3,
                "k_loop_us_per_k_tile": (out["wo"]["us_per_round_of_256_tiles"] - out["xq"]["us_per_round_of_256_tiles"]) / 72.0}
print(f"tail GEMMs: {tot / 1e3:.1f} ms per search over {n_search} searches; K-loop "
      f"{out['_meta']['k_loop_us_per_k_tile']:.2f} us per 256x256x32 K-tile")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
