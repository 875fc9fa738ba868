"""Cycle trace of the ping-pong GEMM kernel's block 0 (diagnostic; GPU box).
RPR_GEMM_TRACE=/tmp/tr.txt RPR_GEMM_TRACE_W=18 python tools/gemm_trace_pp.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
ctx = E.Context.get(0)
M, N, K = 20480, 3072, int(os.environ.get("TRACE_K", 768))
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
for _ in range(3): ctx.linear(A, W)
torch.cuda.synchronize()
t = np.loadtxt(os.environ["RPR_GEMM_TRACE"]).reshape(K // 32, 8, 18)
cyc = t[..., :16]
print("K-tiles", K // 32)
for w in (0, 4):
    tile = cyc[1:, w, 0] - cyc[:-1, w, 0]
    print(f"wave {w}: cycles per K-tile median {np.median(tile):.0f} min {tile.min():.0f} max {tile.max():.0f}")
    mid = cyc[2:-2, w, :]
    nxt0 = cyc[3:-1, w, 0]
    seg = []
    NPH = int(os.environ.get("TRACE_PHASES", 2))   # phases per K-tile of the build under test (2 since round 4, 4 before)
    for ph in range(NPH):
        L = mid[:, ph * 4 + 1] - mid[:, ph * 4 + 0]
        B1 = mid[:, ph * 4 + 2] - mid[:, ph * 4 + 1]
        Mm = mid[:, ph * 4 + 3] - mid[:, ph * 4 + 2]
        end = mid[:, ph * 4 + 4] if ph < NPH - 1 else nxt0
        B2 = end - mid[:, ph * 4 + 3]
        seg.append((np.median(L), np.median(B1), np.median(Mm), np.median(B2)))
    print("   per phase median (L, wait@B1, M, wait@B2):", [tuple(int(x) for x in s) for s in seg])
rt = t[:, 0, 16]
dt_cyc = cyc[-1, 0, 0] - cyc[1, 0, 0]; dt_rt = rt[-1] - rt[1]
print(f"shader clock over the K-loop: {dt_cyc / (dt_rt / 100e6) / 1e9:.3f} GHz  (memtime cycles {dt_cyc:.0f}, realtime ticks {dt_rt:.0f})")
