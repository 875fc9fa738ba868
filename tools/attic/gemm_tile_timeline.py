"""Per-tile timeline of the persistent ping-pong GEMM kernel's block 0 (diagnostic; GPU box): how long are prologue
(start -> first K-tile landed), K-loop and epilogue (K-loop done -> stores issued, LDS free) of a tile when every CU runs
the same launch?  RPR_GEMM_TRACE=/tmp/tr.txt python tools/gemm_tile_timeline.py [M N K resid]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
ctx = E.Context.get(0)
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (256 * 1176, 2304, 768)
resid = len(sys.argv) > 4 and sys.argv[4] == "1"
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
R = torch.randn(M, N, device="cuda") if resid else None
for _ in range(2): ctx.linear(A, W, R)
torch.cuda.synchronize()
t = np.loadtxt(os.environ["RPR_GEMM_TRACE"] + ".tiles").reshape(-1, 4) * 0.01   # us
pro, kl, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
gap = t[1:, 0] - t[:-1, 3] if len(t) > 1 else np.zeros(1)
print(f"M={M} N={N} K={K} resid={resid}: {len(t)} tiles of block 0; per tile (us) median [min..max]:")
for name, v in (("prologue", pro), ("K-loop", kl), ("epilogue", epi), ("between tiles", gap)):
    print(f"  {name:14s} {np.median(v):7.2f} [{v.min():6.2f} .. {v.max():6.2f}]")
if len(t) > 1: print(f"  tile period    {np.median(t[1:, 0] - t[:-1, 0]):7.2f}")
