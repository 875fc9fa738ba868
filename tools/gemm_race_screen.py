"""Race screen of the split-precision GEMM kernels (ping-pong 256x256, LDS-DMA 128-row, skinny): the same launch is
repeated and every output must be bitwise identical to the first, which itself is checked against fp64.
(LDS-DMA ordering bugs show up as rare wrong tiles that depend on timing.)  GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E

ctx = E.Context.get(0)
shapes = [(20480, 768, 768), (20480, 2304, 768), (20480, 3072, 768), (20480, 768, 3072), (5120, 768, 768),
          (10, 768, 768), (10, 768, 3072), (320, 2304, 768), (640, 3072, 768), (1000, 768, 3072), (20470, 832, 768)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = 0
for M, N, K in shapes:
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda")
    first = ctx.linear(A, W, R)
    err = (first.double() - (A.double() @ W.double().t() + R.double())).abs().max().item()
    diff = 0
    for _ in range(reps):
        out = ctx.linear(A, W, R)
        diff += int((out != first).any().item())
    bad += diff + (err > 1e-4)
    print(f"M={M} N={N} K={K}: maxerr {err:.2e}, {diff}/{reps} repetitions differ", flush=True)
print("RACE SCREEN", "FAILED" if bad else "clean")
