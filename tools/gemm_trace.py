import os, sys, torch
sys.path.insert(0, "/root/repo")
from ripor_amd import engine as E
ctx = E.Context.get(0)
M, N, K = 20480, 3072, 768
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
for _ in range(3): ctx.linear(A, W)
torch.cuda.synchronize()
