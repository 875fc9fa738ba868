"""Time rpr_op_linear on a list of M,N,K shapes (diagnostic; GPU box). RPR_GEMM_TILE=256|64|128 forces the kernel.
Usage: python tools/gemm_shapes.py 8192x768x768 768x3072x8192 ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
ctx = E.Context.get(0)
for spec in sys.argv[1:]:
    M, N, K = (int(v) for v in spec.split("x"))
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    for _ in range(3): ctx.linear(A, W, None)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(10): ctx.linear(A, W, None)
    torch.cuda.synchronize()
    pg = ctx.profile_get(); ctx.profile_enable(False)
    st = pg["gemm"] if pg["gemm"]["launches"] else pg["gemm_small"]
    us = st["total_ms"] / st["launches"] * 1e3
    print(f"tile={os.environ.get('RPR_GEMM_TILE', 'auto'):>4s} M={M:5d} N={N:5d} K={K:5d}: {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF", flush=True)
