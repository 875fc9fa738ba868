"""Micro-benchmark of the bf16 training GEMM routes (rpr_op_linear_bf16) on the products of the fine-tune step at bz 128
(8192 decoder / 4096 encoder rows). GEMM launch time from the library's hipEvents. Run on the GPU box.
Usage: [RPR_DEV_LIB=1 RPR_...=..] python tools/gemm_bf16_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E

ctx = E.Context.get(0)
shapes = [(8192, 768, 768, True), (8192, 768, 768, False), (8192, 768, 2304, False), (8192, 768, 3072, True), (8192, 2304, 768, False),
          (8192, 3072, 768, False), (4096, 768, 768, True), (4096, 768, 3072, True), (4096, 2304, 768, False), (4096, 3072, 768, False)]
tot = 0.0
for M, N, K, resid in shapes:
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda") if resid else None
    for _ in range(3):
        ctx.linear_bf16(A, W, R)
    torch.cuda.synchronize()
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(20):
        ctx.linear_bf16(A, W, R)
    torch.cuda.synchronize()
    pg = ctx.profile_get(); ctx.profile_enable(False)
    us = (pg["gemm"]["total_ms"] + pg["gemm_small"]["total_ms"]) / 20 * 1e3
    byt = 2.0 * (M * K + N * K) + 4.0 * M * N * (2 if resid else 1)
    print(f"M={M} N={N} K={K} resid={int(resid)}: gemm {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF  {byt / us / 1e3:6.2f} GB/s-k", flush=True)
    tot += us
print(f"sum {tot:.1f} us")
