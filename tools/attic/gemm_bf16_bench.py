#!/usr/bin/env python3
"""GPU: time the bf16 GEMM routes of the fine-tune step on its shapes (through rpr_op_linear_bf16, conversions excluded by
subtraction of a K = 64 run of the same M, N is not attempted: the op converts A and W on every call, so the numbers below are
GEMM + two conversion passes; use them for A/B only). Usage: [RPR_BF16_W128=0] python tools/gemm_bf16_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E

ctx = E.Context.get(0)
shapes = [(8192, 768, 768, False, True), (8192, 768, 768, False, False), (8192, 2304, 768, False, False), (8192, 3072, 768, True, False),
          (8192, 768, 3072, False, True), (8192, 768, 2304, False, False), (4096, 768, 768, False, True), (4096, 3072, 768, True, False)]
for M, N, K, relu, resid in shapes:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda") if resid else None
    for _ in range(3): ctx.linear_bf16(A, W, R, relu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n): ctx.linear_bf16(A, W, R, relu)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"M={M} N={N} K={K} relu={relu} resid={resid}: {dt*1e6:8.1f} us per call (incl. conversions, alloc, sync)")
