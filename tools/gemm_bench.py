"""Micro-benchmark of rpr_op_linear on the decoder/encoder GEMM shapes of the bench config
(diagnostic; run on the GPU box). Usage: python tools/gemm_bench.py [M ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E

Ms = [int(x) for x in sys.argv[1:]] or [5120]
ctx = E.Context.get(0)
shapes = [(768, 768, True), (2304, 768, False), (3072, 768, False), (768, 3072, True), (256, 768, False)]
for M in Ms:
  tot_t = tot_f = 0.0
  for N, K, resid in shapes:
      A = torch.randn(M, K, device="cuda")
      W = torch.randn(N, K, device="cuda") * K ** -0.5
      if os.environ.get("GEMM_ZERO"):   # data-dependent power: all-zero operands toggle no multiplier bits
          A.zero_(); W.zero_()
      R = torch.randn(M, N, device="cuda") if resid else None
      for _ in range(3):
          ctx.linear(A, W, R)
      torch.cuda.synchronize()
      # time only the GEMM launch (library hipEvents), not the test hook's operand splitting
      ctx.profile_reset(); ctx.profile_enable(True)
      n = 20
      for _ in range(n):
          ctx.linear(A, W, R)
      torch.cuda.synchronize()
      pg = ctx.profile_get(); ctx.profile_enable(False)
      st = pg["gemm"] if pg["gemm"]["launches"] else pg["gemm_small"]
      us = st["total_ms"] / st["launches"] * 1e3
      fl = 2.0 * M * N * K
      ref = A.double() @ W.double().t() + (R.double() if resid else 0)
      err = (ctx.linear(A, W, R).double() - ref).abs().max().item()
      print(f"M={M} N={N} K={K} resid={resid}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF  maxerr {err:.2e}", flush=True)
      w = {768: 3 if K == 768 else 1, 2304: 1, 3072: 1, 256: 1 / 12}[N]  # per-layer multiplicity in a decoder step
      tot_t += us * w; tot_f += fl * w
  print(f"M={M} decoder-layer weighted: {tot_t:.1f} us per layer, {tot_f / tot_t / 1e6:.1f} TF")
