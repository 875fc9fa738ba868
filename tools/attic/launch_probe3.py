"""Probe 3 (GPU box): per-node cost of a hipGraph made of THIS library's kernels (through the C ABI, captured by torch):
one small kernel repeated, two kernels alternating, and what the single-query search costs eagerly vs graph-replayed."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from ripor_amd import engine as E  # noqa: E402

ctx = E.Context.get(0)
lib = ctx.lib
dev = torch.device("cuda:0")
x = torch.randn(16, 768, device=dev)
w = torch.ones(768, device=dev)
out = torch.empty_like(x)
W = torch.randn(768, 768, device=dev) * 0.02
y = torch.empty(16, 768, device=dev)
ctx.set_precision("f32")


def chain(name, body, n=1000):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        sp = C.c_void_p(s.cuda_stream)
        for i in range(3):
            body(i, sp)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                body(i, sp)
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f"{name:60s} graph {(time.perf_counter() - t0) / 3 / n * 1e6:6.2f} us/kernel", flush=True)


def rms(i, sp):
    assert lib.rpr_op_rmsnorm(ctx.handle, C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()), 16, 768,
                              C.c_float(1e-6), sp) == 0


def lin(i, sp):
    assert lib.rpr_op_linear(ctx.handle, C.c_void_p(out.data_ptr()), C.c_void_p(W.data_ptr()), None, C.c_void_p(y.data_ptr()), 16, 768, 768, 0, sp) == 0


chain("library rmsnorm_kernel (16 rows) repeated", rms)
chain("library fp32 GEMM 16x768x768 repeated", lin)
chain("rmsnorm / fp32 GEMM alternating", lambda i, sp: (rms if i & 1 else lin)(i, sp))
