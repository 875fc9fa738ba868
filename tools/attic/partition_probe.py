"""GEMM on a CU-masked stream of n CUs beside an HBM-bound streaming read on the other CUs: what does each keep of its
whole-chip rate? (proxy for lanes specialised by kernel type; GPU box)  Usage: python tools/partition_probe.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
ctx = E.Context.get(0)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if lo <= w * 32 + b < hi) for w in range(8)])
    st = ctypes.c_void_p(); assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)
M, N, K = 21760, 3072, 768
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
big = torch.randn(1 << 30, device="cuda")            # 4 GiB streaming read
def gemm_loop(n):
    for _ in range(n): ctx.linear(A, W, None)
def read_loop(n):
    for _ in range(n): big.sum()
def timed(fn_g, sg, ng, fn_r, sr, nr):
    torch.cuda.synchronize(); t0 = time.time(); eg = er = None
    ctx.profile_reset(); ctx.profile_enable(True)
    if fn_g:
        with torch.cuda.stream(sg):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record(); fn_g(ng); b.record(); eg = (a, b)
    if fn_r:
        with torch.cuda.stream(sr):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record(); fn_r(nr); b.record(); er = (a, b)
    torch.cuda.synchronize()
    ctx.profile_enable(False)
    out = []
    if eg:
        st = ctx.profile_get()["gemm"]      # the library's own events around the GEMM launches only
        out.append(f"GEMM {st['flops'] / (st['total_ms'] * 1e-3) / 1e12:6.1f} TF/s ({st['total_ms'] / max(1, st['launches']) * 1e3:6.1f} us)")
    if er: out.append(f"read {4.0 * big.numel() * nr / (er[0].elapsed_time(er[1]) * 1e-3) / 1e12:5.2f} TB/s")
    return "; ".join(out)
full = torch.cuda.Stream()
gemm_loop(3); read_loop(2); torch.cuda.synchronize()
print("whole chip, alone:   ", timed(gemm_loop, full, 40, None, None, 0), "|", timed(None, None, 0, read_loop, full, 8))
for g_cus in (224, 192, 160, 128):
    sg, sr = masked_stream(0, g_cus), masked_stream(g_cus, 256)
    with torch.cuda.stream(sg): gemm_loop(2)
    with torch.cuda.stream(sr): read_loop(1)
    print(f"GEMM on {g_cus} CUs alone: ", timed(gemm_loop, sg, 40, None, None, 0), f"| read on {256 - g_cus} CUs alone:", timed(None, None, 0, read_loop, sr, 8))
    print(f"   side by side:        ", timed(gemm_loop, sg, 40, read_loop, sr, 8))
