"""Does running two in-flight query batches on two HIP streams overlap the MFMA-bound GEMMs of one with the
HBM-bound attention of the other? (diagnostic; GPU box). Two rpr contexts = two workspaces + graphs."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
from ripor_amd.utils import synth

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NCTX = int(sys.argv[2]) if len(sys.argv) > 2 else 2
GRAPH = (sys.argv[3] != 'eager') if len(sys.argv) > 3 else True
B, L, V, N = 10, 32, 256, 1_000_000
dims = synth.t5_base_dims(L=L, V=V)
sd = synth.make_state_dict(dims, seed=1)
codes = synth.make_codes_fast(N, L, V, seed=1)
ctxs = [E.Context(0) for _ in range(NCTX)]
models = [E.DeviceModel(c, sd, dims) for c in ctxs]
tries = [E.DeviceTrie.from_codes(c, codes, V) for c in ctxs]
ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=3, mean_len=12, std_len=4, min_len=6, max_len=24)
ids = torch.from_numpy(ids).cuda(); mask = torch.from_numpy(mask).cuda()
streams = [torch.cuda.Stream() for _ in range(NCTX)]

def run(n_ctx, iters):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters):
        for i in range(n_ctx):
            with torch.cuda.stream(streams[i]):
                E.search(models[i], tries[i], ids, mask, B, L, use_graph=GRAPH)
    torch.cuda.synchronize()
    return n_ctx * iters * Q / (time.time() - t0)

for i in range(NCTX):
    with torch.cuda.stream(streams[i]):
        E.search(models[i], tries[i], ids, mask, B, L, use_graph=GRAPH)
torch.cuda.synchronize()
print(f"Q={Q} per batch")
print(f"1 stream : {run(1, 3):8.1f} q/s")
for n in range(2, NCTX + 1):
    print(f"{n} streams: {run(n, 3):8.1f} q/s", flush=True)
