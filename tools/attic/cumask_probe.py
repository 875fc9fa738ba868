"""Two half batches on two CU-masked HIP streams (each stream owns half of the CUs): does the MFMA/power-bound GEMM
phase of one overlap the HBM-bound attention phase of the other? (diagnostic; GPU box)
Usage: python tools/cumask_probe.py Q_per_stream pattern[lohi|evenodd|none] [graph|eager]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
from ripor_amd.utils import synth

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PAT = sys.argv[2] if len(sys.argv) > 2 else "lohi"
GRAPH = (sys.argv[3] == "graph") if len(sys.argv) > 3 else False
B, L, V, N = 10, 32, 256, int(os.environ.get("DOCS", 1_000_000))
dims = synth.t5_base_dims(L=L, V=V)
sd = synth.make_state_dict(dims, seed=1)
codes = synth.make_codes_fast(N, L, V, seed=1)
torch.cuda.init(); torch.zeros(1).cuda()
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(1 << b for b in range(32) if bits[w * 32 + b]) for w in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
NS = int(os.environ.get("NSTREAMS", 2))
if PAT == "lohi":
    masks = [[(i * NS) // 256 == k for i in range(256)] for k in range(NS)]
elif PAT == "mod":       # bit i -> stream i % NS
    masks = [[i % NS == k for i in range(256)] for k in range(NS)]
elif PAT == "evenodd":
    masks = [[i % 2 == 0 for i in range(256)], [i % 2 == 1 for i in range(256)]]
elif PAT == "xcd":      # if bit i lives on XCD i % 8: XCDs 0-3 vs 4-7
    masks = [[(i % 8) < 4 for i in range(256)], [(i % 8) >= 4 for i in range(256)]]
else:
    masks = None
streams = [masked_stream(m) for m in masks] if masks else [torch.cuda.Stream() for _ in range(NS)]
ctxs = [E.Context(0) for _ in range(NS)]
models = [E.DeviceModel(c, sd, dims) for c in ctxs]
tries = [E.DeviceTrie.from_codes(c, codes, V) for c in ctxs]
ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=3, mean_len=12, std_len=4, min_len=6, max_len=24)
ids = torch.from_numpy(ids).cuda(); mask = torch.from_numpy(mask).cuda()

OFFSET = float(os.environ.get("OFFSET_MS", 0)) * 1e-3   # host delay before stream 1's first launch (de-phasing)

def run(n, iters):
    torch.cuda.synchronize(); t0 = time.time()
    for it in range(iters):
        for i in range(n):
            if it == 0 and i == 1 and OFFSET > 0:
                time.sleep(OFFSET)
            with torch.cuda.stream(streams[i]):
                E.search(models[i], tries[i], ids, mask, B, L, use_graph=GRAPH)
    torch.cuda.synchronize()
    return n * iters * Q / (time.time() - t0)

for i in range(NS):
    with torch.cuda.stream(streams[i]):
        E.search(models[i], tries[i], ids, mask, B, L, use_graph=GRAPH)
torch.cuda.synchronize()
run(NS, 1)
print(f"pattern {PAT}, Q={Q} per stream, graph={GRAPH}, offset {OFFSET*1e3:.2f} ms: 1 masked stream {run(1, 4):8.1f} q/s; {NS} streams {run(NS, 6):8.1f} q/s", flush=True)
if os.environ.get("PLAIN"):   # the same searches on one ordinary stream with one context, bench.py style
    Q2 = int(os.environ["PLAIN"])
    ids2, mask2 = synth.make_queries(Q2, vocab_size=dims.vocab_size, seed=3, mean_len=12, std_len=4, min_len=6, max_len=24)
    ids2 = torch.from_numpy(ids2).cuda(); mask2 = torch.from_numpy(mask2).cuda()
    for _ in range(2): E.search(models[0], tries[0], ids2, mask2, B, L, use_graph=GRAPH)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(4): E.search(models[0], tries[0], ids2, mask2, B, L, use_graph=GRAPH)
    torch.cuda.synchronize()
    print(f"plain default stream, Q={Q2}: {4 * Q2 / (time.time() - t0):8.1f} q/s", flush=True)
