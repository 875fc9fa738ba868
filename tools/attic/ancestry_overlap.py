"""How many DISTINCT (position, slot) K/V rows do the B beams of a query read at each decode step, against the
B * (t + 1) rows the per-beam kernel fetches? (diagnostic for a shared-ancestor self-attention; GPU box)
Usage: python tools/ancestry_overlap.py [Q] [docs]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
from ripor_amd.utils import synth

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 128
docs = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
B, L = 10, 32
dims = synth.t5_base_dims(L=L); V = dims.decoder_vocab_sizes[0]
ctx = E.Context.get(0)
model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
trie = E.DeviceTrie.from_codes(ctx, synth.make_codes_fast(docs, L, V), V)
ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size)
res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L, taps=True)
torch.cuda.synchronize()
parent = res.taps["step_parent"].cpu().numpy()       # [L, Q, B]: beam slot at step t-1 that slot b of step t extends
# anc[t][q][b][p] = slot at position p of the ancestry of beam b after step t (p <= t; position t is b itself)
tot_rows = tot_distinct = 0
per_t = []
anc = np.zeros((Q, B, L), dtype=np.int64)
for t in range(L):
    if t > 0:
        par = parent[t]                                # [Q, B]
        anc = np.take_along_axis(anc, par[:, :, None].repeat(L, 2), axis=1)
    anc[:, :, t] = np.arange(B)[None, :]
    # rows the NEXT step's attention reads: positions 0..t of every beam
    distinct = 0
    for p in range(t + 1):
        s = np.sort(anc[:, :, p], axis=1)
        distinct += (1 + (np.diff(s, axis=1) != 0).sum(1)).sum()
    rows = Q * B * (t + 1)
    per_t.append(distinct / rows)
    tot_rows += rows; tot_distinct += distinct
print("distinct / fetched per step:", " ".join(f"{x:.2f}" for x in per_t))
print(f"whole search: {tot_distinct / tot_rows:.3f} of the per-beam rows are distinct")
