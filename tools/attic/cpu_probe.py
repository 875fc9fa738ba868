"""Probe: how does the oracle 'port' scale with host threads on the GPU box? (diagnostic only)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import beam_ref, t5_ref
from ripor_amd.utils import synth
threads = int(sys.argv[1]); L = int(sys.argv[2]); nq = int(sys.argv[3])
torch.set_num_threads(threads)
dims = synth.t5_base_dims(L=32)
sd = synth.make_state_dict(dims)
codes = synth.make_codes(10000, 32, 256)
t0 = time.time()
pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), 256)
print("trie dicts", time.time() - t0, flush=True)
model = t5_ref.T5Ref(sd, dims)
ids, mask = synth.make_queries(nq, vocab_size=dims.vocab_size, seed=77)
t0 = time.time()
x = torch.randn(1320, 768); w = torch.randn(3072, 768)
for _ in range(10): y = x @ w.t()
print("10 matmuls", time.time() - t0, flush=True)
t0 = time.time()
beam_ref.beam_search_ref(model, pm, ids, mask, 10, L)
print(f"threads={threads} L={L} nq={nq}: {time.time() - t0:.2f}s", flush=True)
