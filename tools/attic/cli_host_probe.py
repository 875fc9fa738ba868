"""How much of the CLI's retrieval loop (evaluate.constrained_decode_doc) is host work between searches? t5-base dims,
1 M docs, pre-tokenised batches of the automatic size. (diagnostic; GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import evaluate as ev
from ripor_amd.modeling.t5_generative_retriever import T5forDocIDConfig, T5ForDocIDGeneration
from ripor_amd.tasks.generation import PrefixConstrainLogitProcessorFastSparse
from ripor_amd.utils import synth

L, V, N, NQ = 32, 256, 1_000_000, 6980
dims = synth.t5_base_dims(L=L, V=V)
cfg = T5forDocIDConfig.from_dims(dims)
model = T5ForDocIDGeneration(cfg, synth.make_state_dict(dims)).to(0)
model.config.decoding = True
codes = synth.make_codes_fast(N, L, V)
proc = PrefixConstrainLogitProcessorFastSparse.from_codes(codes, V)
table = ev.DocidTable([str(i) for i in range(N)])
ids, mask = synth.make_queries(NQ, vocab_size=dims.vocab_size)
qbs = ev.search_batch_size(cfg, 1, 10, L, -1, 0)
print("queries per search call:", qbs)

def loader():
    for s in range(0, NQ, qbs):
        yield {"id": torch.arange(s, min(NQ, s + qbs)), "input_ids": torch.from_numpy(ids[s:s + qbs]),
               "attention_mask": torch.from_numpy(mask[s:s + qbs])}

os.makedirs("/tmp/cli_probe", exist_ok=True)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    run = ev.constrained_decode_doc(model, loader(), proc, table, L, device=0, out_dir="/tmp/cli_probe", local_rank=0, topk=10)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"pass {rep}: {NQ} queries in {dt:.2f}s = {NQ / dt:.0f} queries/s end to end ({len(run)} qids, run.json written)")
