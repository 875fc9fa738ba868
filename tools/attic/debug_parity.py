"""Diagnostic (GPU box): repeated searches of one golden fixture in every launch mode, status flags, and a per-step
logit comparison of the f16x2 path against the exact-fp32 path. Usage: python tools/debug_parity.py [fixture]"""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import conftest
from ripor_amd import engine as E

name = sys.argv[1] if len(sys.argv) > 1 else "g1_mini_b4_l8"
g = conftest.Golden(name)
ctx = E.Context.get(0)
model = E.DeviceModel(ctx, g.state_dict, g.dims)
trie = E.DeviceTrie.from_codes(ctx, g.codes, g.V)
ids, mask = torch.from_numpy(g.input_ids), torch.from_numpy(g.attention_mask)
exp = g.sequences.reshape(g.Q, g.B, g.L + 1)[:, :, 1:]


def run(tag, **kw):
    ctx.status(clear=True)
    r = E.search(model, trie, ids, mask, g.B, g.L, apply_log_softmax_for_scores=g.log_softmax, **kw)
    torch.cuda.synchronize()
    st = ctx.status(clear=True)
    tok = r.tokens.cpu().numpy()
    bad = [q for q in range(g.Q) if not (tok[q] == exp[q]).all()]
    err = np.abs(r.scores.cpu().numpy() - g.sequences_scores.reshape(g.Q, g.B)).max()
    print(f"{tag:28s} status={st} queries with wrong tokens={bad} max score err={err:.2e}", flush=True)
    return r


print("model f32_only:", model.f32_only)
for i in range(3):
    run(f"graph #{i}")
run("eager", use_graph=False)
run("graph again")
a = run("taps f16x2", taps=True)
ctx.set_precision("f32")
b = run("taps f32", taps=True)
run("graph f32")
ctx.set_precision("f16x2")
la, lb = a.taps["step_logits"].cpu().numpy(), b.taps["step_logits"].cpu().numpy()
ea, eb = a.taps["encoder_out"].cpu().numpy(), b.taps["encoder_out"].cpu().numpy()
print("encoder_out max |f16x2 - f32|:", np.abs(ea - eb).max(), " vs golden:", np.abs(ea - g.z["encoder_out"]).max())
for t in range(g.L):
    print(f"step {t}: max |logit f16x2 - f32| = {np.abs(la[t] - lb[t]).max():.3e}")
    if not (a.taps["step_tokens"][t].cpu().numpy() == b.taps["step_tokens"][t].cpu().numpy()).all():
        print("   selections diverge here"); break
