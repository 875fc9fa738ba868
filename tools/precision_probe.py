"""f16x2 split-precision vs exact-fp32 GEMM mode on the t5-base golden model: per-step logit differences
(diagnostic; GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import Golden
from ripor_amd import engine as E
from ripor_amd.utils import synth

g = Golden("g2_base_b10_l32")
ctx = E.Context.get(0)
model = E.DeviceModel(ctx, g.state_dict, g.dims)
trie = E.DeviceTrie.from_codes(ctx, g.codes, g.V)
ids, mask = torch.from_numpy(g.input_ids), torch.from_numpy(g.attention_mask)
out = {}
for mode in ("f32", "f16x2"):
    ctx.set_precision(mode)
    r = E.search(model, trie, ids, mask, g.B, g.L, taps=True)
    torch.cuda.synchronize()
    out[mode] = (r.taps["step_logits"].cpu().numpy(), r.scores.cpu().numpy(), r.tokens.cpu().numpy())
ctx.set_precision("f16x2")
same = np.array_equal(out["f32"][2], out["f16x2"][2])
# compare logits only while the beams agree (they do on this model)
d = np.abs(out["f32"][0] - out["f16x2"][0])
print(f"tokens identical: {same}; logit |diff| max {d.max():.3e} mean {d.mean():.3e}; "
      f"score |diff| max {np.abs(out['f32'][1] - out['f16x2'][1]).max():.3e}; logit magnitude max {np.abs(out['f32'][0]).max():.1f}")
