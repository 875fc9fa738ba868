"""Diagnostic (GPU box): HIP gradients of rpr_lngknp_backward vs autograd through the CPU oracle, tensor by tensor."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_train import TrainGolden, _inputs, _train_model
from oracle import t5_ref, train_ref
name = sys.argv[1] if len(sys.argv) > 1 else "f4_mini_bz6_l32"
g = TrainGolden(name)
m = _train_model(g)
m.backward(**_inputs(g)); torch.cuda.synchronize()
hip = {k: v.detach().cpu().double().numpy() for k, v in m.train_state().named_grads().items()}
teacher = {k: g.z[k] for k in g.z.files if k.endswith("_scores") and "teacher" in k}
torch.set_num_threads(16)
_, total, og, gn = train_ref.train_step(t5_ref.T5Ref(g.state_dict, g.dims), g.z["input_ids"], g.z["attention_mask"],
                                        g.z["pos_doc_encoding"], g.z["neg_doc_encoding"], teacher)
rows = []
for k, v in hip.items():
    o = og[k].double().numpy().reshape(v.shape)
    rows.append((np.abs(v - o).max() / max(np.abs(o).max(), 1e-30), k, np.abs(o).max(), np.abs(v - o).max()))
rows.sort(reverse=True)
for r in rows[:12]:
    print("rel %.2e  %-60s max|ref| %.3e  max|err| %.3e" % r)
print("median rel err %.2e" % np.median([r[0] for r in rows]))
k = rows[0][1]
v, o = hip[k], og[k].double().numpy().reshape(hip[k].shape)
if v.ndim == 2:
    e = np.abs(v - o)
    r = e.max(axis=1)
    top = np.argsort(-r)[:5]
    print("worst tensor", k, "rows by max error:", [(int(i), float(r[i])) for i in top], " median row error", float(np.median(r)))
