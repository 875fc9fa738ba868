"""Where do the device gradients of a synthetic model variant differ from oracle autograd? Per-tensor error and, for the
worst matrices, the rows that carry it (a single FF unit = a ReLU gate that fell on different sides of 0). GPU box.
Usage: python tools/debug_train_variant.py [shared|scaleup]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import t5_ref, train_ref
from ripor_amd import engine as E
from ripor_amd.utils import synth
variant = sys.argv[1] if len(sys.argv) > 1 else "shared"
SEED = int(sys.argv[2]) if len(sys.argv) > 2 else 91
torch.set_num_threads(1)
kw = dict(shared=dict(shared_output_input_embeds=True), scaleup=dict(scaleup_output_hidden=True))[variant]
L, bz = 8, 3
dims = synth.mini_dims(L=L, V=256, **kw)
sd = synth.make_state_dict(dims, seed=SEED)
ids, mask = synth.make_queries(bz, vocab_size=dims.vocab_size, seed=17, mean_len=9, std_len=3, min_len=5, max_len=13)
codes = synth.make_codes(2 * bz, L, 256, seed=23).astype(np.int64)
pos, neg = codes[:bz], codes[bz:]
prefix = train_ref.PREFIX_LENS[L]
name = {"shared": "shared_codebooks", "scaleup": "scaleup_hidden"}[variant]
teacher = {}
for k in prefix:
    key = "" if k == L else train_ref.TEACHER_KEYS[k]
    teacher[key + "teacher_pos_scores"] = synth.uniform_f32(f"var/{name}/p{k}", (bz,), 30.0)
    teacher[key + "teacher_neg_scores"] = synth.uniform_f32(f"var/{name}/n{k}", (bz,), 30.0)
ref_losses, _, og, gn = train_ref.train_step(t5_ref.T5Ref(sd, dims), ids, mask, pos, neg, teacher)
ctx = E.Context.get(0); ctx.set_precision("f32")
model = E.DeviceModel(ctx, sd, dims); state = E.TrainState(model)
tp = torch.from_numpy(np.stack([teacher[("" if k == L else train_ref.TEACHER_KEYS[k]) + "teacher_pos_scores"] for k in prefix]))
tn = torch.from_numpy(np.stack([teacher[("" if k == L else train_ref.TEACHER_KEYS[k]) + "teacher_neg_scores"] for k in prefix]))
losses = E.lngknp_backward(model, state, torch.from_numpy(ids), torch.from_numpy(mask), torch.from_numpy(np.stack([pos, neg], 1)), tp, tn, prefix)
torch.cuda.synchronize()
print("losses", [float(x) for x in losses], [float(ref_losses[n]) for n in train_ref.LOSS_NAMES[L]])
grads = {k: v.detach().cpu().double().numpy() for k, v in state.named_grads().items()}
errs = []
for k, v in grads.items():
    o = og[k].double().numpy().reshape(v.shape)
    errs.append((np.abs(v - o).max() / max(np.abs(o).max(), 1e-30), k))
errs.sort(reverse=True)
print("SEED", SEED, variant, "worst", f"{errs[0][0]:.3e}", errs[0][1])
for e, k in errs[:4]:
    print(f"{e:.3e} {k}")
for e, k in errs[:3]:
    v = grads[k]; o = og[k].double().numpy().reshape(v.shape)
    if v.ndim == 2:
        re = np.abs(v - o).max(1) / np.abs(o).max(); ce = np.abs(v - o).max(0) / np.abs(o).max()
        print(k, "rows >1e-4:", np.nonzero(re > 1e-4)[0][:10], re[re > 1e-4][:10].round(5), "| cols >1e-4:", int((ce > 1e-4).sum()), "of", ce.size)
