#!/usr/bin/env python3
"""Randomised check of the ranking fine-tune step on the GPU (diagnostic; `python tools/fuzz_train.py [n_cases] [seed]`):
mini-dims models, random batch shapes (bz 1..7, smtid length 8 / 16 / 32, codebook sizes 256 / 100 / 72, query lengths 6..30 with ragged padding), every
GEMM arithmetic (f16x2, bf16, exact f32). HIP gradients of rpr_lngknp_backward against torch autograd through the CPU oracle
(oracle/train_ref.py), tensor by tensor: split-precision and fp32 modes at rounding level (median tensor within 2e-5 of its scale,
every tensor's cosine >= 0.9999, global norm within 1e-4; a ReLU-boundary flip may move single rows more and is reported),
bf16 by cosine >= 0.98 on every tensor above the noise floor (small batches: 0.988 seen) and global norm within 3 %."""
import os, sys, random
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from test_gpu_train import _inputs, _train_model
from oracle import t5_ref, train_ref
from ripor_amd import engine as E
from ripor_amd.utils import synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
torch.set_num_threads(16)
ctx = E.Context.get(0)


class Z(dict):
    @property
    def files(self):
        return list(self.keys())


class G:
    pass


only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None
precs = os.environ.get("FUZZ_PREC", "f16x2,bf16,f32").split(",")
for case in range(n_cases):
    bz, L, seed = rng.randint(1, 7), rng.choice([8, 16, 32]), rng.randint(1, 10_000)
    V, max_len = rng.choice([256, 256, 100, 72]), rng.randint(8, 30)
    enc_layers, d_ff = rng.choice([1, 2]), rng.choice([128, 256])
    if only is not None and case != only:
        continue
    dims = synth.mini_dims(L=L, V=V, enc_layers=enc_layers, d_ff=d_ff)
    g = G()
    g.bz, g.L, g.V, g.dims, g.seed = bz, L, V, dims, seed
    g.state_dict = synth.make_state_dict(dims, seed=seed)
    ids, mask = synth.make_queries(bz, vocab_size=dims.vocab_size, seed=seed, min_len=6, max_len=max_len)
    codes = synth.make_codes(2 * bz, L, V, seed=seed).astype(np.int64)
    z = Z(input_ids=ids, attention_mask=mask, pos_doc_encoding=codes[:bz], neg_doc_encoding=codes[bz:])
    for pfx, name in [(L, "")] + [(k, f"smtid_{k}_") for k in (4, 8, 16) if k < L]:
        z[f"{name}teacher_pos_scores"] = synth.uniform_f32(f"ft/p{pfx}/{seed}", (bz,), 30.0)
        z[f"{name}teacher_neg_scores"] = synth.uniform_f32(f"ft/n{pfx}/{seed}", (bz,), 30.0)
    g.z = z
    teacher = {k: z[k] for k in z.files if k.endswith("_scores") and "teacher" in k}
    losses_ref, total_ref, og, gn_ref = train_ref.train_step(t5_ref.T5Ref(g.state_dict, g.dims), ids, mask, z["pos_doc_encoding"],
                                                              z["neg_doc_encoding"], teacher)
    line = [f"case {case:2d}: bz={bz} L={L} V={V} Lq={ids.shape[1]} enc={dims.num_layers} dff={dims.d_ff}"]
    for prec in precs:
        ctx.set_precision(prec)
        try:
            m = _train_model(g)
            out = m.backward(**_inputs(g))
            torch.cuda.synchronize()
            hip = {k: v.detach().cpu().double().numpy() for k, v in m.train_state().named_grads().items()}
        finally:
            ctx.set_precision("f16x2")
        worst, worst_k, min_cos = 0.0, "", 1.0
        gsq = 0.0
        rels = []
        for k, v in hip.items():
            o = og[k].double().numpy().reshape(v.shape)
            gsq += float((v ** 2).sum())
            scale = max(np.abs(o).max(), 1e-30)
            rel = np.abs(v - o).max() / scale
            rels.append(rel)
            if rel > worst: worst, worst_k = rel, k
            no, nv = np.linalg.norm(o), np.linalg.norm(v)
            if no > 1e-6 * gn_ref:
                min_cos = min(min_cos, float((v * o).sum() / max(no * nv, 1e-300)))
        gn = gsq ** 0.5
        if os.environ.get("FUZZ_VERBOSE"):
            import re as _re
            byblk = {}
            for (k, v), r in zip(hip.items(), rels):
                mm = _re.match(r"decoder\.block\.(\d+)\.", k)
                key = f"dec{int(mm.group(1)):02d}" if mm else ("enc" if k.startswith("encoder") else "other")
                byblk.setdefault(key, []).append(r)
            print(prec, "max rel error by block:", {k: f"{max(v):.1e}" for k, v in sorted(byblk.items())})
            rows = sorted(((np.abs(v - og[k].double().numpy().reshape(v.shape)).max() / max(np.abs(og[k].double().numpy()).max(), 1e-30), k)
                           for k, v in hip.items()), reverse=True)[:6]
            print(prec, [(f"{r:.1e}", k) for r, k in rows])
            k = rows[0][1]; e = np.abs(hip[k] - og[k].double().numpy().reshape(hip[k].shape))
            idx = np.argwhere(e > 0.1 * e.max())
            print("   entries above 10 % of the worst error:", len(idx), "rows", sorted(set(idx[:, 0].tolist()))[:12], "cols", sorted(set(idx[:, 1].tolist()))[:12] if e.ndim == 2 else "")
        if prec == "bf16":
            assert min_cos >= 0.98 and abs(gn - gn_ref) <= 0.03 * gn_ref, (case, prec, min_cos, gn, gn_ref, worst_k)
        else:
            # A pre-activation within rounding distance of 0 takes the other branch of ReLU'(x) in an implementation that sums
            # in another order: one FF unit's gradient row changes by a visible amount and the layers below follow at the
            # 1e-3 level (case 3 of seed 3, exact-fp32 mode: one row of block 8's wi gradient off by 2e-2, blocks 9-11 at 4e-6, everything
            # below at 2e-3). So: the global norm must agree, every tensor must point the same way, and a tensor away from
            # rounding level is only accepted when the worst one is a wi gradient.
            med = float(np.median(rels))
            flips = [(r, k) for (k, _), r in zip(hip.items(), rels) if "DenseReluDense.wi" in k and r > 2e-4]
            if worst > 2e-4:   # only a ReLU-boundary flip may do this: the wi gradient of the layer it happened in shows it,
                # one FF unit of one row gains or loses its whole gradient (seen up to 0.17 of the tensor's largest entry)
                assert flips and abs(gn - gn_ref) <= 2e-3 * gn_ref and min_cos >= 0.99, (case, prec, med, worst, worst_k, gn, gn_ref, min_cos)
                line.append(f"[{prec}: ReLU-boundary flip, {max(flips)[0]:.1e} in {max(flips)[1]}, worst tensor {worst:.1e} {worst_k}, median {med:.1e}]")
            else:
                assert med <= 2e-5 and abs(gn - gn_ref) <= 1e-4 * gn_ref and min_cos >= 0.9999, (case, prec, med, gn, gn_ref, min_cos)
        line.append(f"{prec}: worst {worst:.1e} median {float(np.median(rels)):.1e} cos {min_cos:.5f} |g| {gn / gn_ref:.5f}")
        del m
    print("; ".join(line), flush=True)
print(f"{n_cases} cases passed")
