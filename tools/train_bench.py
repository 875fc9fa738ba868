#!/usr/bin/env python3
"""Time the ranking fine-tune step (BASELINE config 5): t5-base dims, bz examples per step PER GPU, two teacher-forced
passes of L = 32 positions over queries of ~16 tokens, backward, gradient all-reduce (RCCL when launched on several
ranks), AdamW. Rank 0 prints one JSON line (examples/s of the whole job, max step time over the ranks).
Usage: python tools/train_bench.py [--bz 128] [--steps 5]
       python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/train_bench.py"""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
from ripor_amd.utils import synth

ap = argparse.ArgumentParser()
ap.add_argument("--bz", type=int, default=128)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--len", type=int, default=32, dest="L")
ap.add_argument("--precision", default="f16x2", choices=["f16x2", "f32", "bf16"])
args = ap.parse_args()
L, V, bz = args.L, 256, args.bz
dims = synth.t5_base_dims(L=L, V=V)
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
ctx = E.Context.get(local)
ctx.set_precision(args.precision)
model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
state = E.TrainState(model)
ids, mask = synth.make_queries(bz * world, vocab_size=dims.vocab_size, seed=5, mean_len=16, std_len=5, min_len=6, max_len=64)
ids, mask = ids[rank::world], mask[rank::world]          # every rank its own examples, one padded length for all
Lq = (ids.shape[1] + 7) // 8 * 8
ids = np.pad(ids, ((0, 0), (0, Lq - ids.shape[1]))); mask = np.pad(mask, ((0, 0), (0, Lq - mask.shape[1])))
codes = synth.make_codes(2 * bz, L, V, seed=5).astype(np.int64).reshape(2, bz, L).transpose(1, 0, 2).copy()
prefix = [L, 4, 8, 16][: {8: 2, 16: 3, 32: 4}[L]]
tp = torch.from_numpy(np.stack([synth.uniform_f32(f"tb/p{k}", (bz,), 30.0) for k in prefix]))
tn = torch.from_numpy(np.stack([synth.uniform_f32(f"tb/n{k}", (bz,), 30.0) for k in prefix]))
ids_t, mask_t, codes_t = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda(), torch.from_numpy(codes).cuda()


def step():   # backward with the bucketed gradient exchange overlapped (RPR_GRAD_OVERLAP=0: serial), clip, AdamW
    return E.train_step(model, state, ids_t, mask_t, codes_t, tp, tn, prefix, lr=1e-6)


first = step(); torch.cuda.synchronize()
if world > 1: dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    last = step()
torch.cuda.synchronize()
if world > 1: dist.barrier()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
if world > 1:
    t = torch.tensor([dt], device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt = float(t)
ctx.profile_reset(); ctx.profile_enable(True)
E.lngknp_backward(model, state, ids_t, mask_t, codes_t, tp, tn, prefix); torch.cuda.synchronize()
st = ctx.profile_get(); ctx.profile_enable(False)
flops_fwd = 2.0 * (bz * float(mask.sum(1).mean()) * (dims.num_layers * (4 * 768 * 768 + 2 * 768 * 3072) + 12 * 2 * 768 * 768)
                   + bz * 2 * L * 12 * (6 * 768 * 768 + 2 * 768 * 3072))
if rank == 0:
  print(json.dumps({"task": "lng_knp margin-MSE fine-tune step (forward + backward + gradient all-reduce + AdamW), t5-base dims",
                  "gemm_precision": args.precision, "n_gpus": world, "scaling": "weak", "bz_per_gpu": bz,
                  "bz": bz, "L": L, "enc_len_padded": int(Lq), "ms_per_step": dt * 1e3, "examples_per_s": world * bz / dt,
                  "loss_first": [float(x) for x in first], "loss_last": [float(x) for x in last],
                  "params": state.total, "approx_tflops": 3 * flops_fwd / dt / 1e12,
                  "backward_kernel_ms": {k: round(v["total_ms"], 3) for k, v in st.items()}}))
