"""One forced-tail search on a mini model with L = 32 (the shape the second-generation attention kernels take), results
dumped to an .npz — run once per setting of RPR_TAIL_ATTN_GEN / RPR_ENC_ATTN_MFMA / RPR_STEP_CROSS_MFMA (read when the
library loads) by tests/test_gpu_attn_generations.py, which compares the dumps.
usage: python tools/attn_gen_dump.py OUT.npz [beams] [L]"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ripor_amd import engine as E  # noqa: E402
from ripor_amd.utils import synth  # noqa: E402


def main(out: str, B: int, L: int = 32) -> None:
    Q, V, N, seed = 37, 256, 20000, 11
    dims = synth.mini_dims(L=L, V=V, enc_layers=2, d_ff=256)
    sd = synth.make_state_dict(dims, seed=seed)
    codes = synth.make_codes(N, L, V, seed=seed)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=32)
    mask[3, 1] = 0            # a hole in a key mask
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    dumps = {}
    for name, depths in (("auto", None), ("fork3", [3]), ("fork5_7", [5, 7])):
        if depths is not None:
            ctx.set_fork_depths(depths)
        res = E.search_guarded(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L).result()
        torch.cuda.synchronize()
        dumps[name + "_tokens"] = res.tokens.cpu().numpy()
        dumps[name + "_scores"] = res.scores.cpu().numpy()
        dumps[name + "_forks"] = np.array([[f["depth"], f["forced"], f["left"]] for f in ctx.last_fork_stats()], dtype=np.int64).reshape(-1, 3)
    dumps["encoder_out"] = model.encode(torch.from_numpy(ids), torch.from_numpy(mask)).float().cpu().numpy()
    dumps["mask"] = mask
    np.savez(out, **dumps)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, int(sys.argv[3]) if len(sys.argv) > 3 else 32)
