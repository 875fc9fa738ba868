"""The path's only collective: gather every rank's ranked results on all ranks at the end of a run.

The reference has no collective on this path: each rank writes ``run_{rank}.json`` and a second
process merges them (t5_pretrainer/evaluate.py:130-132,503-520). Here the shards (equal size thanks
to the DistributedSampler-style wrap-around padding, ripor_amd/dataset/sharding.py) are exchanged
with one ``all_gather`` per tensor — RCCL over xGMI on GPUs ("nccl" backend), gloo in the CPU tests —
about 1.3 KB per query, so latency- not bandwidth-bound.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist


def all_gather_results(*tensors: torch.Tensor) -> Tuple[torch.Tensor, ...]:
    """Each input is this rank's shard (e.g. qids, tokens, scores, row_lo, row_hi — callers pass only what
    they consume) with identical leading size on every rank. Returns the concatenation over ranks
    (rank-major) of every tensor, in the order given."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tuple(tensors)
    world = dist.get_world_size()
    outs = []
    for t in tensors:
        t = t.contiguous()
        buf = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t)  # concatenated along dim 0, rank-major
        outs.append(buf)
    return tuple(outs)


def merge_by_qid(qids: torch.Tensor, tokens: torch.Tensor, scores: torch.Tensor, row_lo: torch.Tensor,
                 row_hi: torch.Tensor) -> Dict[int, dict]:
    """Duplicates created by the wrap-around padding collapse onto the same qid (they carry
    identical results), like the dict merge of the reference (evaluate.py:508-515)."""
    out: Dict[int, dict] = {}
    q = qids.cpu().tolist()
    tok, sc, lo, hi = tokens.cpu(), scores.cpu(), row_lo.cpu(), row_hi.cpu()
    for i, qid in enumerate(q):
        out[int(qid)] = {"tokens": tok[i], "scores": sc[i], "row_lo": lo[i], "row_hi": hi[i]}
    return out
