"""not-gpu: the N>1 path on CPU with world_size 2 over gloo — DistributedSampler-style sharding of
the queries, one all_gather of the ranked results, merge by qid == the single-process result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ripor_amd.dataset.sharding import shard_indices
from ripor_amd.dist_gather import all_gather_results, merge_by_qid

N_QUERIES, B, L = 7, 3, 4


def _fake_search(qids: torch.Tensor):
    """Deterministic stand-in for rpr_search (no GPU here): results are a function of the qid."""
    q = qids.to(torch.int64)
    tokens = (q[:, None, None] * 7 + torch.arange(B)[None, :, None] * 3 + torch.arange(L)[None, None, :]) % 256
    scores = (q[:, None].float() + 1.0) / (torch.arange(B)[None, :].float() + 1.0)
    lo = q[:, None] * 10 + torch.arange(B)[None, :]
    return tokens.to(torch.int32), scores, lo, lo + 1


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qids = torch.tensor([100 + i for i in shard_indices(N_QUERIES, world, rank)])
        tok, sc, lo, hi = _fake_search(qids)
        merged = merge_by_qid(*all_gather_results(qids, tok, sc, lo, hi))
        if rank == 0:
            out.put({k: {n: v.tolist() for n, v in d.items()} for k, d in merged.items()})
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    qids = torch.tensor([100 + i for i in range(N_QUERIES)])
    tok, sc, lo, hi = _fake_search(qids)
    single = merge_by_qid(qids, tok, sc, lo, hi)
    assert set(got) == set(single) and len(got) == N_QUERIES  # wrap-around duplicate collapsed
    for k, d in single.items():
        for n, v in d.items():
            assert got[k][n] == v.tolist(), (k, n)


def test_single_process_gather_is_identity():
    qids = torch.arange(4)
    tok, sc, lo, hi = _fake_search(qids)
    out = all_gather_results(qids, tok, sc, lo, hi)
    assert all(torch.equal(a, b) for a, b in zip(out, (qids, tok, sc, lo, hi)))
