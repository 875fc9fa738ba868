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


# ---- the CLI's gathered run.json (constrained_decode_doc(gather=True)) on two gloo ranks ---------------------------
class _FakeTrie:
    def __init__(self, n):
        import numpy as np
        self.perm = np.arange(n)[::-1].copy()          # sorted row -> docid index


class _FakeProcessor:
    def __init__(self, n):
        self._t = _FakeTrie(n)

    def trie(self, device):
        return self._t


class _Out:
    pass


def _fake_generate(model, processor, input_ids=None, num_beams=None, **kw):
    """Stand-in for the HIP search: results are a function of the first token (= the qid here)."""
    q = input_ids[:, 0].to(torch.int64)
    o = _Out()
    lo = (q[:, None] * 3 + torch.arange(num_beams)[None, :]) % 50
    o.row_lo, o.row_hi = lo.reshape(-1), (lo + 1 + (q[:, None] % 2)).reshape(-1)   # one or two docs per smtid
    o.sequences_scores = ((q[:, None].float() + 1.0) / (torch.arange(num_beams)[None, :].float() + 2.0)).reshape(-1)
    o.sequences = torch.zeros((q.numel() * num_beams, 5), dtype=torch.long)
    return o


def _cli_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ripor_amd import evaluate as ev
        ev.generate_for_constrained_prefix_beam_search = _fake_generate
        idx = shard_indices(N_QUERIES, world, rank)
        batches = [{"input_ids": torch.tensor([[200 + i, 1] for i in idx[s:s + 2]]),
                    "attention_mask": torch.ones((len(idx[s:s + 2]), 2), dtype=torch.long),
                    "id": torch.tensor([200 + i for i in idx[s:s + 2]])} for s in range(0, len(idx), 2)]
        table = ev.DocidTable([f"D{j}" for j in range(64)])
        ev.constrained_decode_doc(None, batches, _FakeProcessor(64), table, 4, "cpu", out_dir, rank, topk=B, gather=True)
    finally:
        dist.destroy_process_group()


def test_cli_gathered_run_json_equals_single_process(tmp_path):
    import json
    from ripor_amd import evaluate as ev
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    multi = tmp_path / "multi"
    multi.mkdir()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, str(multi))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(os.listdir(multi)) == ["run.json"]          # no per-rank parts
    merged = ev.merge_runs(str(multi))                         # the ..._2 step keeps the gathered file
    # single process, file path
    single_dir = tmp_path / "single"
    single_dir.mkdir()
    ev.generate_for_constrained_prefix_beam_search = _fake_generate
    batches = [{"input_ids": torch.tensor([[200 + i, 1]]), "attention_mask": torch.ones((1, 2), dtype=torch.long),
                "id": torch.tensor([200 + i])} for i in range(N_QUERIES)]
    table = ev.DocidTable([f"D{j}" for j in range(64)])
    ev.constrained_decode_doc(None, batches, _FakeProcessor(64), table, 4, "cpu", str(single_dir), 0, topk=B)
    single = ev.merge_runs(str(single_dir))
    assert json.loads(json.dumps(merged)) == json.loads(json.dumps(single)) and len(single) == N_QUERIES
