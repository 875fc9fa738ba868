"""not-gpu: the N>1 path on CPU with world_size 2 over gloo — DistributedSampler-style sharding of
the queries, one all_gather of the ranked results, merge by qid == the single-process result."""
import os
import socket

import pytest

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


# ---- bench.py's own launcher: `python bench.py --gpus N` must become N ranks ----------------------------------------
def test_bench_gpus_flag_launches_n_ranks():
    """The driver may call `python bench.py --gpus 8` without torchrun: the script then re-executes itself under
    torch.distributed.run. RPR_BENCH_LAUNCH_ONLY stops after the rendezvous (no GPU here), gloo stands in for RCCL."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RPR_BENCH_LAUNCH_ONLY="1", RPR_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_world_size"] == 2 and out["ranks_seen"] == [0, 1]
    # a launcher that provides a different world size than --gpus is refused instead of mislabelled
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"], env=env2, capture_output=True,
                        text=True, timeout=120)
    assert r2.returncode != 0 and "refusing" in (r2.stderr + r2.stdout)


def test_bench_launch_command():
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.launch_command(1, {}, ["bench.py"]) is None
    assert bench.launch_command(8, {"WORLD_SIZE": "8"}, ["bench.py", "--gpus", "8"]) is None
    cmd = bench.launch_command(8, {}, ["bench.py", "--gpus", "8", "--steps", "3"])
    assert "--nproc-per-node=8" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "3"]


# ---- data-parallel gradient exchange of the training step (config 5) on two gloo ranks -------------------------------------
def _grad_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ripor_amd import engine as E

        class _State:   # the flat gradient buffer of a replica (TrainState.grads), on the CPU here
            pass
        st = _State()
        n = 1000
        st.grads = (torch.arange(n, dtype=torch.float32) + 1.0) * (rank + 1)
        E.allreduce_grads(st, bucket_elems=256)      # 4 chunks
        if rank == 0:
            out.put(st.grads.tolist())
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_averages_over_ranks():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = ((torch.arange(1000, dtype=torch.float32) + 1.0) * 1.5).tolist()    # mean of x and 2x
    assert got == expect


def _bucket_worker(rank, world, port, out, mode="allreduce"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ripor_amd import engine as E
        n = 1000
        grads = (torch.arange(n, dtype=torch.float32) + 1.0) * (rank + 1)
        ex = E.GradExchange(grads, mode=mode)
        assert ex.active and ex.world == world and ex.stream is None    # CPU tensors: no communication stream
        # the order rpr_lngknp_backward_buckets hands the buckets over: layers last to first, then the front of the buffer
        for off, cnt in [(700, 300), (400, 300), (250, 150), (0, 250)]:
            ex.on_bucket(off, cnt)
        ex.finish()
        if rank == 0:
            out.put(grads.tolist())
        # a bucket list that does not cover the buffer must be refused (an element would stay un-reduced)
        ex.on_bucket(0, 10)
        try:
            ex.finish()
            ok = False
        except AssertionError:
            ok = True
        if rank == 0:
            out.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("allreduce", 2), ("mesh", 2), ("mesh", 3)])
def test_bucketed_gradient_exchange_averages_over_ranks(mode, world):
    """engine.GradExchange (the overlapped exchange of the training step: per bucket, as the backward hands it over, either
    one asynchronous all-reduce or the mesh form — every rank receives its 1/W shard from all peers (all_to_all_single),
    sums it, the shards are gathered back; buckets that do not divide by W are padded — then DDP's averaging) on gloo ranks."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, out, mode)) for r in range(world)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    refused = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mean_factor = sum(range(1, world + 1)) / world
    assert got == ((torch.arange(1000, dtype=torch.float32) + 1.0) * mean_factor).tolist()
    assert refused
