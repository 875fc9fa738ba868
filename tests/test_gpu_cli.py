"""-m gpu: the reference's shell pipeline for this path, end to end on synthetic data
(full_scripts/full_evaluate_t5seq_aq_encoder.sh:176-205): build_list_smtid_to_nextids ->
`python -m t5_pretrainer.evaluate --task=t5seq_aq_retrieve_docids` -> `..._2` (merge + evaluate), with a
checkpoint directory, docid_to_smtid.json, raw.tsv queries, a SentencePiece tokenizer trained offline and
synthetic qrels. The run.json written by the CLI must equal what the Python API returns for the same
tokenised queries."""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_world(root):
    import sentencepiece as spm
    from ripor_amd.modeling.t5_generative_retriever import T5SeqAQEncoder
    from ripor_amd.utils import synth
    L, V, N = 8, 256, 600
    ckpt = os.path.join(root, "checkpoint")
    os.makedirs(ckpt)
    random.seed(0)
    words = ["what", "is", "the", "how", "to", "of", "in", "a", "best", "price", "weather", "define"] + [f"w{i}" for i in range(200)]
    corpus = os.path.join(root, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(2000):
            f.write(" ".join(random.choice(words) for _ in range(random.randint(3, 12))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(ckpt, "spiece"), vocab_size=256,
                                   model_type="unigram", pad_id=0, eos_id=1, unk_id=2, bos_id=-1, pad_piece="<pad>",
                                   eos_piece="</s>", unk_piece="<unk>", hard_vocab_limit=False, minloglevel=2)
    json.dump({"tokenizer_class": "T5Tokenizer", "extra_ids": 0, "model_max_length": 512},
              open(os.path.join(ckpt, "tokenizer_config.json"), "w"))
    dims = synth.mini_dims(L=L, V=V, enc_layers=2, d_ff=128, vocab_size=512)
    T5SeqAQEncoder.from_synthetic(dims, seed=77).save_pretrained(ckpt)
    codes = synth.make_codes(N, L, V, seed=77)
    data = os.path.join(root, "msmarco_toyset")           # "msmarco" in the path -> dataset name MSMARCO
    os.makedirs(os.path.join(data, "aq_smtid"))
    os.makedirs(os.path.join(data, "dev_queries"))
    d2s_path = os.path.join(data, "aq_smtid", "docid_to_smtid.json")
    json.dump({str(100 + i): [-1] + [int(x) for x in row] for i, row in enumerate(codes)}, open(d2s_path, "w"))
    queries = {str(900 + i): " ".join(random.choice(words) for _ in range(random.randint(3, 9))) for i in range(11)}
    with open(os.path.join(data, "dev_queries", "raw.tsv"), "w") as f:
        for qid, text in queries.items():
            f.write(f"{qid}\t{text}\n")
    return ckpt, d2s_path, os.path.join(data, "dev_queries"), codes, queries, dims


def _run(args, env=None):
    e = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    e.update(env or {})
    p = subprocess.run([sys.executable] + args, cwd=REPO, env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-4000:]
    return p.stdout


def test_shell_pipeline_end_to_end(tmp_path):
    ckpt, d2s_path, qdir, codes, queries, dims = _make_world(str(tmp_path))
    out_dir = os.path.join(str(tmp_path), "out")
    B, L = 5, 8
    # the preprocess step is host only: it must work with every GPU hidden
    _run(["-m", "t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids", "--docid_to_smtid_path", d2s_path],
         env={"HIP_VISIBLE_DEVICES": "", "ROCR_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
    assert os.path.exists(os.path.join(os.path.dirname(d2s_path), "list_smtid_to_nextids.rprtrie"))
    # one process per GPU through torch.distributed.run, like the script's torch.distributed.launch
    out_txt = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
          "--master-port", "29533", "-m", "t5_pretrainer.evaluate", f"--pretrained_path={ckpt}", f"--out_dir={out_dir}",
          "--task=t5seq_aq_retrieve_docids", f"--docid_to_smtid_path={d2s_path}",
          "--q_collection_paths=" + json.dumps([qdir]), "--batch_size=4", f"--max_new_token_for_docid={L}", f"--topk={B}"])
    assert "trie cache:" in out_txt, "the retrieval task did not take the binary trie cache written by the preprocess step"
    run_part = os.path.join(out_dir, "MSMARCO", "run_0.json")
    assert os.path.exists(run_part)
    run = json.load(open(run_part))
    assert set(run) == set(queries)
    # synthetic qrels: the top document of every query is relevant -> MRR@10 must be 1.0
    qrel = {qid: {max(docs, key=docs.get): 1} for qid, docs in run.items()}
    qrel_path = os.path.join(str(tmp_path), "msmarco_toyset", "dev_qrel.json")
    json.dump(qrel, open(qrel_path, "w"))
    _run(["-m", "t5_pretrainer.evaluate", "--task=t5seq_aq_retrieve_docids_2", f"--out_dir={out_dir}",
          "--q_collection_paths=" + json.dumps([qdir]), "--eval_qrel_path=" + json.dumps([qrel_path])])
    merged = json.load(open(os.path.join(out_dir, "MSMARCO", "run.json")))
    assert merged == run and not os.path.exists(run_part)
    perf = json.load(open(os.path.join(out_dir, "MSMARCO", "perf.json")))
    assert perf["mrr_10"] == 1.0 and "recall_10" in perf

    # the CLI result == the Python API on the same tokenised queries
    from transformers import AutoTokenizer
    from ripor_amd import engine as E
    from ripor_amd.evaluate import QueryCollection, query_batches
    from ripor_amd.modeling.t5_generative_retriever import T5SeqAQEncoder
    tok = AutoTokenizer.from_pretrained(ckpt)
    coll = QueryCollection(qdir)
    model = T5SeqAQEncoder.from_pretrained(ckpt).to(0)
    ctx = E.Context.get(0)
    trie = E.DeviceTrie.from_codes(ctx, codes, 256)
    for batch in query_batches(coll, tok, list(range(len(coll))), 4, 256):
        res = E.search(model.base_model.engine_model(), trie, batch["input_ids"], batch["attention_mask"], B, L)
        torch.cuda.synchronize()
        lo, hi, sc = res.row_lo.cpu().numpy(), res.row_hi.cpu().numpy(), res.scores.cpu().numpy()
        for qi, qid in enumerate(batch["id"].tolist()):
            expect = {}
            for b in range(B):
                for row in trie.perm[lo[qi, b]:hi[qi, b]]:
                    expect[str(100 + int(row))] = float(sc[qi, b]) * L
            got = run[str(qid)]
            assert set(got) == set(expect)
            for d in expect:
                assert abs(got[d] - expect[d]) < 1e-3


def test_bench_two_ranks_on_one_gpu_through_gloo():
    """The N > 1 path of bench.py end to end on the one GPU of the test box: two ranks (gloo instead of RCCL, both on
    cuda:0 through the RPR_BENCH_DEVICE test hook) shard the query pool and gather the ranked results. (The training
    leg's bucketed gradient exchange: tests/test_gpu_train.py::test_gradient_buckets_are_handed_over_during_the_backward for
    the device-side hand-off, tests/test_dist_gloo.py for the collective; a gloo all-reduce of 0.94 GB of device memory
    takes 30 s per step, too slow for this suite. RCCL itself cannot be exercised on a one-GPU box.)"""
    import json
    import os
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, RPR_BENCH_BACKEND="gloo", RPR_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--batch", "64", "--docs", "200000", "--no-roofline", "--no-cpu-baseline", "--no-exact-fp32", "--secondary", ""]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["gather_bytes_per_rank"] > 0
    assert d["value"] > 0 and d["config"]["queries_per_step_per_gpu"] == 64
    print("[bench x2 gloo]", d["value"], d["forced_tail"])


def test_eight_rank_rehearsal_of_the_scaling_bench_and_the_cli(tmp_path):
    """the driver's SCALE command (`bench.py --gpus 8`) and an 8-rank `evaluate` run with the RCCL gather must
    work the first time an 8-GPU node exists. Rehearsed here with 8 gloo ranks sharing the one GPU of the test box
    (RPR_BENCH_DEVICE / RPR_EVAL_DEVICE hooks; reference evaluate.py:181-182, :468, :503-520): 8 distinct query shards, the
    end-of-run gather carries 8 x the per-rank bytes, the train leg's bucketed gradient exchange runs over the 8 ranks, and the
    CLI's gathered run.json equals the single-process one."""
    import socket
    repo = REPO

    def free_port():
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    # 1. bench.py --gpus 8, search + one f16x2 train step of 8 examples per rank (a gloo all-reduce of 0.94 GB per step)
    env = dict(os.environ, RPR_BENCH_BACKEND="gloo", RPR_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(repo, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1",
           "--batch", "64", "--docs", "200000", "--no-roofline", "--no-cpu-baseline", "--no-exact-fp32",
           "--secondary", "train,train_f16x2_only", "--train-steps", "1", "--train-bz", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env, cwd=repo)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["rccl_world_size"] == 8 and d["scaling"] == "weak"
    assert d["config"]["queries_per_step_per_gpu"] == 64 and d["value"] > 0
    per_rank = 64 * 10 * 32 * 4 + 64 * 10 * 4                       # tokens int32 [Q, B, L] + scores f32 [Q, B], one timed step
    assert d["gather_bytes_per_rank"] == per_rank and d["gathered_bytes_total"] == 8 * per_rank
    assert d["distinct_shards"] == 8, d.get("distinct_shards")
    tr = d["secondary"]["train_step"]
    assert "error" not in tr, tr
    assert tr["n_gpus"] == 8 and tr["allreduce_bytes_per_step_per_rank"] > 8e8 and tr["value"] > 0
    print("[bench x8 gloo]", round(d["value"], 1), "q/s;", "train", round(tr["ms_per_step"], 1), "ms/step over gloo")

    # 2. the evaluate CLI: 8 ranks + --gather_results=1 against one process
    ckpt, d2s_path, qdir, codes, queries, dims = _make_world(str(tmp_path))
    B, L = 5, 8
    _run(["-m", "t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids", "--docid_to_smtid_path", d2s_path])
    common = [f"--pretrained_path={ckpt}", "--task=t5seq_aq_retrieve_docids", f"--docid_to_smtid_path={d2s_path}",
              "--q_collection_paths=" + json.dumps([qdir]), "--batch_size=2", f"--max_new_token_for_docid={L}", f"--topk={B}"]
    out1, out8 = os.path.join(str(tmp_path), "out1"), os.path.join(str(tmp_path), "out8")
    _run(["-m", "t5_pretrainer.evaluate", f"--out_dir={out1}", "--gather_results=1"] + common)
    _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
          "-m", "t5_pretrainer.evaluate", f"--out_dir={out8}", "--gather_results=1"] + common,
         env={"RPR_DIST_BACKEND": "gloo", "RPR_EVAL_DEVICE": "0"})
    files8 = sorted(os.listdir(os.path.join(out8, "MSMARCO")))
    assert files8 == ["run.json"], files8                            # rank 0 wrote the merged file, no per-rank parts
    run1 = json.load(open(os.path.join(out1, "MSMARCO", [f for f in os.listdir(os.path.join(out1, "MSMARCO")) if f.startswith("run")][0])))
    run8 = json.load(open(os.path.join(out8, "MSMARCO", "run.json")))
    assert set(run8) == set(run1) == set(queries)                    # 11 queries over 8 ranks: wrap-around duplicates merged away
    for q in run1:
        assert set(run8[q]) == set(run1[q]), q
        for doc, sc in run1[q].items():
            assert abs(run8[q][doc] - sc) < 1e-4, (q, doc)


@pytest.mark.gpu
def test_randomised_parity_sweep():
    """tools/fuzz_parity.py, 16 seeded random cases (trie size, length, beams, V, skew, duplicates, log-softmax, explicit forks,
    radix selection; the small ones also against the CPU oracle): forced tail == step loop, radix selection == single block."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "fuzz_parity.py"), "16", "77"], capture_output=True, text=True,
                       timeout=900, cwd=repo)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "16 cases passed" in r.stdout


@pytest.mark.gpu
def test_randomised_training_shapes():
    """tools/fuzz_train.py, 5 seeded random batch shapes (bz, smtid length, ragged query lengths, encoder depth, d_ff) in all
    three GEMM arithmetics: gradients of rpr_lngknp_backward against torch autograd through the CPU oracle."""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "fuzz_train.py"), "5", "11"], capture_output=True, text=True,
                       timeout=1200, cwd=repo)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert "5 cases passed" in r.stdout
