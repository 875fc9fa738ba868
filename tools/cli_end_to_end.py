#!/usr/bin/env python3
"""End-to-end wall clock of the drop-in CLI (run on the GPU box): what a user of full_evaluate_t5seq_aq_encoder.sh:176-205 waits
for, next to the kernel-only rate of bench.py — VERDICT r5 item 4c.

Builds a synthetic MS MARCO-shaped world in a scratch directory (t5-base-dims checkpoint directory with a SentencePiece
tokenizer trained on the spot, docid_to_smtid.json for N docs x 32 codes written as real JSON, dev_queries/raw.tsv with Q
queries), then times, as subprocesses exactly like the shell script runs them:

  1. python -m t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids   (JSON -> binary trie cache; once per corpus)
  2. python -m t5_pretrainer.evaluate --task=t5seq_aq_retrieve_docids --topk=B --batch_size=1 --max_new_token_for_docid=32
     per beam count B (tokenisation, trie-cache load, searches, docid fan-out, run_0.json)
  3. python -m t5_pretrainer.evaluate --task=t5seq_aq_retrieve_docids_2  (merge + metrics on synthetic qrels)

Prints one JSON object; the CLI's own "timing ..." lines (ripor_amd/evaluate.py) give the host-side shares.
Usage: python tools/cli_end_to_end.py [--docs 8841823] [--queries 6980] [--beams 10,1000] [--workdir /tmp/rpr_cli] [--out x.json]"""
import argparse
import json
import os
import random
import shutil
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def write_docid_to_smtid(path, codes):
    """{"0000000": [-1,  12, 255, ...], ...} as real JSON, fixed-width rows assembled with numpy (json.dump of 8.8 M lists
    takes minutes in Python; whitespace inside the arrays is legal JSON)."""
    N, L = codes.shape
    lut = np.zeros((65536 if codes.max() > 255 else 256, 6), dtype=np.uint8)
    w = 6 if codes.max() > 255 else 4
    for v in range(lut.shape[0]):
        lut[v, :w] = np.frombuffer(f"{v:>{w - 1}d},".encode(), dtype=np.uint8)
    row_len = 15 + L * w + 2    # "0000000": [-1, | L fixed-width numbers, the last comma turned into ']' | ',' | newline
    chunk = 1 << 20
    with open(path, "wb") as f:
        f.write(b"{\n")
        for s in range(0, N, chunk):
            e = min(N, s + chunk)
            buf = np.full((e - s, row_len), ord(" "), dtype=np.uint8)
            ids = np.arange(s, e)
            buf[:, 0] = ord('"')
            for d in range(7):
                buf[:, 1 + d] = ord("0") + (ids // 10 ** (6 - d)) % 10
            buf[:, 8:8 + 7] = np.frombuffer(b'": [-1,', dtype=np.uint8)
            body = lut[codes[s:e].astype(np.int64), :w].reshape(e - s, L * w)
            buf[:, 15:15 + L * w] = body
            buf[:, 15 + L * w - 1] = ord("]")
            buf[:, 15 + L * w] = ord(",")
            buf[:, 15 + L * w + 1] = ord("\n")
            if e == N:
                buf[-1, 15 + L * w] = ord(" ")
            f.write(buf.tobytes())
        f.write(b"}\n")


def run(cmd, env, timeout=3600):
    t0 = time.perf_counter()
    p = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    dt = time.perf_counter() - t0
    if p.returncode != 0:
        raise SystemExit("command failed: " + " ".join(cmd) + "\n" + p.stdout[-2000:] + "\n" + p.stderr[-4000:])
    timing = {}
    for line in p.stdout.splitlines():
        if line.startswith("timing "):
            k, v = line[len("timing "):].split(": ", 1)
            timing[k] = json.loads(v)
        if line.startswith("queries per search call:"):
            timing["queries_per_search_call"] = int(line.split(":")[1].split("(")[0])
    return dt, timing, p.stdout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=8841823)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--beams", default="10,1000")
    ap.add_argument("--workdir", default="/tmp/rpr_cli")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import sentencepiece as spm
    from ripor_amd.modeling.t5_generative_retriever import T5SeqAQEncoder
    from ripor_amd.utils import synth

    root = a.workdir
    shutil.rmtree(root, ignore_errors=True)
    ckpt = os.path.join(root, "checkpoint")
    data = os.path.join(root, "msmarco_synth")           # "msmarco" in the path -> dataset name MSMARCO
    os.makedirs(ckpt)
    os.makedirs(os.path.join(data, "aq_smtid"))
    os.makedirs(os.path.join(data, "dev_queries"))
    L, V = 32, 256
    out = {"docs": a.docs, "queries": a.queries, "len": L, "model": "t5-base dims, synthetic weights", "setup_s": {}}

    t0 = time.perf_counter()
    random.seed(0)
    words = [f"{random.choice('bcdfghklmnprstvw')}{random.choice('aeiou')}{random.choice('bcdfghklmnprstvw')}{random.choice('aeiou')}{i % 97}"
             for i in range(4000)]
    corpus = os.path.join(root, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(20000):
            f.write(" ".join(random.choice(words) for _ in range(random.randint(3, 12))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(ckpt, "spiece"), vocab_size=8000, model_type="unigram",
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1, pad_piece="<pad>", eos_piece="</s>", unk_piece="<unk>",
                                   hard_vocab_limit=False, minloglevel=2)
    json.dump({"tokenizer_class": "T5Tokenizer", "extra_ids": 0, "model_max_length": 512}, open(os.path.join(ckpt, "tokenizer_config.json"), "w"))
    out["setup_s"]["tokenizer"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    dims = synth.t5_base_dims(L=L)
    T5SeqAQEncoder.from_synthetic(dims, seed=77).save_pretrained(ckpt)
    out["setup_s"]["checkpoint"] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    codes = synth.make_codes_fast(a.docs, L, V)
    d2s_path = os.path.join(data, "aq_smtid", "docid_to_smtid.json")
    write_docid_to_smtid(d2s_path, codes)
    out["setup_s"]["docid_to_smtid_json"] = round(time.perf_counter() - t0, 2)
    out["docid_to_smtid_json_bytes"] = os.path.getsize(d2s_path)
    del codes
    qdir = os.path.join(data, "dev_queries")
    with open(os.path.join(qdir, "raw.tsv"), "w") as f:
        for i in range(a.queries):
            f.write(f"{1000000 + i}\t" + " ".join(random.choice(words) for _ in range(random.randint(3, 9))) + "\n")

    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    dt, _, _ = run([sys.executable, "-m", "t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids", "--docid_to_smtid_path", d2s_path], env)
    out["build_list_smtid_to_nextids_s"] = round(dt, 2)
    out["trie_cache_bytes"] = os.path.getsize(os.path.join(data, "aq_smtid", "list_smtid_to_nextids.rprtrie"))
    out["retrieve"] = {}
    for B in [int(x) for x in a.beams.split(",") if x]:
        out_dir = os.path.join(root, f"out_b{B}")
        dt, timing, txt = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                               "--master-port", "29541", "-m", "t5_pretrainer.evaluate", f"--pretrained_path={ckpt}", f"--out_dir={out_dir}",
                               "--task=t5seq_aq_retrieve_docids", f"--docid_to_smtid_path={d2s_path}",
                               "--q_collection_paths=" + json.dumps([qdir]), "--batch_size=1", f"--max_new_token_for_docid={L}", f"--topk={B}"], env)
        assert "trie cache:" in txt
        run_path = os.path.join(out_dir, "MSMARCO", "run_0.json")
        e = {"wall_s": round(dt, 2), "queries_per_s_end_to_end": round(a.queries / dt, 1), "run_json_bytes": os.path.getsize(run_path)}
        e.update(timing)
        loop = timing.get("constrained_decode_doc", {}).get("total_s")
        if loop:
            e["queries_per_s_decode_loop"] = round(a.queries / loop, 1)   # tokenisation + searches + fan-out + run_0.json, without process start-up and loads
        # merge + metrics, as the script's second command (synthetic qrels: the top document of every query is relevant)
        t0 = time.perf_counter()
        runj = json.load(open(run_path))
        qrel_path = os.path.join(data, f"dev_qrel_b{B}.json")     # under the msmarco_* directory: evaluate() names the dataset from the path
        json.dump({qid: {max(docs, key=docs.get): 1} for qid, docs in runj.items() if docs}, open(qrel_path, "w"))
        e["docs_per_query_mean"] = round(float(np.mean([len(d) for d in runj.values()])), 1)
        del runj
        dt2, _, _ = run([sys.executable, "-m", "t5_pretrainer.evaluate", "--task=t5seq_aq_retrieve_docids_2", f"--out_dir={out_dir}",
                         "--q_collection_paths=" + json.dumps([qdir]), "--eval_qrel_path=" + json.dumps([qrel_path])], env)
        e["merge_and_metrics_s"] = round(dt2, 2)
        e["mrr_10"] = json.load(open(os.path.join(out_dir, "MSMARCO", "perf.json"))).get("mrr_10")
        out["retrieve"][f"beams{B}"] = e
        shutil.rmtree(out_dir, ignore_errors=True)
    print(json.dumps(out, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(out, open(a.out, "w"), indent=1)
    shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
