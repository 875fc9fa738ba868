"""Caller side of the constrained-beam-search path: build the trie inputs, run the search over a
query collection, map smtids to docids, write ``run_{rank}.json`` / ``run.json``, evaluate.

Mirrors the generative-retrieval tasks of reference t5_pretrainer/evaluate.py:
  ``constrained_decode`` :45-85, ``constrained_decode_doc`` :87-132, ``constrained_decode_smtid`` :134-178,
  ``t5seq_aq_retrieve_docids`` :396-487, ``t5seq_aq_retrieve_docids_2`` :489-526,
  ``t5seq_aq_get_qid_to_smtid_rankdata(_2)`` :528-655, ``evaluate`` :268-291, ``__main__`` dispatch :657-690,
with the same CLI flags (``EvalArguments`` subset, reference arguments.py:145-212), the same output
layout ``out_dir/<get_dataset_name(q_dir)>/run_{local_rank}.json`` and the same score convention
(``float(score_f32) * max_new_token`` per docid).

Differences by design: the per-level dict-of-strings / pickle cache is replaced by the device trie
(binary cache ``list_smtid_to_nextids.rprtrie`` next to ``docid_to_smtid.json``), and smtid -> docids
uses the sorted-row range returned by the search instead of a dict of 8.8 M strings (the dict path
is still accepted for drop-in callers).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ._lib import RiporHipError
from .tasks.generation import PrefixConstrainLogitProcessorFastSparse, generate_for_constrained_prefix_beam_search
from .utils.metrics import load_and_evaluate
from .utils.utils import convert_ptsmtids_to_strsmtid, get_dataset_name

QUERY_PREFIX = "query: "  # reference dataset/dataset.py:15


class DocidTable:
    """docids in code-row order; with the trie permutation this replaces ``smtid_to_docids``."""

    def __init__(self, docids: Sequence[str]):
        self.docids = list(docids)


def rankdata_from_ranges(qids, row_lo, row_hi, scores, perm, docids, max_new_token, apply_log_softmax_for_scores=False,
                         into: Optional[Dict[int, Dict[str, float]]] = None) -> Dict[int, Dict[str, float]]:
    """``{qid: {docid: score}}`` from sorted-row ranges (docids of a returned smtid = perm[lo:hi]); scores follow the
    reference (evaluate.py:118-127: sequences_scores * max_new_token unless log-softmax mode). Inputs are nested
    lists / arrays ``[Q]``, ``[Q, B]``."""
    out = {} if into is None else into
    for qid, los, his, rel_scores in zip(qids, row_lo, row_hi, scores):
        cur = out[int(qid)] = {}
        for l, h, rel_score in zip(los, his, rel_scores):
            if h <= l:
                print("smtid not in smtid_to_docid")
                continue
            for row in perm[l:h]:
                cur[docids[int(row)]] = rel_score if apply_log_softmax_for_scores else rel_score * max_new_token
    return out


class _Done:
    def synchronize(self):
        pass


class _HostCopier:
    """Device results -> pinned host memory on a side stream: ``to_host`` returns (event, host tensors); the event
    completes when the copies have, without waiting for anything enqueued on the main stream afterwards."""

    def __init__(self, device):
        self.device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.stream = torch.cuda.Stream(self.device)

    def to_host(self, *tensors):
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        host = []
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            for t in tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                h.copy_(t, non_blocking=True)
                t.record_stream(self.stream)
                host.append(h)
            done = torch.cuda.Event()
            done.record(self.stream)
        return done, tuple(host)


def constrained_decode_doc(model, dataloader, prefix_constrain_processor, smtid_to_docids, max_new_token, device,
                           out_dir, local_rank, topk=100, apply_log_softmax_for_scores=False, write=True, gather=False):
    """reference evaluate.py:87-132. ``smtid_to_docids``: the reference's dict
    ``{"c1_.._cL": [docids]}`` or a :class:`DocidTable` (range lookup, no strings).

    ``gather=True`` (multi-process runs with a DocidTable): instead of one ``run_{rank}.json`` per rank, the ranks'
    row ranges and scores are exchanged with one RCCL all_gather (ripor_amd/dist_gather.py) and rank 0 writes the
    merged ``run.json`` itself — the ``..._2`` merge step then finds it complete (:func:`merge_runs`)."""
    import torch.distributed as dist
    from .dist_gather import all_gather_results
    qid_to_rankdata: Dict[int, Dict[str, float]] = {}
    use_ranges = isinstance(smtid_to_docids, DocidTable)
    gather = bool(gather) and use_ranges and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    kept = []   # per batch (qids, row_lo, row_hi, scores) on the device, for the gather
    _dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    copier = _HostCopier(_dev) if use_ranges and not gather and _dev.type == "cuda" else None
    pending = None   # (qids, event, host tensors) of the previous batch: decoded while the GPU runs the current one

    def finish(p):
        qids, done, (sc, lo, hi), guard = p
        done.synchronize()
        if guard is not None:
            r = guard.result()               # waits for this batch's status words only; repeats the batch if a guard fired
            if guard.repeated:
                sc, lo, hi = r.scores[:, :topk].cpu(), r.row_lo[:, :topk].cpu(), r.row_hi[:, :topk].cpu()
        perm = prefix_constrain_processor.trie(device).perm
        rankdata_from_ranges(qids, lo.tolist(), hi.tolist(), sc.tolist(), perm, smtid_to_docids.docids, max_new_token,
                             apply_log_softmax_for_scores, into=qid_to_rankdata)

    # where the wall clock of the call goes (printed once at the end: host-side share of an end-to-end run, tools/cli_end_to_end.py)
    import time
    tm = {"batches": 0, "queries": 0, "tokenize_and_batch_s": 0.0, "enqueue_search_s": 0.0, "wait_and_fanout_s": 0.0, "write_json_s": 0.0}
    t_call = time.perf_counter()
    loader_it = iter(dataloader)
    while True:
        t0 = time.perf_counter()
        batch = next(loader_it, None)
        tm["tokenize_and_batch_s"] += time.perf_counter() - t0
        if batch is None:
            break
        t0 = time.perf_counter()
        with torch.no_grad():
            # pinned + non_blocking: a pageable host-to-device copy is ordered behind the previous batch's search on the
            # stream and blocks the host until then (tools/cli_end_to_end.py at beam 1000: the host sat 136 of 145 s in this
            # call and turned the previous batch into run.json entries only afterwards)
            inputs = {k: (v.pin_memory().to(device, non_blocking=True) if _dev.type == "cuda" else v.to(device))
                      for k, v in batch.items() if k != "id"}
            outputs = generate_for_constrained_prefix_beam_search(
                model, prefix_constrain_processor, input_ids=inputs["input_ids"].long(),
                attention_mask=inputs["attention_mask"].long(), max_new_tokens=max_new_token, output_scores=True,
                return_dict=True, return_dict_in_generate=True, num_beams=topk, num_return_sequences=topk,
                apply_log_softmax_for_scores=apply_log_softmax_for_scores,
                defer_status=copier is not None)   # the side-stream path checks the guards in finish(), one batch later
        batch_qids = batch["id"].cpu().tolist()
        tm["enqueue_search_s"] += time.perf_counter() - t0
        tm["batches"] += 1
        tm["queries"] += len(batch_qids)
        t0 = time.perf_counter()
        if gather:
            kept.append((batch["id"].to(outputs.row_lo.device), outputs.row_lo.view(-1, topk), outputs.row_hi.view(-1, topk),
                         outputs.sequences_scores.view(-1, topk)))
            continue
        if use_ranges:
            # the search is asynchronous: its results travel to pinned host memory on a side stream, and the previous
            # batch is turned into {docid: score} dicts while this one runs (the reference synchronises Q*B times a step)
            res = (outputs.sequences_scores.view(-1, topk), outputs.row_lo.view(-1, topk), outputs.row_hi.view(-1, topk))
            nxt = (batch_qids,) + (copier.to_host(*res) if copier else (_Done(), tuple(t.cpu() for t in res))) + (getattr(outputs, "guard", None),)
            if pending is not None:
                finish(pending)
            pending = nxt
            tm["wait_and_fanout_s"] += time.perf_counter() - t0
        else:
            relevant_scores = outputs.sequences_scores.view(-1, topk).cpu().tolist()
            str_smtids = convert_ptsmtids_to_strsmtid(outputs.sequences.view(-1, topk, max_new_token + 1), max_new_token)
            for qid, ranked_smtids, rel_scores in zip(batch_qids, str_smtids, relevant_scores):
                cur = qid_to_rankdata[qid] = {}
                for smtid, rel_score in zip(ranked_smtids, rel_scores):
                    if smtid not in smtid_to_docids:
                        print(f"smtid: {smtid} not in smtid_to_docid")
                    else:
                        for docid in smtid_to_docids[smtid]:
                            cur[docid] = rel_score if apply_log_softmax_for_scores else rel_score * max_new_token
    if pending is not None:
        t0 = time.perf_counter()
        finish(pending)
        tm["wait_and_fanout_s"] += time.perf_counter() - t0
    if gather:
        gdev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if kept:
            qids = torch.cat([k[0] for k in kept]); lo = torch.cat([k[1] for k in kept])
            hi = torch.cat([k[2] for k in kept]); sc = torch.cat([k[3] for k in kept])
        else:   # a rank whose shard is empty (fewer queries than ranks cannot happen with wrap-around padding,
                # but an empty collection can): zero-size tensors keep the collective well-formed
            qids = torch.zeros((0,), dtype=torch.long, device=gdev); lo = torch.zeros((0, topk), dtype=torch.long, device=gdev)
            hi = torch.zeros((0, topk), dtype=torch.long, device=gdev); sc = torch.zeros((0, topk), dtype=torch.float32, device=gdev)
        qids, sc, lo, hi = all_gather_results(qids, sc, lo, hi)       # equal shard sizes (wrap-around padding)
        if dist.get_rank() == 0:
            # leftovers of an earlier per-rank run would make merge_runs prefer them over the gathered file
            for stale in [p for p in os.listdir(out_dir) if p.startswith("run_") and p.endswith(".json")]:
                os.remove(os.path.join(out_dir, stale))
            perm = prefix_constrain_processor.trie(device).perm
            rankdata_from_ranges(qids.cpu().tolist(), lo.cpu().tolist(), hi.cpu().tolist(), sc.cpu().tolist(), perm,
                                 smtid_to_docids.docids, max_new_token, apply_log_softmax_for_scores, into=qid_to_rankdata)
            if write:
                with open(os.path.join(out_dir, "run.json"), "w") as fout:
                    json.dump(qid_to_rankdata, fout)
        dist.barrier()
        return qid_to_rankdata
    if write:
        t0 = time.perf_counter()
        with open(os.path.join(out_dir, f"run_{local_rank}.json"), "w") as fout:
            json.dump(qid_to_rankdata, fout)
        tm["write_json_s"] = time.perf_counter() - t0
    tm["total_s"] = time.perf_counter() - t_call
    print("timing constrained_decode_doc: " + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()}))
    return qid_to_rankdata


def constrained_decode(model, dataloader, prefix_constrain_processor, smtid_to_docid, max_new_token, device, out_dir,
                       local_rank, topk=100, write=True):
    """reference evaluate.py:45-85: ``{qid: {smtid_string: score}}`` -> ``qid_to_smtid_{rank}.json``
    (scores are the raw ``sequences_scores``, not multiplied by the length)."""
    qid_to_rankdata: Dict[int, Dict[str, float]] = {}
    for batch in dataloader:
        with torch.no_grad():
            inputs = {k: v.to(device) for k, v in batch.items() if k != "id"}
            outputs = generate_for_constrained_prefix_beam_search(
                model, prefix_constrain_processor, input_ids=inputs["input_ids"].long(),
                attention_mask=inputs["attention_mask"].long(), max_new_tokens=max_new_token, output_scores=True,
                return_dict=True, return_dict_in_generate=True, num_beams=topk, num_return_sequences=topk)
        batch_qids = batch["id"].cpu().tolist()
        str_smtids = convert_ptsmtids_to_strsmtid(outputs.sequences.view(-1, topk, max_new_token + 1), max_new_token)
        relevant_scores = outputs.sequences_scores.view(-1, topk).cpu().tolist()
        lo = outputs.row_lo.view(-1, topk).cpu().tolist()
        hi = outputs.row_hi.view(-1, topk).cpu().tolist()
        for qid, ranked, rel, los, his in zip(batch_qids, str_smtids, relevant_scores, lo, hi):
            cur = qid_to_rankdata[qid] = {}
            for smtid, rel_score, l, h in zip(ranked, rel, los, his):
                known = (smtid in smtid_to_docid) if smtid_to_docid is not None else (h > l)
                if not known:
                    print(f"smtid: {smtid} not in smtid_to_docid")
                else:
                    cur[smtid] = rel_score
    if write:
        with open(os.path.join(out_dir, f"qid_to_smtid_{local_rank}.json"), "w") as fout:
            json.dump(qid_to_rankdata, fout)
    return qid_to_rankdata


def constrained_decode_smtid(model, dataloader, prefix_constrain_processor, smtid_to_docids, max_new_token, device,
                             out_dir, local_rank, topk=100, apply_log_softmax_for_scores=False, write=True):
    """reference evaluate.py:134-178: nested ``{qid: {smtid: {docid: score}}}`` for the training-data
    generation pass (prefix search: max_new_token in {4, 8, 16, 32}, many docids per smtid) ->
    ``qid_smtid_rankdata_{rank}.json``. ``smtid_to_docids``: the reference's dict or a DocidTable."""
    out: Dict[int, Dict[str, Dict[str, float]]] = {}
    use_ranges = isinstance(smtid_to_docids, DocidTable)
    for batch in dataloader:
        with torch.no_grad():
            inputs = {k: v.to(device) for k, v in batch.items() if k != "id"}
            outputs = generate_for_constrained_prefix_beam_search(
                model, prefix_constrain_processor, input_ids=inputs["input_ids"].long(),
                attention_mask=inputs["attention_mask"].long(), max_new_tokens=max_new_token, output_scores=True,
                return_dict=True, return_dict_in_generate=True, num_beams=topk, num_return_sequences=topk,
                apply_log_softmax_for_scores=apply_log_softmax_for_scores)
        batch_qids = batch["id"].cpu().tolist()
        str_smtids = convert_ptsmtids_to_strsmtid(outputs.sequences.view(-1, topk, max_new_token + 1), max_new_token)
        relevant_scores = outputs.sequences_scores.view(-1, topk).cpu().tolist()
        lo = outputs.row_lo.view(-1, topk).cpu().tolist()
        hi = outputs.row_hi.view(-1, topk).cpu().tolist()
        perm = prefix_constrain_processor.trie(device).perm if use_ranges else None
        for qid, ranked, rel, los, his in zip(batch_qids, str_smtids, relevant_scores, lo, hi):
            cur = out[qid] = {}
            for smtid, rel_score, l, h in zip(ranked, rel, los, his):
                docs = cur[smtid] = {}
                score = rel_score if apply_log_softmax_for_scores else rel_score * max_new_token
                if use_ranges:
                    for row in perm[l:h]:
                        docs[smtid_to_docids.docids[int(row)]] = score
                elif smtid in smtid_to_docids:
                    for docid in smtid_to_docids[smtid]:
                        docs[docid] = score
    if write:
        with open(os.path.join(out_dir, f"qid_smtid_rankdata_{local_rank}.json"), "w") as fout:
            json.dump(out, fout)
    return out


def build_smtid_to_docids(docid_to_smtids: Dict[str, Sequence[int]], max_new_token: int) -> Dict[str, List[str]]:
    """reference evaluate.py:439-446 (kept for drop-in callers that want the dict)."""
    out: Dict[str, List[str]] = {}
    for docid, smtids in docid_to_smtids.items():
        assert smtids[0] == -1, smtids
        out.setdefault("_".join(str(x) for x in smtids[1:1 + max_new_token]), []).append(docid)
    return out


def trie_cache_path(docid_to_smtid_path: str) -> str:
    return os.path.join(os.path.dirname(docid_to_smtid_path), "list_smtid_to_nextids.rprtrie")


def fresh_trie_cache(docid_to_smtid_path: str) -> Optional[str]:
    """Path of the binary trie cache if it exists, stores the docids and was built from the JSON as it is now (same
    size and mtime as recorded at build time), else None. Host only."""
    from .engine import trie_file_info
    cache = trie_cache_path(docid_to_smtid_path)
    if not (os.path.exists(cache) and os.path.exists(docid_to_smtid_path)):
        return None
    try:
        info = trie_file_info(cache)
    except RiporHipError:
        return None
    st = os.stat(docid_to_smtid_path)
    if info["key_bytes"] and info["src_size"] == st.st_size and info["src_mtime_ns"] == st.st_mtime_ns:
        return cache
    return None


def load_docid_table(docid_to_smtid_path: str, vocab_size: int, max_new_token: int, device=None):
    """Returns (processor, DocidTable) for ``docid_to_smtid.json`` ({"docid": [-1, c1..cL]}).

    Fast path (replaces the reference's pickle cache, evaluate.py:404-408,428-432): when
    ``list_smtid_to_nextids.rprtrie`` next to the JSON is fresh (written by ``python -m
    t5_pretrainer.aq_preprocess.build_list_smtid_to_nextids``), the sorted code matrix, the permutation and the docid
    strings come from it — no JSON parse, no sort. A search over ``max_new_token`` < L positions walks the first
    ``max_new_token`` columns of the same trie (sub-smtid retrieval, evaluate.py:442).
    Otherwise the JSON is parsed (streaming C++ reader) and the code matrix truncated to ``max_new_token`` columns."""
    from .engine import read_docid_to_smtid
    cache = fresh_trie_cache(docid_to_smtid_path)
    if cache is not None:
        proc = PrefixConstrainLogitProcessorFastSparse.from_trie_cache(cache, vocab_size)
        assert proc.max_len >= max_new_token, (proc.max_len, max_new_token)
        print(f"trie cache: {cache}")
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        return proc, DocidTable(proc.trie(device).docids())
    try:  # streaming C++ reader: seconds and ~0.6 GB for the 8.8 M-doc MS MARCO file
        docids, full = read_docid_to_smtid(docid_to_smtid_path)
    except RiporHipError as e:  # e.g. escaped characters in a docid: the general (slow) JSON path
        print(f"rpr_d2s reader declined ({e}); falling back to json.load")
        with open(docid_to_smtid_path) as fin:
            docid_to_smtids = json.load(fin)
        docids = list(docid_to_smtids.keys())
        assert all(docid_to_smtids[d][0] == -1 for d in docids[:16])
        full = np.asarray([docid_to_smtids[d][1:] for d in docids], dtype=np.int64)
    codes = np.ascontiguousarray(full[:, :max_new_token]).astype(np.int64)
    assert codes.shape[1] == max_new_token, (codes.shape, max_new_token)  # evaluate.py:449
    return PrefixConstrainLogitProcessorFastSparse.from_codes(codes, vocab_size), DocidTable(docids)


# ----------------------------------------------------------------------------- query side
class QueryCollection:
    """``raw.tsv`` (``id\\ttext``) reader with the "query: " prefix (reference
    dataset/dataset.py:266-332, id_style="row_id", add_prefix=True, is_query=True)."""

    def __init__(self, data_dir: str):
        self.ids, self.texts = [], []
        with open(os.path.join(data_dir, "raw.tsv")) as reader:
            for line in reader:
                if len(line) > 1:
                    id_, *data = line.split("\t")
                    self.ids.append(id_.strip())
                    self.texts.append(QUERY_PREFIX + " ".join(" ".join(data).splitlines()))

    def __len__(self):
        return len(self.ids)


def query_batches(collection: QueryCollection, tokenizer, indices: Sequence[int], batch_size: int, max_length: int = 256):
    """pad-to-longest / truncate batches like CollectionDataWithDocIDLoader.collate_fn
    (reference dataset/dataloader.py:62-79)."""
    for s in range(0, len(indices), batch_size):
        sel = indices[s:s + batch_size]
        enc = tokenizer([collection.texts[i] for i in sel], add_special_tokens=True, padding="longest",
                        truncation="longest_first", max_length=max_length, return_attention_mask=True)
        # "decoder_input_ids": the reference loader adds the rows' smtid lists, [-1] each for a query collection
        # (dataset.py:296-305, dataloader.py:76); nothing on this path reads them, the key is kept for drop-in callers
        yield {"input_ids": torch.tensor(enc["input_ids"]), "attention_mask": torch.tensor(enc["attention_mask"]),
               "decoder_input_ids": torch.full((len(sel), 1), -1, dtype=torch.long),
               "id": torch.tensor([int(collection.ids[i]) for i in sel], dtype=torch.long)}


def search_batch_size(cfg, batch_size: int, topk: int, max_new_token: int, flag: int = -1, device=None) -> int:
    """Queries per ``rpr_search`` call for the CLI tasks. The reference scripts pass ``--batch_size=1`` (retrieval,
    topk 1000) or 4 (rank-data generation, topk 100) because their decoding loop is host-bound; this path is built
    for many queries in flight and its results do not depend on how queries are batched (tests/test_gpu_fullsize.py),
    so by default (``--search_batch_size=-1``) the tasks regroup the query stream into the largest batch whose
    workspace (dominated by the self-attention KV cache, 2*Ndec*L*B*inner*4 bytes per query) fits ~60 % of the free
    HBM, capped at 2176 and aligned to whole rounds of GEMM tiles (``align_to_gemm_rounds``). ``--search_batch_size=0`` keeps ``--batch_size``; a positive value is used as given."""
    if flag == 0:
        return batch_size
    if flag > 0:
        return flag
    nd, inner = cfg.num_decoder_layers, cfg.num_heads * cfg.d_kv
    per_query = 2 * nd * max_new_token * topk * inner * 4                       # K and V cache
    per_query += topk * (6 * cfg.d_model + 4 * inner + 3 * cfg.d_ff) * 4          # activations + f16 planes of a step
    per_query += topk * max(cfg.decoder_vocab_sizes) * 8 + 64 * 1024            # logits, trie bounds, encoder side
    try:
        free, _ = torch.cuda.mem_get_info(device)
    except Exception:
        free = 64 << 30
    auto = max(1, min(2176, int(0.6 * free // per_query)))
    return max(batch_size, align_to_gemm_rounds(auto, topk, cfg.d_model))


def align_to_gemm_rounds(q_max: int, beams: int, d_model: int, lane_min_rows: int = 10240) -> int:
    """Largest Q <= q_max (but not below 0.75 q_max) whose decoder rows fill whole rounds of 256x256 GEMM tiles for the
    d_model-wide projections (then also for the 3x / 4x wider ones): a launch just over a whole number of rounds leaves
    most of the chip idle in its last round. A batch of at least ``lane_min_rows`` decoder rows runs as two halves on two lanes
    of 128 CUs each (``rpr_set_lane_split``), smaller ones on the 256 CUs. Measured: t5-large, beam 100: 160 queries in
    flight (63 row tiles x 4 = 252 tiles) 116.8 q/s vs 108.7 q/s at 128 (200 tiles); t5-base, beam 10: 2176 -> 255 tiles on
    one stream, 2150 -> 2 x 126 tiles on two lanes. Small batches (less than one round of tiles) are left alone."""
    cols = max(1, (d_model + 255) // 256)

    def eff(q):
        lanes = 2 if lane_min_rows and q >= 2 and q * beams >= lane_min_rows else 1
        tiles = ((((q + lanes - 1) // lanes) * beams + 255) // 256) * cols
        cus = 256 // lanes
        return tiles / float(((tiles + cus - 1) // cus) * cus)

    if ((q_max * beams + 255) // 256) * cols < 200:
        return q_max
    best, best_eff = q_max, eff(q_max)
    for q in range(q_max, max(1, int(0.75 * q_max)) - 1, -1):
        e = eff(q)
        if e > best_eff + 0.02:
            best, best_eff = q, e
        if best_eff >= 0.98:
            break
    return best


def ddp_setup():
    """reference evaluate.py:181-182; RCCL is torch's "nccl" backend on ROCm."""
    import torch.distributed as dist
    if "RANK" in os.environ and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RPR_DIST_BACKEND: test hook (an 8-rank rehearsal on the one GPU of a test box runs over gloo; RCCL needs a GPU per rank)
        dist.init_process_group(backend=os.environ.get("RPR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo"))


def _device_index(local_rank: int) -> int:
    """The HIP device of this rank: its local rank, like the reference (``model.to(args.local_rank)``). RPR_EVAL_DEVICE: test hook
    that puts every rank on one device (tests/test_gpu_cli.py: 8 ranks on the single GPU of the test box)."""
    return int(os.environ.get("RPR_EVAL_DEVICE", local_rank))


def _list_flag(v):
    if isinstance(v, (list, tuple)) and len(v) == 1 and isinstance(v[0], str) and v[0].lstrip().startswith("["):
        return json.loads(v[0])  # list-valued flags arrive as one JSON string (evaluate.py:457-459)
    return list(v) if isinstance(v, (list, tuple)) else [v]


def t5seq_aq_retrieve_docids(args):
    """reference evaluate.py:396-487."""
    import torch.distributed as dist
    from transformers import AutoTokenizer
    from .dataset.sharding import shard_indices
    from .modeling.t5_generative_retriever import T5SeqAQEncoder

    ddp_setup()
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    local_rank = max(0, int(args.local_rank if args.local_rank >= 0 else os.environ.get("LOCAL_RANK", 0)))
    import time
    t0 = time.perf_counter()
    model = T5SeqAQEncoder.from_pretrained(args.pretrained_path)
    model.eval()
    if len(set(model.config.decoder_vocab_sizes)) != 1:
        raise ValueError("not valid decoder_vocab_size")
    max_new_token = args.max_new_token_for_docid
    device = _device_index(local_rank)
    t1 = time.perf_counter()
    processor, table = load_docid_table(args.docid_to_smtid_path, model.config.decoder_vocab_sizes[0], max_new_token,
                                        device=device)
    t2 = time.perf_counter()
    if rank == 0:
        print("max_new_token: ", max_new_token)
        os.makedirs(args.out_dir, exist_ok=True)
    tokenizer = AutoTokenizer.from_pretrained(args.pretrained_path)
    model.to(device)
    model.base_model.config.decoding = True
    if rank == 0:
        print("timing setup: " + json.dumps({"read_checkpoint_s": round(t1 - t0, 3), "docid_table_and_trie_s": round(t2 - t1, 3),
                                             "tokenizer_and_weights_to_device_s": round(time.perf_counter() - t2, 3)}))
    for data_dir in _list_flag(args.q_collection_paths):
        coll = QueryCollection(data_dir)
        out_dir = os.path.join(args.out_dir, get_dataset_name(data_dir))
        print("out_dir: ", out_dir)
        os.makedirs(out_dir, exist_ok=True)  # every rank: removes the reference's mkdir race
        qbs = search_batch_size(model.config, args.batch_size, args.topk, max_new_token, args.search_batch_size, device)
        if rank == 0:
            print(f"queries per search call: {qbs} (--batch_size={args.batch_size})")
        loader = query_batches(coll, tokenizer, shard_indices(len(coll), world, rank), qbs, 256)
        constrained_decode_doc(model.base_model, loader, processor, table, max_new_token, device=device,
                               out_dir=out_dir, local_rank=local_rank, topk=args.topk,
                               apply_log_softmax_for_scores=args.apply_log_softmax_for_scores,
                               gather=bool(args.gather_results))


def merge_runs(out_dir: str, expected_files: Optional[int] = None) -> Dict[str, Dict[str, float]]:
    """reference evaluate.py:496-524: merge ``run_*.json`` into ``run.json`` and delete the parts."""
    run_path = os.path.join(out_dir, "run.json")
    parts = [p for p in os.listdir(out_dir) if "run" in p and p != "run.json"]
    if os.path.exists(run_path) and not parts:
        # written by rank 0 after the RCCL gather (constrained_decode_doc(gather=True)): already complete
        with open(run_path) as fin:
            merged = json.load(fin)
        print("run.json is already merged (gathered over RCCL): {} queries".format(len(merged)))
        return merged
    if os.path.exists(run_path):
        print("old run.json exisit.")
        os.remove(run_path)
    sub_paths = [p for p in os.listdir(out_dir) if "run" in p]
    if expected_files is not None:
        assert len(sub_paths) == expected_files, (sub_paths, expected_files)
    merged: Dict[str, Dict[str, float]] = {}
    for sub_path in sub_paths:
        with open(os.path.join(out_dir, sub_path)) as fin:
            for qid, rankdata in json.load(fin).items():
                merged.setdefault(qid, {}).update(rankdata)
    print("length of pids and avg rankdata length in qid_to_rankdata: {}, {}".format(
        len(merged), np.mean([len(xs) for xs in merged.values()]) if merged else 0.0))
    with open(run_path, "w") as fout:
        json.dump(merged, fout)
    for sub_path in sub_paths:
        os.remove(os.path.join(out_dir, sub_path))
    return merged


def t5seq_aq_get_qid_to_smtid_rankdata(args):
    """reference evaluate.py:528-611 (train-query pass that produces the (query, smtid-prefix, docids)
    rank data): same search with ``--max_new_token`` in {4, 8, 16, 32} over ``--train_query_dir``."""
    import torch.distributed as dist
    from transformers import AutoTokenizer
    from .dataset.sharding import shard_indices
    from .modeling.t5_generative_retriever import T5SeqAQEncoder

    ddp_setup()
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    local_rank = max(0, int(args.local_rank if args.local_rank >= 0 else os.environ.get("LOCAL_RANK", 0)))
    model = T5SeqAQEncoder.from_pretrained(args.pretrained_path)
    model.eval()
    if len(set(model.config.decoder_vocab_sizes)) != 1:
        raise ValueError("not valid decoder_vocab_size")
    assert args.max_new_token in [4, 8, 16, 32], args.max_new_token
    processor, table = load_docid_table(args.docid_to_smtid_path, model.config.decoder_vocab_sizes[0], args.max_new_token,
                                        device=local_rank)
    os.makedirs(args.out_dir, exist_ok=True)
    tokenizer = AutoTokenizer.from_pretrained(args.pretrained_path)
    coll = QueryCollection(args.train_query_dir)
    model.to(local_rank)
    model.base_model.config.decoding = True
    qbs = search_batch_size(model.config, args.batch_size, args.topk, args.max_new_token, args.search_batch_size, local_rank)
    if rank == 0:
        print(f"queries per search call: {qbs} (--batch_size={args.batch_size})")
    loader = query_batches(coll, tokenizer, shard_indices(len(coll), world, rank), qbs, 256)
    constrained_decode_smtid(model.base_model, loader, processor, table, args.max_new_token, device=local_rank,
                             out_dir=args.out_dir, local_rank=local_rank, topk=args.topk,
                             apply_log_softmax_for_scores=args.apply_log_softmax_for_scores)


def merge_qid_smtid_rankdata(out_dir: str, expected_files: Optional[int] = None):
    """reference evaluate.py:613-655: merge ``qid_smtid_rankdata_*.json`` into ``qid_smtid_rankdata.json``."""
    final = os.path.join(out_dir, "qid_smtid_rankdata.json")
    if os.path.exists(final):
        print("old run.json exisit.")
        os.remove(final)
    sub_paths = [p for p in os.listdir(out_dir) if "qid_smtid_rankdata" in p]
    if expected_files is not None:
        assert len(sub_paths) == expected_files, (sub_paths, expected_files)
    merged: Dict[str, Dict[str, Dict[str, float]]] = {}
    for sub_path in sub_paths:
        with open(os.path.join(out_dir, sub_path)) as fin:
            for qid, by_smtid in json.load(fin).items():
                cur = merged.setdefault(qid, {})
                for smtid, docs in by_smtid.items():
                    cur.setdefault(smtid, {}).update(docs)
    smtid_lengths = [len(v) for v in merged.values()]
    doc_lengths = [len(d) for v in merged.values() for d in v.values()]
    qs = [0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0]
    print("smtid_length per query: ", np.quantile(smtid_lengths, qs) if smtid_lengths else [])
    print("doc_length per smtid: ", np.quantile(doc_lengths, qs) if doc_lengths else [])
    with open(final, "w") as fout:
        json.dump(merged, fout)
    for sub_path in sub_paths:
        os.remove(os.path.join(out_dir, sub_path))
    return merged


def t5seq_aq_get_qid_to_smtid_rankdata_2(args):
    return merge_qid_smtid_rankdata(args.out_dir, torch.cuda.device_count() if torch.cuda.is_available() else None)


def t5seq_aq_retrieve_docids_2(args):
    """reference evaluate.py:489-526."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else None
    for data_dir in _list_flag(args.q_collection_paths):
        merge_runs(os.path.join(args.out_dir, get_dataset_name(data_dir)), expected_files=n)
    return evaluate(args)


def evaluate(args):
    """reference evaluate.py:268-291."""
    eval_qrel_path = _list_flag(args.eval_qrel_path)
    eval_metric = args.eval_metric
    if isinstance(eval_metric, (list, tuple)) and len(eval_metric) == 1 and isinstance(eval_metric[0], str):
        eval_metric = json.loads(eval_metric[0])
    if len(eval_metric) < len(eval_qrel_path):   # zip() would silently drop the qrels without a metric list
        raise ValueError(f"--eval_metric has {len(eval_metric)} entries for {len(eval_qrel_path)} --eval_qrel_path files")
    res_all: Dict[str, dict] = {}
    for qrel_file_path, metrics in zip(eval_qrel_path, eval_metric):
        if qrel_file_path is None:
            continue
        res = {}
        name = get_dataset_name(qrel_file_path)
        for metric in metrics:
            res.update(load_and_evaluate(qrel_file_path=qrel_file_path,
                                         run_file_path=os.path.join(args.out_dir, name, "run.json"), metric=metric))
        res_all.setdefault(name, {}).update(res)
        with open(os.path.join(args.out_dir, name, "perf.json"), "a") as f:
            json.dump(res, f)
    with open(os.path.join(args.out_dir, "perf_all_datasets.json"), "a") as f:
        json.dump(res_all, f)
    return res_all


def get_args(argv=None):
    """EvalArguments fields used by the generative-retrieval branch (reference arguments.py:145-212)."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_path", default="")
    ap.add_argument("--out_dir", default="")
    ap.add_argument("--task", default="")
    ap.add_argument("--docid_to_smtid_path", default=None)
    ap.add_argument("--q_collection_paths", nargs="+", default=[])
    ap.add_argument("--eval_qrel_path", nargs="+", default=[])
    # reference arguments.py:170-175: one metric list per entry of the stock --eval_qrel_path (MSMARCO dev,
    # TREC-DL 2019 graded / binary, TREC-DL 2020 graded / binary)
    ap.add_argument("--eval_metric", nargs="+", default=[["mrr_10", "recall"], ["ndcg_cut"], ["mrr_10", "recall"],
                                                          ["ndcg_cut"], ["mrr_10", "recall"]])
    ap.add_argument("--batch_size", type=int, default=64)
    ap.add_argument("--gather_results", type=int, default=1,
                    help="multi-process runs: 1 = one RCCL all_gather, rank 0 writes run.json; 0 = run_{rank}.json files")
    ap.add_argument("--search_batch_size", type=int, default=-1,
                    help="queries per search call: -1 = as many as fit the HBM (>= --batch_size), 0 = --batch_size")
    ap.add_argument("--max_new_token_for_docid", type=int, default=32)
    ap.add_argument("--topk", type=int, default=200)
    ap.add_argument("--local_rank", "--local-rank", type=int, default=-1)
    ap.add_argument("--max_new_token", type=int, default=None)
    ap.add_argument("--train_query_dir", default=None)
    ap.add_argument("--apply_log_softmax_for_scores", type=lambda s: str(s).lower() in ("1", "true", "yes"),
                    default=False)
    return ap.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    if args.task == "t5seq_aq_retrieve_docids":
        t5seq_aq_retrieve_docids(args)
    elif args.task == "t5seq_aq_retrieve_docids_2":
        t5seq_aq_retrieve_docids_2(args)
    elif args.task == "t5seq_aq_get_qid_to_smtid_rankdata":
        t5seq_aq_get_qid_to_smtid_rankdata(args)
    elif args.task == "t5seq_aq_get_qid_to_smtid_rankdata_2":
        t5seq_aq_get_qid_to_smtid_rankdata_2(args)
    elif args.task == "evaluate":
        evaluate(args)
    else:
        raise ValueError(f"task: {args.task} is not valid.")


if __name__ == "__main__":
    main()
