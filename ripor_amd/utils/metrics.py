"""Ranking metrics for run.json files without pytrec_eval (not installed here).

Mirrors the call surface of reference t5_pretrainer/utils/metrics.py (``truncate_run`` :9-15,
``mrr_k`` :18-25, ``load_and_evaluate`` :63-79) for the metrics the generative-retrieval branch of
full_evaluate_t5seq_aq_encoder.sh asks for (mrr_10, recall, ndcg_cut).

Rules restated from trec_eval 9.0 (the C library pytrec_eval wraps; not available offline, so these are pinned by
hand-computed vectors in tests/test_host_logic.py rather than by running it):
  * ranking (trec_eval form_res_rels.c, comp_sim_docno): documents of a query are ordered by score DESCENDING and,
    at equal score, by docno DESCENDING (byte-wise string compare); the score field is a C float (trec_eval.h
    TEXT_RESULTS.sim), so scores are compared after rounding to float32;
  * only queries present in BOTH the run and the qrels are evaluated and averaged (pytrec_eval RelevanceEvaluator);
  * recip_rank (m_recip_rank.c): 1 / rank of the first document with relevance >= 1, 0 if none is retrieved;
  * recall_k (m_recall.c): relevant documents among the first k / all documents with relevance >= 1 in the qrels,
    k in 5, 10, 15, 20, 30, 100, 200, 500, 1000;
  * ndcg_cut_k (m_ndcg_cut.c): gain = the relevance level itself (linear), discount log2(rank + 1), ideal DCG from the
    qrels' positive levels sorted descending and cut at k; same cutoffs.
``truncate_run`` is the reference's own helper (stable sort by score only, so a tie AT the cut keeps run-file order).
"""
from __future__ import annotations

import json
import math
from typing import Dict

RECALL_CUTS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)
NDCG_CUTS = (5, 10, 15, 20, 30, 100, 200, 500, 1000)


def truncate_run(run: Dict[str, Dict[str, float]], k: int):
    out = {}
    for qid, docs in run.items():
        ranked = sorted(docs.items(), key=lambda item: item[1], reverse=True)  # stable, like the reference
        out[qid] = dict(ranked[:k])
    return out


def _trec_rank(docs: Dict[str, float]):
    import numpy as np
    return [d for d, _ in sorted(docs.items(), key=lambda it: (float(np.float32(it[1])), it[0]), reverse=True)]


def mrr_k(run, qrel, k: int, agg: bool = True):
    truncated = truncate_run(run, k)
    per_q = {}
    for qid, docs in truncated.items():
        if qid not in qrel:
            continue
        rel = qrel[qid]
        rr = 0.0
        for rank, d in enumerate(_trec_rank(docs), start=1):
            if rel.get(d, 0) > 0:
                rr = 1.0 / rank
                break
        per_q[qid] = {"recip_rank": rr}
    if agg:
        return sum(v["recip_rank"] for v in per_q.values()) / max(1, len(per_q))
    return per_q


def _recall(docs, rel, cut):
    n_rel = sum(1 for v in rel.values() if v > 0)
    if n_rel == 0:
        return 0.0
    got = sum(1 for d in _trec_rank(docs)[:cut] if rel.get(d, 0) > 0)
    return got / n_rel


def _ndcg(docs, rel, cut):
    gains = [rel.get(d, 0) for d in _trec_rank(docs)[:cut]]
    dcg = sum(g / math.log2(i + 2) for i, g in enumerate(gains) if g > 0)
    ideal = sorted((v for v in rel.values() if v > 0), reverse=True)[:cut]
    idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal))
    return dcg / idcg if idcg > 0 else 0.0


def evaluate(run, qrel, metric: str, agg: bool = True):
    fn, cuts = {"recall": (_recall, RECALL_CUTS), "ndcg_cut": (_ndcg, NDCG_CUTS)}[metric]
    per_q = {q: {f"{metric}_{c}": fn(docs, qrel[q], c) for c in cuts} for q, docs in run.items() if q in qrel}
    if not agg:
        return per_q
    n = max(1, len(per_q))
    return {f"{metric}_{c}": sum(v[f"{metric}_{c}"] for v in per_q.values()) / n for c in cuts}


def load_and_evaluate(qrel_file_path: str, run_file_path: str, metric: str):
    with open(qrel_file_path) as f:
        qrel = json.load(f)
    with open(run_file_path) as f:
        run = json.load(f)
    if "TREC" in qrel_file_path:
        assert ("binary" not in qrel_file_path) == (metric == "ndcg" or metric == "ndcg_cut")
    if metric == "mrr_10":
        res = mrr_k(run, qrel, k=10)
        print("MRR@10:", res)
        return {"mrr_10": res}
    res = evaluate(run, qrel, metric=metric)
    print(metric, "==>", res)
    return res
