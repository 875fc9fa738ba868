"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path (ripor_amd/*).

CPU restatement of the reference's trie-constrained beam search, with the reference's cost
profile kept on purpose (host-side dict + scipy-CSR mask in float64, top-2B, Python scorer loop,
optional full-prefix decoder recompute) so it can double as the ``cpu_baseline`` "port".

Follows:
  * reference t5_pretrainer/evaluate.py:410-424 and aq_preprocess/build_list_smtid_to_nextids.py:21-41
    -> :func:`build_list_smtid_to_nextids`
  * reference t5_pretrainer/tasks/generation.py:603-677 -> :class:`PrefixMaskRef` (and :class:`SortedPrefixMaskRef`, the same
    mask from the sorted code matrix for corpora whose dicts do not fit host RAM)
  * reference t5_pretrainer/tasks/generation.py:35-251 (wrapper: encoder once, expand xB, scorer)
    and :253-575 (loop) -> :func:`beam_search_ref`
  * third-party transformers==4.17.0 ``BeamSearchScorer.process/finalize`` (source not available
    offline; behaviour restated from SURVEY.md Appendix C: with eos_token_id=None the first B of
    the 2B sorted candidates are taken, hypotheses are only added in finalize with
    score = sum / len**length_penalty, sorted ascending (stable) and popped)
  * reference t5_pretrainer/evaluate.py:87-132 + utils/utils.py:46-59 -> :func:`constrained_decode_doc_ref`

Float semantics kept: logits float32; ``logits + (1-mask)*(-1e9) + beam_scores`` in float64
(the CSR mask is float64, SURVEY.md §3.2 Quirk B); ``sequences_scores`` float32.

Tie rule (torch.topk leaves it unspecified): candidates ordered by (score desc, flat index asc).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import scipy.sparse as sp
import torch


# ----------------------------------------------------------------------------- trie dicts
def build_list_smtid_to_nextids(docid_to_smtids: Dict[str, Sequence[int]]) -> List[Dict[str, List[int]]]:
    first = next(iter(docid_to_smtids.values()))
    levels: List[Dict[str, set]] = [dict() for _ in range(len(first) - 1)]
    for _docid, smtids in docid_to_smtids.items():
        for i in range(len(smtids) - 1):
            key = "_".join(str(x) for x in smtids[: i + 1])
            levels[i].setdefault(key, set()).add(int(smtids[i + 1]))
    return [{k: list(v) for k, v in lvl.items()} for lvl in levels]


def build_smtid_to_docids(docid_to_smtids: Dict[str, Sequence[int]], max_new_token: int) -> Dict[str, List[str]]:
    """reference evaluate.py:439-446."""
    out: Dict[str, List[str]] = {}
    for docid, smtids in docid_to_smtids.items():
        assert smtids[0] == -1, smtids
        sid = "_".join(str(x) for x in smtids[1 : 1 + max_new_token])
        out.setdefault(sid, []).append(docid)
    return out


class PrefixMaskRef:
    """One CSR matrix per level over a global prefix->row map; unknown prefixes give an
    all-zero row (reference generation.py:656-661,675)."""

    def __init__(self, list_smtid_to_nextids, vocab_size: int):
        self.vocab_size = vocab_size
        self.row_of: Dict[str, int] = {}
        self.mats = []
        for level in list_smtid_to_nextids:
            rows, cols = [], []
            for key, nxt in level.items():
                r = self.row_of.setdefault(key, len(self.row_of))
                rows.extend([r] * len(nxt))
                cols.extend(nxt)
            data = np.ones(len(rows), dtype=np.float64)
            self.mats.append(sp.csr_matrix((data, (rows, cols)), shape=(len(self.row_of), vocab_size)))

    def __call__(self, ids: np.ndarray) -> np.ndarray:
        R, T = ids.shape
        if T == 1:
            keys = ["-1"] * R
        else:
            keys = ["-1_" + "_".join(map(str, row[1:])) for row in ids.tolist()]
        idx = np.zeros(R, dtype=np.int64)
        unknown = []
        for i, k in enumerate(keys):
            r = self.row_of.get(k)
            if r is None:
                unknown.append(i)
            else:
                idx[i] = r
        mat = self.mats[T - 1]
        # a known prefix of a deeper level may index past this level's row count only if it is
        # not a level-(T-1) prefix, which cannot happen for a key of length T-1.
        mask = mat[idx].toarray()
        if unknown:
            mask[np.asarray(unknown)] = 0.0
        return mask  # float64 [R, V]


class SortedPrefixMaskRef:
    """The same function as :class:`PrefixMaskRef` — ``mask[r, v] = 1`` iff some document's code sequence starts with
    the prefix of row r followed by v, an unknown prefix gives an all-zero row (reference generation.py:656-661,675) —
    evaluated on the lexicographically sorted code matrix instead of the reference's dict of strings: the dicts of the
    8.8 M-document corpus do not fit host RAM (tests/test_gpu_fullsize.py), the matrix is 283 MB. A prefix is a row
    range, narrowed one ``np.searchsorted`` pair per new token and remembered per prefix (a beam extends its parent's
    prefix). Pinned to PrefixMaskRef by tests/test_oracle_golden.py::test_sorted_matrix_mask_equals_the_dict_mask.
    Codes < 256 (one byte per position)."""

    def __init__(self, codes: np.ndarray, vocab_size: int):
        codes = np.asarray(codes)
        assert codes.ndim == 2 and 0 <= int(codes.min()) and int(codes.max()) < 256
        self.vocab_size = vocab_size
        N, L = codes.shape
        Lp = (L + 7) // 8 * 8
        b = np.zeros((N, Lp), dtype=np.uint8)
        b[:, :L] = codes
        keys = b.view(">u8")                                   # [N, Lp / 8] big-endian words: word order = lexicographic order
        order = np.lexsort(tuple(keys[:, k] for k in range(keys.shape[1] - 1, -1, -1)))
        self.sorted = np.ascontiguousarray(b[order, :L])
        self.ranges = {(): (0, N)}
        self.next_tokens = {}

    def _range(self, prefix: tuple):
        r = self.ranges.get(prefix)
        if r is None:
            lo, hi = self._range(prefix[:-1])
            k = len(prefix) - 1
            if hi > lo and k < self.sorted.shape[1] and 0 <= prefix[-1] < 256:
                col = self.sorted[lo:hi, k]
                r = (lo + int(np.searchsorted(col, prefix[-1], "left")), lo + int(np.searchsorted(col, prefix[-1], "right")))
            else:
                r = (lo, lo)
            self.ranges[prefix] = r
        return r

    def __call__(self, ids: np.ndarray) -> np.ndarray:
        R, T = ids.shape
        mask = np.zeros((R, self.vocab_size), dtype=np.float64)
        if T - 1 >= self.sorted.shape[1]:
            return mask
        for i, row in enumerate(ids.tolist()):
            key = tuple(row[1:])
            nxt = self.next_tokens.get(key)
            if nxt is None:
                lo, hi = self._range(key)
                nxt = np.unique(self.sorted[lo:hi, T - 1]) if hi > lo else np.zeros(0, dtype=np.uint8)
                self.next_tokens[key] = nxt = nxt[nxt < self.vocab_size]
            mask[i, nxt] = 1.0
        return mask


# ----------------------------------------------------------------------------- scorer (HF 4.17 behaviour)
class _Hyps:
    def __init__(self, num_beams: int):
        self.num_beams = num_beams
        self.beams = []  # (score, ids)
        self.worst = 1e9

    def add(self, ids, sum_logprobs: float, length_penalty: float):
        score = sum_logprobs / (ids.shape[-1] ** length_penalty)
        if len(self.beams) < self.num_beams or score > self.worst:
            self.beams.append((score, ids))
            if len(self.beams) > self.num_beams:
                srt = sorted((s, i) for i, (s, _) in enumerate(self.beams))
                del self.beams[srt[0][1]]
                self.worst = srt[1][0]
            else:
                self.worst = min(score, self.worst)


def _process(next_scores, next_tokens, next_indices, Q: int, B: int):
    """eos_token_id is None, so no candidate is ever skipped: the first B of 2B are kept."""
    nb_scores = torch.zeros((Q, B), dtype=next_scores.dtype)
    nb_tokens = torch.zeros((Q, B), dtype=next_tokens.dtype)
    nb_idx = torch.zeros((Q, B), dtype=next_indices.dtype)
    for q in range(Q):
        slot = 0
        for rank in range(next_scores.shape[1]):
            nb_scores[q, slot] = next_scores[q, rank].item()
            nb_tokens[q, slot] = next_tokens[q, rank].item()
            nb_idx[q, slot] = q * B + next_indices[q, rank].item()
            slot += 1
            if slot == B:
                break
    return nb_scores.view(-1), nb_tokens.view(-1), nb_idx.view(-1)


def _finalize(ids: torch.Tensor, beam_scores: torch.Tensor, Q: int, B: int, keep: int,
              length_penalty: float = 1.0):
    T = ids.shape[1]
    seqs = torch.zeros((Q * keep, T), dtype=torch.long)
    best = torch.zeros(Q * keep, dtype=torch.float32)
    for q in range(Q):
        hyp = _Hyps(B)
        for b in range(B):
            hyp.add(ids[q * B + b], beam_scores[q * B + b].item(), length_penalty)
        srt = sorted(hyp.beams, key=lambda x: x[0])
        for j in range(keep):
            score, tok = srt.pop()
            best[q * keep + j] = score
            seqs[q * keep + j] = tok
    return seqs, best


# ----------------------------------------------------------------------------- the search
@torch.no_grad()
def beam_search_ref(model, mask_fn, input_ids, attention_mask, num_beams: int, max_new_tokens: int,
                    apply_log_softmax_for_scores: bool = False, use_kv_cache: bool = False,
                    record: Optional[dict] = None):
    """Returns ``(sequences int64 [Q*B, L+1], sequences_scores float32 [Q*B])``.

    ``model`` is an oracle.t5_ref.T5Ref (``use_kv_cache=False``: full-prefix recompute each step,
    like the reference) or T5RefCached (``use_kv_cache=True``).
    """
    input_ids = torch.as_tensor(input_ids, dtype=torch.long)
    attention_mask = torch.as_tensor(attention_mask, dtype=torch.long)
    Q, B, L = input_ids.shape[0], num_beams, max_new_tokens
    enc = model.encode(input_ids, attention_mask)
    if record is not None:
        record["encoder_out"] = enc.numpy().copy()
    enc = enc.repeat_interleave(B, dim=0)
    enc_mask = attention_mask.repeat_interleave(B, dim=0)
    ids = torch.zeros((Q * B, 1), dtype=torch.long)
    beam_scores = torch.zeros((Q, B), dtype=torch.float32)
    beam_scores[:, 1:] = -1e9
    beam_scores = beam_scores.view(-1)
    if use_kv_cache:
        model.start(enc, enc_mask)
    steps = []
    while True:
        if use_kv_cache:
            logits = model.step(ids[:, -1] if ids.shape[1] > 1 else None, Q * B)
        else:
            logits = model.last_logits(ids, enc, enc_mask)
        scores = torch.log_softmax(logits, dim=-1) if apply_log_softmax_for_scores else logits
        mask = torch.tensor(mask_fn(ids.numpy()))  # float64
        processed = scores + (1.0 - mask) * (-1e9)  # -> float64
        nxt = processed + beam_scores[:, None].expand_as(processed)
        V = nxt.shape[-1]
        nxt = nxt.view(Q, B * V)
        srt_scores, srt_idx = torch.sort(nxt, dim=1, descending=True, stable=True)
        top_scores, top_idx = srt_scores[:, : 2 * B], srt_idx[:, : 2 * B]
        top_beam = torch.div(top_idx, V, rounding_mode="floor")
        top_tok = top_idx % V
        if record is not None:
            steps.append({"logits": logits.numpy().copy(), "top_scores": top_scores.numpy().copy(),
                          "top_beam": top_beam.numpy().copy(), "top_tok": top_tok.numpy().copy()})
        beam_scores, beam_tok, beam_idx = _process(top_scores, top_tok, top_beam, Q, B)
        ids = torch.cat([ids[beam_idx, :], beam_tok.unsqueeze(-1)], dim=-1)
        if use_kv_cache:
            model.reorder(beam_idx)
        if ids.shape[-1] >= L + 1:
            break
    seqs, seq_scores = _finalize(ids, beam_scores, Q, B, B)
    if record is not None:
        record["steps"] = steps
    return seqs, seq_scores


# ----------------------------------------------------------------------------- caller
def smtid_strings(sequences: torch.Tensor, B: int, L: int) -> List[List[str]]:
    """reference utils/utils.py:46-59 on ``sequences.view(-1, B, L+1)``."""
    seq = torch.as_tensor(sequences).view(-1, B, L + 1).tolist()
    return [["_".join(str(x) for x in s[1:]) for s in beams] for beams in seq]


def constrained_decode_doc_ref(qids, sequences, sequences_scores, smtid_to_docids, B: int, L: int,
                               apply_log_softmax_for_scores: bool = False):
    """reference evaluate.py:115-128: smtid -> every docid under it gets float(score_f32)*L."""
    out = {}
    strs = smtid_strings(sequences, B, L)
    scores = torch.as_tensor(sequences_scores).view(-1, B).tolist()
    for qid, ranked, rel in zip(qids, strs, scores):
        out[qid] = {}
        for smtid, s in zip(ranked, rel):
            if smtid not in smtid_to_docids:
                continue
            for docid in smtid_to_docids[smtid]:
                out[qid][docid] = s if apply_log_softmax_for_scores else s * L
    return out
