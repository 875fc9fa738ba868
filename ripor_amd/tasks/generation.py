"""Drop-in for the reference's constrained prefix beam search entry points.

Mirrors reference t5_pretrainer/tasks/generation.py:
  * ``generate_for_constrained_prefix_beam_search`` (:35-251) — same name, keyword arguments and
    return attributes (``.sequences`` LongTensor ``[Q*B', L+1]`` with column 0 = start id 0, rows
    grouped per query best-first; ``.sequences_scores`` float32 ``[Q*B']`` = sum(step scores)/(L+1)),
    same ValueErrors for inconsistent beam arguments. The whole loop (:253-575) runs inside
    ``rpr_search`` of libripor_hip.so — no per-step host work.
  * ``PrefixConstrainLogitProcessorFastSparse`` (:603-677) — same constructor
    ``(list_smtid_to_nextids, vocab_size)`` and ``__call__(input_ids, scores) -> mask``; instead of
    per-level scipy CSR matrices over string keys it owns a device trie (sorted code matrix).
    ``from_codes`` / ``from_docid_to_smtid`` build the same object straight from the docid table
    without materialising the dicts (which need tens of GB for 8.8 M docs).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import engine as E


class BeamSearchEncoderDecoderOutput:
    """The reference's return object (HF 4.17 ``BeamSearchEncoderDecoderOutput``, generation.py:554-564).

    ``sequences`` / ``sequences_scores`` are what every caller on the path reads (evaluate.py:76-77, :116-117, :164-165).
    ``scores`` (tuple of L float64 tensors ``[Q*B, V]``: the processed scores of every step, logits or log-probabilities
    plus ``(1 - valid_mask) * (-1e9)``, reference :453-468) and ``beam_indices`` (per returned beam slot the tuple of
    the L parent indices ``q*B + slot`` it descended from, :521-522, :548-552) are produced ON FIRST ACCESS: the search
    proper never forms V logits per step behind a fork, so reading either attribute runs one more search of the same
    batch step by step with the library's debug taps (no forks, no lanes, eager) and converts its per-step record.
    ``None`` when the call was made without ``output_scores``."""

    _LAZY = ("scores", "beam_indices")

    def __init__(self, sequences=None, sequences_scores=None, scores=None, beam_indices=None, encoder_attentions=None,
                 encoder_hidden_states=None, decoder_attentions=None, cross_attentions=None, decoder_hidden_states=None,
                 row_lo=None, row_hi=None, guard=None, _step_record=None):
        self.sequences, self.sequences_scores = sequences, sequences_scores
        self._scores, self._beam_indices = scores, beam_indices
        self.encoder_attentions, self.encoder_hidden_states = encoder_attentions, encoder_hidden_states
        self.decoder_attentions, self.cross_attentions = decoder_attentions, cross_attentions
        self.decoder_hidden_states = decoder_hidden_states
        # extras of this implementation: sorted-row range of every returned smtid (docids = perm[lo:hi])
        self.row_lo, self.row_hi = row_lo, row_hi
        # with defer_status=True: the E.GuardedSearch whose result() must be consulted before the tensors above are trusted
        self.guard = guard
        self._step_record = _step_record      # callable -> (scores tuple, beam_indices tuple), or None

    def _materialise(self):
        if self._step_record is not None:
            self._scores, self._beam_indices = self._step_record()
            self._step_record = None

    @property
    def scores(self):
        self._materialise()
        return self._scores

    @property
    def beam_indices(self):
        self._materialise()
        return self._beam_indices

    def __getitem__(self, k):
        return getattr(self, k)


def _per_step_record(em, trie, input_ids, attention_mask, B, L, K, log_softmax):
    """``scores`` and ``beam_indices`` of the reference's output object from one tapped search (see the class above)."""
    ctx = em.ctx
    saved = ctx.get_precision()
    # the sticky words of the shared ctx may hold flags of a search nobody has checked yet (deferred guards): read and clear
    # them before the tapped run (a stale SATURATED bit would trigger a needless fp32 rerun), hand them back afterwards
    before = ctx.status(clear=True)
    try:
        res = E.search(em, trie, input_ids, attention_mask, B, L, apply_log_softmax_for_scores=log_softmax, taps=True)
        torch.cuda.synchronize(ctx.device)
        if ctx.status(clear=True) & E._lib.STATUS_SATURATED and saved != "f32":
            ctx.set_precision("f32")
            try:
                res = E.search(em, trie, input_ids, attention_mask, B, L, apply_log_softmax_for_scores=log_softmax, taps=True)
                torch.cuda.synchronize(ctx.device)
                ctx.status(clear=True)
            finally:
                ctx.set_precision(saved)
    finally:
        ctx.keep_status(before)
    t = res.taps
    Q, V = input_ids.shape[0], em.V
    logits = t["step_logits"].view(L, Q * B, V)                                  # fp32, row = q*B + slot of the step's beams
    words = t["step_valid"].view(L, Q * B, V // 64)                              # bit (token % 64) of word token // 64
    bits = (words.unsqueeze(-1) >> torch.arange(64, device=words.device, dtype=torch.int64)) & 1
    valid = bits.view(L, Q * B, V).to(torch.float64)
    base = torch.log_softmax(logits, dim=-1) if log_softmax else logits          # fp32 like the reference (:453-455)
    scores = tuple((base[s].to(torch.float64) + (1.0 - valid[s]) * (-1e9)) for s in range(L))
    parent = t["step_parent"].view(L, Q, B).cpu().numpy()                        # slot of the previous step every new slot came from
    hist = [[() for _ in range(B)] for _ in range(Q)]
    for s in range(L):
        hist = [[hist[q][int(parent[s, q, b])] + (q * B + int(parent[s, q, b]),) for b in range(B)] for q in range(Q)]
    beam_indices = tuple(hist[q][b] for q in range(Q) for b in range(K))          # first K slots of every query (:548-552)
    return scores, beam_indices


class PrefixConstrainLogitProcessorFastSparse:
    def __init__(self, list_smtid_to_nextids: Optional[Sequence[Dict[str, List[int]]]], vocab_size: int,
                 _codes: Optional[np.ndarray] = None, _trie_cache: Optional[str] = None):
        self.vocab_size = int(vocab_size)
        self.list_smtid_to_next_smtids = list_smtid_to_nextids
        self._tries: Dict[int, E.DeviceTrie] = {}
        self._trie_cache = _trie_cache
        if _trie_cache is not None and _codes is None and list_smtid_to_nextids is None:
            # binary cache written by aq_preprocess.build_list_smtid_to_nextids (sorted codes + permutation + docids):
            # nothing is parsed or sorted here, trie() loads the file per device (rpr_trie_load validates it)
            info = E.trie_file_info(_trie_cache)
            if info["V"] > self.vocab_size:
                raise ValueError("smtid token >= vocab_size")
            self.codes = None
            self.max_len = info["L"]
            return
        if _codes is None:
            _codes = self._codes_from_dicts(list_smtid_to_nextids)
        self.codes = np.ascontiguousarray(_codes, dtype=np.uint16)
        if self.codes.ndim != 2 or self.codes.shape[0] == 0:
            raise ValueError("empty docid code matrix")
        if int(self.codes.max()) >= self.vocab_size:
            raise ValueError("smtid token >= vocab_size")
        self.max_len = self.codes.shape[1]

    # -- constructors -------------------------------------------------------------------------
    @classmethod
    def from_codes(cls, codes: np.ndarray, vocab_size: int):
        """codes ``[N, L]``: row i = smtid of docid index i."""
        return cls(None, vocab_size, _codes=codes)

    @classmethod
    def from_trie_cache(cls, path: str, vocab_size: int):
        """The binary cache next to ``docid_to_smtid.json`` (replaces the reference's ``list_smtid_to_nextids.pkl``,
        evaluate.py:404-408): see :func:`ripor_amd.engine.build_trie_file`."""
        return cls(None, vocab_size, _trie_cache=path)

    @classmethod
    def from_docid_to_smtid(cls, docid_to_smtids: Dict[str, Sequence[int]], vocab_size: int):
        """The reference's ``docid_to_smtid.json`` content ``{"docid": [-1, c1..cL]}``; returns
        ``(processor, docids)`` with ``docids[i]`` the docid of code row i (file order)."""
        docids = list(docid_to_smtids.keys())
        first = docid_to_smtids[docids[0]]
        assert first[0] == -1, first  # reference evaluate.py:441
        codes = np.asarray([docid_to_smtids[d][1:] for d in docids], dtype=np.int64)
        return cls(None, vocab_size, _codes=codes), docids

    @staticmethod
    def _codes_from_dicts(levels) -> np.ndarray:
        """Enumerate the root-to-leaf paths of the reference's per-level dicts
        (``levels[l]["-1_c1_.._cl"] = [next ids]``, reference evaluate.py:410-424)."""
        if not levels:
            raise ValueError("list_smtid_to_nextids is empty")
        L = len(levels)
        paths: List[List[int]] = []
        stack = [("-1", [])]
        while stack:
            key, pref = stack.pop()
            depth = len(pref)
            if depth == L:
                paths.append(pref)
                continue
            for nid in levels[depth].get(key, []):
                stack.append((key + "_" + str(int(nid)), pref + [int(nid)]))
        if not paths:
            raise ValueError("list_smtid_to_nextids has no complete smtid")
        return np.asarray(paths, dtype=np.int64)

    # -- device trie ----------------------------------------------------------------------------
    def trie(self, device) -> "E.DeviceTrie":
        ctx = E.Context.get(device)
        idx = ctx.device.index
        if idx not in self._tries:
            if self.codes is None:
                self._tries[idx] = E.DeviceTrie.load(ctx, self._trie_cache, V=self.vocab_size)
            else:
                self._tries[idx] = E.DeviceTrie.from_codes(ctx, self.codes, self.vocab_size)
        return self._tries[idx]

    def docid_rows(self, device, lo: int, hi: int) -> np.ndarray:
        """Original code-row indices (docid indices) under the sorted range [lo, hi)."""
        return self.trie(device).perm[lo:hi]

    def __call__(self, input_ids: torch.Tensor, next_token_scores: torch.Tensor = None) -> torch.Tensor:
        """valid_mask ``[R, vocab_size]`` float64 (reference :666-677), computed by the device kernel."""
        dev = input_ids.device if input_ids.is_cuda else torch.device("cuda", torch.cuda.current_device())
        mask = self.trie(dev).mask(input_ids.detach().cpu().numpy())
        return torch.from_numpy(mask.astype(np.float64)).to(input_ids.device)


def generate_for_constrained_prefix_beam_search(
        model, valid_smtids, inputs: Optional[torch.Tensor] = None, max_length: Optional[int] = None,
        min_length=None, do_sample=None, early_stopping=None, num_beams: Optional[int] = None, temperature=None,
        top_k=None, top_p=None, typical_p=None, repetition_penalty=None, bad_words_ids=None, bos_token_id=None,
        pad_token_id=None, eos_token_id=None, length_penalty=None, no_repeat_ngram_size=None,
        encoder_no_repeat_ngram_size=None, num_return_sequences: Optional[int] = None, max_time=None,
        max_new_tokens: Optional[int] = None, decoder_start_token_id=None, use_cache=None, num_beam_groups=None,
        diversity_penalty=None, prefix_allowed_tokens_fn=None, logits_processor=None, stopping_criteria=None,
        constraints=None, output_attentions=None, output_hidden_states=None, output_scores=None,
        return_dict_in_generate=None, forced_bos_token_id=None, forced_eos_token_id=None,
        remove_invalid_values=None, synced_gpus: Optional[bool] = False,
        apply_log_softmax_for_scores: Optional[bool] = False, apply_prefix_tree: Optional[bool] = False,
        **model_kwargs):
    model.eval()
    input_ids = model_kwargs.pop("input_ids", inputs)
    attention_mask = model_kwargs.pop("attention_mask", None)
    if input_ids is None:
        raise ValueError("`input_ids` has to be defined.")
    if attention_mask is None:  # HF _prepare_attention_mask_for_generation with pad_token_id = 0
        attention_mask = (input_ids != 0).long()
    num_beams = 1 if num_beams is None else int(num_beams)
    num_return_sequences = 1 if num_return_sequences is None else int(num_return_sequences)
    num_beam_groups = 1 if num_beam_groups is None else int(num_beam_groups)
    if num_beam_groups > num_beams:
        raise ValueError("`num_beam_groups` has to be smaller or equal to `num_beams`")
    if num_beam_groups != 1 or do_sample or constraints is not None:
        raise ValueError("only plain beam search (num_beam_groups=1, do_sample=False, no constraints) is supported")
    if num_return_sequences > num_beams:
        raise ValueError("`num_return_sequences` has to be smaller or equal to `num_beams`.")
    # decoder prompt is the single start id, so max_length = max_new_tokens + 1 (reference :153-154)
    if max_length is None and max_new_tokens is not None:
        max_length = int(max_new_tokens) + 1
    if max_length is None:
        raise ValueError("`max_length` needs to be a stopping_criteria for now.")
    L = int(max_length) - 1
    if L < 1:
        raise ValueError(f"max_new_tokens must be >= 1, got {L}")
    if not isinstance(valid_smtids, PrefixConstrainLogitProcessorFastSparse):
        raise TypeError("valid_smtids must be a PrefixConstrainLogitProcessorFastSparse")

    em = model.engine_model()
    trie = valid_smtids.trie(model.device)
    # Guards (E.GuardedSearch): the split-precision GEMMs carry activations in f16 planes — a value outside their range is
    # clamped and flagged on the device and the batch is then repeated on the exact-fp32 MFMA path instead of returning
    # rankings computed from clipped tensors; the forced-tail search runs in its optimistic mode and is repeated exactly
    # when a query was left unforced at the last fork. The reference call is synchronous (>= 2*B*Q syncs per step) and so
    # is this one by default; with defer_status=True the check happens when the caller asks for `outputs.guard.result()`
    # (ripor_amd/evaluate.py: while the next batch runs).
    guard = E.search_guarded(em, trie, input_ids, attention_mask, num_beams, L,
                             apply_log_softmax_for_scores=bool(apply_log_softmax_for_scores))
    defer = bool(model_kwargs.pop("defer_status", False)) and bool(return_dict_in_generate)
    res = guard._res if defer else guard.result()
    Q, B, K = input_ids.shape[0], num_beams, num_return_sequences
    tok = res.tokens[:, :K, :].to(torch.long)
    seqs = torch.cat([torch.zeros((Q, K, 1), dtype=torch.long, device=tok.device), tok], dim=2).reshape(Q * K, L + 1)
    if not return_dict_in_generate:
        return seqs
    record = None
    if output_scores and em.V % 64 == 0:   # (the debug taps need a vocab size on the 64 grid)
        ids_keep, mask_keep, ls = input_ids, attention_mask, bool(apply_log_softmax_for_scores)
        record = lambda: _per_step_record(em, trie, ids_keep, mask_keep, B, L, K, ls)   # noqa: E731  (run on first access)
    elif output_scores:
        def record():   # the reference always returns them: fail loudly on access instead of handing out None
            raise NotImplementedError(f".scores / .beam_indices need a decoder vocab size that is a multiple of 64 (got {em.V}): the "
                                      "per-step taps of rpr_search are laid out on 64-token words; .sequences and .sequences_scores "
                                      "do not depend on it")
    return BeamSearchEncoderDecoderOutput(
        sequences=seqs,
        sequences_scores=res.scores[:, :K].reshape(Q * K) if output_scores else None,
        row_lo=res.row_lo[:, :K].reshape(Q * K), row_hi=res.row_hi[:, :K].reshape(Q * K), guard=guard if defer else None,
        _step_record=record)
