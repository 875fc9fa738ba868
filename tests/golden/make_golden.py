#!/usr/bin/env python3
"""Generate golden vectors by running the *imported* reference (/root/reference) on CPU.

Runs ONLY in the build container (needs /root/reference); the GPU box and the test-suite use
the committed ``*.npz`` outputs. No reference source is copied: the reference modules are
imported in place through a compatibility shim (transformers 5.15 here vs the reference's pinned
4.17.0), and what is stored is data: the seeds/dims needed to regenerate
the inputs with ripor_amd.utils.synth plus the reference's outputs.

What runs unmodified from the reference:
  t5_pretrainer.modeling.t5_generative_retriever.T5ForDocIDGeneration (forward, embeds, logits)
  t5_pretrainer.tasks.generation.beam_search_for_constrained_prefix   (the beam loop)
  t5_pretrainer.tasks.generation.PrefixConstrainLogitProcessorFastSparse
  t5_pretrainer.utils.utils.convert_ptsmtids_to_strsmtid
What is restated here because transformers 4.17 is not installed: BeamSearchScorer (behaviour per
SURVEY.md Appendix C), stopping criteria / logits-processor containers, and the generate()
wrapper's prologue (encoder once, repeat_interleave(B), decoder_start ids = 0, MaxLength(L+1)).

Usage:  python tests/golden/make_golden.py [--only NAME]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from ripor_amd.utils import synth  # noqa: E402


# ----------------------------------------------------------------------------- shim
def install_shim():
    import transformers
    import transformers.pytorch_utils as pu
    from transformers.modeling_outputs import BaseModelOutput
    from transformers.utils import ModelOutput
    import transformers.models.t5.modeling_t5 as mt5

    uj = types.ModuleType("ujson")
    uj.load, uj.loads, uj.dump, uj.dumps = json.load, json.loads, json.dump, json.dumps
    sys.modules["ujson"] = uj

    mpu = types.ModuleType("transformers.utils.model_parallel_utils")
    mpu.assert_device_map = lambda *a, **k: None
    mpu.get_device_map = lambda *a, **k: None
    sys.modules["transformers.utils.model_parallel_utils"] = mpu

    pu.torch_int_div = lambda a, b: torch.div(a, b, rounding_mode="floor")

    Orig = mt5.T5Stack

    class T5StackCompat(Orig):
        def __init__(self, config, embed_tokens=None):
            super().__init__(config)
            if embed_tokens is not None:
                self.embed_tokens = embed_tokens

        def forward(self, *args, head_mask=None, cross_attn_head_mask=None, output_attentions=None,
                    output_hidden_states=None, return_dict=None, **kw):
            return super().forward(*args, **kw)

    mt5.T5Stack = T5StackCompat

    # --- transformers.generation_beam_search (HF 4.17 behaviour, restated) ---
    class BeamHypotheses:
        def __init__(self, num_beams, length_penalty, early_stopping):
            self.length_penalty, self.early_stopping, self.num_beams = length_penalty, early_stopping, num_beams
            self.beams, self.worst_score = [], 1e9

        def __len__(self):
            return len(self.beams)

        def add(self, hyp, sum_logprobs):
            score = sum_logprobs / (hyp.shape[-1] ** self.length_penalty)
            if len(self) < self.num_beams or score > self.worst_score:
                self.beams.append((score, hyp))
                if len(self) > self.num_beams:
                    srt = sorted([(s, i) for i, (s, _) in enumerate(self.beams)])
                    del self.beams[srt[0][1]]
                    self.worst_score = srt[1][0]
                else:
                    self.worst_score = min(score, self.worst_score)

        def is_done(self, best_sum_logprobs, cur_len):
            if len(self) < self.num_beams:
                return False
            if self.early_stopping:
                return True
            return self.worst_score >= best_sum_logprobs / cur_len ** self.length_penalty

    class BeamScorer:
        pass

    class BeamSearchScorer(BeamScorer):
        def __init__(self, batch_size, num_beams, device, length_penalty=1.0, do_early_stopping=False,
                     num_beam_hyps_to_keep=1, num_beam_groups=1, **kw):
            self.num_beams, self.device = num_beams, device
            self.length_penalty, self.do_early_stopping = length_penalty, do_early_stopping
            self.num_beam_hyps_to_keep, self.num_beam_groups = num_beam_hyps_to_keep, num_beam_groups
            self.group_size = num_beams // num_beam_groups
            self._beam_hyps = [BeamHypotheses(num_beams, length_penalty, do_early_stopping)
                               for _ in range(batch_size)]
            self._done = torch.tensor([False] * batch_size, dtype=torch.bool, device=device)
            if not isinstance(num_beams, int) or num_beams <= 1:
                if not ALLOW_SINGLE_BEAM:
                    raise ValueError("`num_beams` has to be an integer strictly greater than 1")

        @property
        def is_done(self):
            return bool(self._done.all())

        def process(self, input_ids, next_scores, next_tokens, next_indices, pad_token_id=None, eos_token_id=None):
            cur_len = input_ids.shape[-1]
            batch_size = len(self._beam_hyps)
            nbs = torch.zeros((batch_size, self.group_size), dtype=next_scores.dtype, device=self.device)
            nbt = torch.zeros((batch_size, self.group_size), dtype=next_tokens.dtype, device=self.device)
            nbi = torch.zeros((batch_size, self.group_size), dtype=next_indices.dtype, device=self.device)
            for b, hyp in enumerate(self._beam_hyps):
                if self._done[b]:
                    nbs[b, :], nbt[b, :], nbi[b, :] = 0, pad_token_id, 0
                    continue
                k = 0
                for rank, (tok, sc, idx) in enumerate(zip(next_tokens[b], next_scores[b], next_indices[b])):
                    bb = b * self.group_size + idx
                    if (eos_token_id is not None) and (tok.item() == eos_token_id):
                        if rank >= self.group_size:
                            continue
                        hyp.add(input_ids[bb].clone(), sc.item())
                    else:
                        nbs[b, k], nbt[b, k], nbi[b, k] = sc, tok, bb
                        k += 1
                    if k == self.group_size:
                        break
                if k < self.group_size:
                    raise ValueError("not enough non-eos candidates")
                self._done[b] = self._done[b] or hyp.is_done(next_scores[b].max().item(), cur_len)
            return {"next_beam_scores": nbs.view(-1), "next_beam_tokens": nbt.view(-1),
                    "next_beam_indices": nbi.view(-1)}

        def finalize(self, input_ids, final_beam_scores, final_beam_tokens, final_beam_indices, max_length,
                     pad_token_id=None, eos_token_id=None):
            batch_size = len(self._beam_hyps)
            for b, hyp in enumerate(self._beam_hyps):
                if self._done[b]:
                    continue
                for beam_id in range(self.num_beams):
                    bb = b * self.num_beams + beam_id
                    hyp.add(input_ids[bb], final_beam_scores[bb].item())
            keep = self.num_beam_hyps_to_keep
            sent_lengths = input_ids.new(batch_size * keep)
            best, best_scores = [], torch.zeros(batch_size * keep, device=self.device, dtype=torch.float32)
            for i, hyp in enumerate(self._beam_hyps):
                srt = sorted(hyp.beams, key=lambda x: x[0])
                for j in range(keep):
                    score, h = srt.pop()
                    sent_lengths[keep * i + j] = len(h)
                    best.append(h)
                    best_scores[i * keep + j] = score
            sent_max_len = min(sent_lengths.max().item() + 1, max_length)
            decoded = input_ids.new(batch_size * keep, sent_max_len)
            if sent_lengths.min().item() != sent_lengths.max().item():
                assert pad_token_id is not None
                decoded.fill_(pad_token_id)
            for i, h in enumerate(best):
                decoded[i, : sent_lengths[i]] = h
                if sent_lengths[i] < max_length:
                    decoded[i, sent_lengths[i]] = eos_token_id
            return {"sequences": decoded, "sequence_scores": best_scores}

    gbs = types.ModuleType("transformers.generation_beam_search")
    gbs.BeamScorer, gbs.BeamSearchScorer, gbs.ConstrainedBeamSearchScorer = BeamScorer, BeamSearchScorer, object
    sys.modules["transformers.generation_beam_search"] = gbs

    @dataclass
    class BeamSearchEncoderDecoderOutput(ModelOutput):
        sequences: torch.LongTensor = None
        sequences_scores: Optional[torch.FloatTensor] = None
        scores: Optional[Tuple[torch.FloatTensor]] = None
        beam_indices: Optional[Tuple] = None
        encoder_attentions: Optional[Tuple] = None
        encoder_hidden_states: Optional[Tuple] = None
        decoder_attentions: Optional[Tuple] = None
        cross_attentions: Optional[Tuple] = None
        decoder_hidden_states: Optional[Tuple] = None

    gu = types.ModuleType("transformers.generation_utils")
    for nm in ["GreedySearchOutput", "SampleOutput", "BeamSearchOutput", "BeamSampleOutput",
               "BeamSearchDecoderOnlyOutput", "GreedySearchEncoderDecoderOutput",
               "GreedySearchDecoderOnlyOutput", "SampleEncoderDecoderOutput", "SampleDecoderOnlyOutput",
               "BeamSampleEncoderDecoderOutput", "BeamSampleDecoderOnlyOutput"]:
        setattr(gu, nm, object)
    gu.BeamSearchEncoderDecoderOutput = BeamSearchEncoderDecoderOutput
    sys.modules["transformers.generation_utils"] = gu

    class LogitsProcessor:
        pass

    class LogitsProcessorList(list):
        def __call__(self, input_ids, scores, **kw):
            for p in self:
                scores = p(input_ids, scores)
            return scores

    glp = types.ModuleType("transformers.generation_logits_process")
    glp.LogitsProcessor, glp.LogitsProcessorList = LogitsProcessor, LogitsProcessorList
    for nm in ["EncoderNoRepeatNGramLogitsProcessor", "ExponentialDecayLengthPenalty",
               "ForcedBOSTokenLogitsProcessor", "ForcedEOSTokenLogitsProcessor", "HammingDiversityLogitsProcessor",
               "InfNanRemoveLogitsProcessor", "MinLengthLogitsProcessor", "NoBadWordsLogitsProcessor",
               "NoRepeatNGramLogitsProcessor", "PrefixConstrainedLogitsProcessor",
               "RepetitionPenaltyLogitsProcessor", "TemperatureLogitsWarper", "TopKLogitsWarper",
               "TopPLogitsWarper", "TypicalLogitsWarper", "LogitNormalization"]:
        setattr(glp, nm, object)
    sys.modules["transformers.generation_logits_process"] = glp

    class MaxLengthCriteria:
        def __init__(self, max_length):
            self.max_length = max_length

        def __call__(self, input_ids, scores, **kw):
            return input_ids.shape[-1] >= self.max_length

    class MaxTimeCriteria:
        pass

    class StoppingCriteria:
        pass

    class StoppingCriteriaList(list):
        def __call__(self, input_ids, scores, **kw):
            return any(c(input_ids, scores) for c in self)

        @property
        def max_length(self):
            for c in self:
                if isinstance(c, MaxLengthCriteria):
                    return c.max_length
            return None

    gsc = types.ModuleType("transformers.generation_stopping_criteria")
    gsc.MaxLengthCriteria, gsc.MaxTimeCriteria = MaxLengthCriteria, MaxTimeCriteria
    gsc.StoppingCriteria, gsc.StoppingCriteriaList = StoppingCriteria, StoppingCriteriaList
    gsc.validate_stopping_criteria = lambda sc, ml: sc
    sys.modules["transformers.generation_stopping_criteria"] = gsc

    gbc = types.ModuleType("transformers.generation_beam_constraints")
    gbc.Constraint = object
    gbc.DisjunctiveConstraint = object
    gbc.PhrasalConstraint = object
    sys.modules["transformers.generation_beam_constraints"] = gbc
    return BaseModelOutput, BeamSearchScorer, StoppingCriteriaList, MaxLengthCriteria


ALLOW_SINGLE_BEAM = False


def load_reference():
    shim = install_shim()
    # the reference's t5_pretrainer is a namespace package (no __init__.py); this repo's alias package
    # of the same name is a regular package and would win regardless of path order, so drop the repo
    # root from sys.path (ripor_amd.utils.synth is already imported) before importing the reference
    sys.path[:] = [p_ for p_ in sys.path if os.path.realpath(p_ or ".") != os.path.realpath(REPO)]
    for k_ in [k_ for k_ in sys.modules if k_ == "t5_pretrainer" or k_.startswith("t5_pretrainer.")]:
        del sys.modules[k_]
    sys.path.insert(0, REF)
    os.chdir(REF)  # decoder_start_token_path default is relative (t5_generative_retriever.py:51)
    import importlib
    # importing t5_pretrainer.tasks.generation pulls transformers.* names through the shim
    gen = importlib.import_module("t5_pretrainer.tasks.generation")
    mod = importlib.import_module("t5_pretrainer.modeling.t5_generative_retriever")
    utils = importlib.import_module("t5_pretrainer.utils.utils")
    for m_ in (gen, mod, utils):  # the golden vectors must come from the reference, never from this repo
        assert os.path.realpath(m_.__file__).startswith(REF + os.sep), m_.__file__
    M = mod.T5ForDocIDGeneration
    M.adjust_logits_during_generation = lambda self, logits, **kw: logits

    def _upd(self, outputs, model_kwargs, is_encoder_decoder=False):
        model_kwargs["past"] = None
        return model_kwargs

    M._update_model_kwargs_for_generation = _upd
    return gen, mod, utils, shim


def build_reference_model(mod, dims: synth.ModelDims, sd_np, which="t5-base"):
    cfg = mod.T5forDocIDConfig(
        vocab_size=dims.vocab_size, d_model=dims.d_model, d_kv=dims.d_kv, d_ff=dims.d_ff,
        num_layers=dims.num_layers, num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
        relative_attention_num_buckets=dims.relative_attention_num_buckets,
        relative_attention_max_distance=dims.relative_attention_max_distance,
        decoder_vocab_sizes=list(dims.decoder_vocab_sizes), decoding=True,
        shared_output_input_embeds=dims.shared_output_input_embeds,
        scaleup_output_hidden=dims.scaleup_output_hidden,
        decoder_start_token_path=f"./t5_decoder_start_token_embeds/{which}.npy",
        feed_forward_proj="relu", dropout_rate=0.0, use_cache=False,
        decoder_start_token_id=0, pad_token_id=0, eos_token_id=1,
    )
    cfg._attn_implementation = "eager"
    model = mod.T5ForDocIDGeneration(cfg)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    sd["encoder.embed_tokens.weight"] = sd["shared.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "decoder.embed_tokens" not in m]  # 5.x-only unused table
    assert not missing and not unexpected, (missing, unexpected)
    model.eval()
    return model


@torch.no_grad()
def run_reference(gen, utils, shim, model, processor, input_ids, attention_mask, B, L, log_softmax=False):
    BaseModelOutput, BeamSearchScorer, StoppingCriteriaList, MaxLengthCriteria = shim
    input_ids = torch.as_tensor(input_ids, dtype=torch.long)
    attention_mask = torch.as_tensor(attention_mask, dtype=torch.long)
    Q = input_ids.shape[0]
    enc = model.encoder(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
    scorer = BeamSearchScorer(batch_size=Q, num_beams=B, device=torch.device("cpu"), length_penalty=1.0,
                              do_early_stopping=False, num_beam_hyps_to_keep=B)
    out = gen.beam_search_for_constrained_prefix(
        model, processor, torch.zeros((Q * B, 1), dtype=torch.long), scorer,
        stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(L + 1)]),
        output_scores=True, return_dict_in_generate=True,
        apply_log_softmax_for_scores=log_softmax,
        encoder_outputs=BaseModelOutput(last_hidden_state=enc.repeat_interleave(B, 0)),
        attention_mask=attention_mask.repeat_interleave(B, 0), use_cache=False)
    run_reference.last_beam_indices = np.asarray([list(map(int, bi)) for bi in out.beam_indices], dtype=np.int64)   # [Q*B, L]
    strs = utils.convert_ptsmtids_to_strsmtid(out.sequences.view(-1, B, L + 1), L)
    # per-step processed scores (float64) -> per-step sorted top-(B+1) margins are derived by tests
    step_scores = np.stack([s.numpy() for s in out.scores])  # [L, Q*B, V] float64 (without beam score)
    return enc.numpy(), out.sequences.numpy(), out.sequences_scores.numpy(), strs, step_scores


def replay_top(step_scores, Q, B, V):
    """Per step and query: the reference's sorted top-(B+1) cumulative candidates (float64 score, flat index
    beam*V + token), recomputed from its per-step processed scores exactly as generation.py:457-492 combines
    them (processed + beam_scores, beam_scores init [0, -1e9, ...] float32, :418-420). Rank B (0-based) is the
    first candidate that did NOT become a beam: tests use the gap between ranks B-1 and B as the pruning margin."""
    L = step_scores.shape[0]
    K = min(B + 1, B * V)
    beam = np.zeros((Q, B), dtype=np.float64)
    beam[:, 1:] = np.float32(-1e9)
    ts = np.zeros((L, Q, K), dtype=np.float64)
    ti = np.zeros((L, Q, K), dtype=np.int32)
    for t in range(L):
        cand = (step_scores[t].reshape(Q, B, V) + beam[:, :, None]).reshape(Q, B * V)
        order = np.argsort(-cand, axis=1, kind="stable")[:, :K]
        ts[t] = np.take_along_axis(cand, order, axis=1)
        ti[t] = order
        beam = ts[t][:, :B]
    return ts, ti


def reference_trie(gen, codes):
    """Build the reference's own structures from the synthetic code matrix, with the reference's
    dict-building loop restated (evaluate.py:410-424 cannot be imported: top-level `import faiss`)."""
    d2s = synth.codes_to_docid_to_smtid(codes)
    L = codes.shape[1]
    lst = [dict() for _ in range(L)]
    for _docid, smtids in d2s.items():
        for i in range(len(smtids) - 1):
            key = "_".join(str(x) for x in smtids[: i + 1])
            lst[i].setdefault(key, set()).add(int(smtids[i + 1]))
    lst = [{k: list(v) for k, v in lvl.items()} for lvl in lst]
    return d2s, lst


# ----------------------------------------------------------------------------- fixtures
CASES = {
    # name: (dims factory kwargs, N docs, Q, B, L, V, seed, extras)
    "g1_mini_b4_l8": dict(kind="mini", N=1000, Q=8, B=4, L=8, V=256, seed=101),
    "g1_mini_b10_l32": dict(kind="mini", N=1000, Q=6, B=10, L=32, V=256, seed=102),
    "g1_mini_b2_l4_v1024": dict(kind="mini", N=1000, Q=4, B=2, L=4, V=1024, seed=103),
    "g1_mini_b10_l8_tiny_trie": dict(kind="mini", N=7, Q=3, B=10, L=8, V=256, seed=104),  # < B leaves
    "g1_mini_b4_l8_logsoftmax": dict(kind="mini", N=1000, Q=4, B=4, L=8, V=256, seed=105, log_softmax=True),
    "g1_mini_b4_l8_shared": dict(kind="mini", N=500, Q=4, B=4, L=8, V=256, seed=106, shared=True),
    # decoder vocab sizes that are not multiples of 64 (the library pads the token axis internally)
    "g1_mini_b4_l8_v100": dict(kind="mini", N=1000, Q=4, B=4, L=8, V=100, seed=107),
    "g1_mini_b4_l6_v200_logsoftmax": dict(kind="mini", N=800, Q=3, B=4, L=6, V=200, seed=108, log_softmax=True),
    "g2_base_b10_l32": dict(kind="base", N=1000, Q=4, B=10, L=32, V=256, seed=201),
    # t5-large decoder shape (24 layers, 16 heads, d=1024 are forced by the reference ctor), B=100 top-k stress
    "g3_large_b100_l16": dict(kind="large", N=3000, Q=4, B=100, L=16, V=256, seed=301),
    # BASELINE config 4 exactly: t5-large decoder, beam 100, len 32; four queries so that the per-rank parity
    # test (tests/test_gpu_parity.py) keeps its force even if one query sits on a pruning near-tie
    "g3_large_b100_l32": dict(kind="large", N=3000, Q=4, B=100, L=32, V=256, seed=302),
    # BASELINE config 1 at its stated shape: t5-base dims, 1k-doc trie, beam 1 (greedy), 64 queries. HF 4.17's
    # BeamSearchScorer refuses num_beams <= 1, so the restated scorer is allowed a single beam
    # here; everything else (model forward, processor, beam loop) is the imported reference.
    "g4_base_b1_l32_q64": dict(kind="base", N=1000, Q=64, B=1, L=32, V=256, seed=401, single_beam=True),
    # BASELINE config 4 at the REAL t5-large dimensions (d_model 1024, d_ff 4096, 24 + 24 layers, 16 heads; the g3
    # fixtures shrink d_ff and the encoder): beam 100, short smtids so that the reference's full-prefix recompute stays
    # within minutes on the build container's 8 cores. Pins the K = 4096 FF path inside a 24-layer decoder end to end.
    "g5_largefull_b100_l8": dict(kind="large_full", N=3000, Q=2, B=100, L=8, V=256, seed=501),
    # t5-3b decoder shape (24 layers, 32 heads of d_kv = 128, d = 1024: t5_generative_retriever.py:128-133): the 128-dim head
    # path (generic attention kernels, d_kv-strided KV cache); one encoder layer and a small d_ff keep the fixture small
    "g7_3b_b10_l8": dict(kind="3b", N=3000, Q=4, B=10, L=8, V=256, seed=701),
    # RIPOR's 16 x 1024 codebook variant (reference full_16_1024_scripts/full_evaluate_t5seq_aq_encoder.sh:19-22: M=16,
    # nbits=10) at the real t5-base dimensions, at the headline beam and at the training-data beam
    "g6_base_v1024_b10_l16": dict(kind="base", N=3000, Q=4, B=10, L=16, V=1024, seed=601),
    "g6_base_v1024_b100_l16": dict(kind="base", N=3000, Q=2, B=100, L=16, V=1024, seed=602),
}

# SURVEY §8 row f4 (BASELINE config 5): T5SeqAQEncoderForLngKnpMarginMSE.forward on a seeded batch
# (reference modeling/t5_generative_retriever.py:902-966). smtid length 8 / 16 / 32 -> 2 / 3 / 4 losses.
TRAIN_CASES = {
    "f4_mini_bz6_l32": dict(kind="mini", bz=6, L=32, V=256, seed=501),
    "f4_mini_bz4_l16": dict(kind="mini", bz=4, L=16, V=256, seed=502),
    "f4_mini_bz4_l8": dict(kind="mini", bz=4, L=8, V=256, seed=503),
    # t5-base dims WITH the reference's gradients (sampled entries + norms): pins the backward at full dims to the reference
    # directly, not through the oracle's autograd
    "f4_base_bz4_l32": dict(kind="base", bz=4, L=32, V=256, seed=504, backward=True),
}


def make_case(name, spec, gen, mod, utils, shim):
    kind, N, Q, B, L, V, seed = spec["kind"], spec["N"], spec["Q"], spec["B"], spec["L"], spec["V"], spec["seed"]
    shared = spec.get("shared", False)
    if kind == "mini":
        dims = synth.mini_dims(L=L, V=V, shared_output_input_embeds=shared)
    elif kind == "base":
        dims = synth.t5_base_dims(L=L, V=V, vocab_size=2048, shared_output_input_embeds=shared)
    elif kind == "large":
        dims = synth.ModelDims(vocab_size=512, d_model=1024, d_kv=64, d_ff=512, num_layers=1, num_decoder_layers=24,
                               num_heads=16, decoder_vocab_sizes=[V] * L, shared_output_input_embeds=shared)
    elif kind == "large_full":
        dims = synth.t5_large_dims(L=L, V=V, vocab_size=2048, shared_output_input_embeds=shared)
    elif kind == "3b":
        dims = synth.ModelDims(vocab_size=512, d_model=1024, d_kv=128, d_ff=512, num_layers=1, num_decoder_layers=24,
                               num_heads=32, decoder_vocab_sizes=[V] * L, shared_output_input_embeds=shared)
    else:
        raise ValueError(kind)
    t0 = time.time()
    sd = synth.make_state_dict(dims, seed=seed)
    model = build_reference_model(mod, dims, sd, which="t5-3b" if kind == "3b" else "t5-large" if kind.startswith("large") else "t5-base")
    codes = synth.make_codes(N, L, V, seed=seed)
    d2s, lst = reference_trie(gen, codes)
    processor = gen.PrefixConstrainLogitProcessorFastSparse(lst, V)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=20)
    global ALLOW_SINGLE_BEAM
    ALLOW_SINGLE_BEAM = bool(spec.get("single_beam", False))
    enc, seqs, scores, strs, step_scores = run_reference(
        gen, utils, shim, model, processor, ids, mask, B, L, spec.get("log_softmax", False))
    ALLOW_SINGLE_BEAM = False
    top_scores, top_idx = replay_top(step_scores, Q, B, V)
    # processor-only vectors (G4): valid and invalid prefixes at a few depths
    pm = {}
    for T in sorted({1, 2, min(3, L), L}):
        pref = np.zeros((6, T), dtype=np.int64)
        for r in range(6):
            if T > 1:
                pref[r, 1:] = codes[(r * 131) % N, : T - 1]
        if T > 1:
            pref[5, T - 1] = (int(pref[5, T - 1]) + 1) % V  # most likely an unknown prefix
        m = processor(torch.from_numpy(pref), torch.zeros((6, V)))
        pm[f"pm_prefix_T{T}"] = pref
        pm[f"pm_mask_T{T}"] = np.packbits(m.numpy().astype(np.uint8), axis=1)
    # margins: for each step/query the sorted top-(B+1) cumulative candidates are recomputable only
    # with beam scores; store the reference's per-step processed scores compactly (float64 -> top 2B
    # per row is not enough), so store full scores only for small cases.
    out = dict(
        spec=json.dumps(dict(spec, name=name, dims=dims.__dict__)),
        input_ids=ids, attention_mask=mask, codes=codes,
        encoder_out=enc.astype(np.float32), sequences=seqs.astype(np.int64),
        sequences_scores=scores.astype(np.float32), smtid_strings=np.array(strs),
        top_scores=top_scores, top_idx=top_idx,
        **pm,
    )
    if step_scores.size <= 500_000:   # larger cases keep the compact per-step top-(B+1) record only
        out["step_scores"] = step_scores
    else:
        out["step_scores_first"] = step_scores[:2]
        out["step_scores_last"] = step_scores[-1:]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: {time.time() - t0:.1f}s -> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


TEACHER_KEYS = {8: ["", "smtid_4_"], 16: ["", "smtid_4_", "smtid_8_"], 32: ["", "smtid_4_", "smtid_8_", "smtid_16_"]}


def train_batch(name, dims, bz, L, V, seed):
    """Seeded batch in the layout LngKnpMarginMSEforT5SeqAQCollator emits (reference dataset/data_collator.py:11-88;
    dataset/dataset.py:488-500: doc encoding = smtid[1:], decoder_input_ids = smtid[:-1] with smtid[0] = -1; the
    positive and the negative example of a row carry the same query text)."""
    ids, mask = synth.make_queries(bz, vocab_size=dims.vocab_size, seed=seed, max_len=20)
    codes = synth.make_codes(2 * bz, L, V, seed=seed).astype(np.int64)
    pos, neg = codes[:bz], codes[bz:]
    neg[:, :2] = pos[:, :2]          # hard negatives share a prefix with the positive, like beam-search rank data
    teacher = {}
    for k in TEACHER_KEYS[L]:
        for side in ("pos", "neg"):
            teacher[f"{k}teacher_{side}_scores"] = synth.uniform_f32(f"train/{name}/{k}{side}", (bz,), 30.0, seed)
    return ids, mask, pos, neg, teacher


def grad_samples(name, shape, n=48):
    """Seeded flat indices at which a gradient tensor is sampled for the fixture (the whole tensor would be too big to
    commit for the 768x3072 matrices); tests recompute the same indices from the tensor's name and shape."""
    numel = int(np.prod(shape))
    if numel <= 4096:
        return np.arange(numel, dtype=np.int64)
    return np.sort(synth.randint(f"gradidx/{name}", (n,), 0, numel, seed=7)).astype(np.int64)


def train_step_reference(m, inputs, names_in_order):
    """One training step the way the reference runs it (tasks/trainer.py:203-275 + HF Trainer defaults used by
    main.py:131-155): total loss = sum of the task losses (ln_to_weight = 1 each, arguments.py:109-119), backward,
    clip_grad_norm_(1.0), AdamW(lr, betas (0.9, 0.999), eps 1e-8, weight_decay 0). fp32 here (the reference trains
    under bf16 autocast; the fixture pins the arithmetic the step implements, not bf16 rounding)."""
    params = {k: p for k, p in m.base_model.named_parameters() if p.requires_grad}
    for p in params.values():
        p.grad = None
    losses = m(**inputs)
    total = sum(losses[k] for k in names_in_order)
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in params.items() if p.grad is not None}
    gnorm = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    return losses, float(total), grads, gnorm, params


def make_train_case(name, spec, gen, mod, utils, shim):
    kind, bz, L, V, seed = spec["kind"], spec["bz"], spec["L"], spec["V"], spec["seed"]
    dims = synth.mini_dims(L=L, V=V) if kind == "mini" else synth.t5_base_dims(L=L, V=V, vocab_size=2048)
    t0 = time.time()
    sd = synth.make_state_dict(dims, seed=seed)
    base = build_reference_model(mod, dims, sd)
    base.config.decoding = False      # T5SeqAQEncoder.__init__ (t5_generative_retriever.py:774): no logits in training
    Cls = mod.T5SeqAQEncoderForLngKnpMarginMSE
    m = Cls.__new__(Cls)              # the ctor only loads a checkpoint dir; the forward below is the reference's own
    torch.nn.Module.__init__(m)
    m.base_model, m.config, m.loss_fn = base, base.config, torch.nn.MSELoss()
    m.eval()
    ids, mask, pos, neg, teacher = train_batch(name, dims, bz, L, V, seed)
    start = np.full((bz, 1), -1, dtype=np.int64)

    def tq(codes):
        return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask),
                "decoder_input_ids": torch.from_numpy(np.concatenate([start, codes[:, :-1]], axis=1))}

    inputs = {"pos_tokenized_query": tq(pos), "neg_tokenized_query": tq(neg),
              "pos_doc_encoding": torch.from_numpy(pos), "neg_doc_encoding": torch.from_numpy(neg)}
    inputs.update({k: torch.from_numpy(v) for k, v in teacher.items()})
    with torch.no_grad():
        losses = m(**inputs)
        # per-position student scores from the reference's own hidden states and codebooks (diagnostic granularity)
        ph = base(**tq(pos)).decoder_last_hidden_state
        nh = base(**tq(neg)).decoder_last_hidden_state
        pos_pp = (ph * m.decode(torch.from_numpy(pos))).sum(-1).numpy()
        neg_pp = (nh * m.decode(torch.from_numpy(neg))).sum(-1).numpy()
    # ---- backward + one optimizer step of the reference (second half)
    extra = {}
    if spec.get("backward", kind == "mini"):
        loss_names = sorted(losses.keys())
        _, total, grads, gnorm, params = train_step_reference(m, inputs, loss_names)
        keep = {k: g for k, g in grads.items() if "decoder.embed_tokens" not in k}   # transformers-5.x-only unused table
        # the encoder's token table is the shared one (tied in 4.17): its gradient is reported under shared.weight
        gn = sorted(keep)
        extra["grad_names"] = np.array(gn)
        extra["grad_norms"] = np.array([float(keep[k].double().norm()) for k in gn])
        extra["grad_samples"] = np.concatenate([keep[k].reshape(-1)[torch.from_numpy(grad_samples(k, keep[k].shape))].numpy()
                                                for k in gn]).astype(np.float32)
        extra["grad_sample_counts"] = np.array([len(grad_samples(k, keep[k].shape)) for k in gn])
        extra["total_loss"] = np.float64(total)
        extra["grad_global_norm"] = np.float64(gnorm)
        # one AdamW step at lr 1e-4 after clipping to norm 1.0, then the loss on the same batch again
        lr = spec.get("lr", 2e-6)   # small enough that the random-weight model stays in the linear regime of one step
        opt = torch.optim.AdamW([p for p in params.values()], lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)
        torch.nn.utils.clip_grad_norm_([p for p in params.values()], 1.0)
        before = {k: params[k].detach().clone() for k in gn}
        opt.step()
        extra["step_lr"] = np.float64(lr)
        extra["param_delta_samples"] = np.concatenate(
            [(params[k].detach() - before[k]).reshape(-1)[torch.from_numpy(grad_samples(k, before[k].shape))].numpy()
             for k in gn]).astype(np.float32)
        with torch.no_grad():
            after = m(**inputs)
        extra["losses_after_step"] = np.array([float(after[k]) for k in loss_names], dtype=np.float64)
    out = dict(spec=json.dumps(dict(spec, name=name, dims=dims.__dict__)), input_ids=ids, attention_mask=mask,
               pos_doc_encoding=pos, neg_doc_encoding=neg, pos_position_scores=pos_pp.astype(np.float32),
               neg_position_scores=neg_pp.astype(np.float32),
               loss_names=np.array(sorted(losses.keys())),
               losses=np.array([float(losses[k]) for k in sorted(losses.keys())], dtype=np.float64))
    out.update(teacher)
    out.update(extra)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: {time.time() - t0:.1f}s losses {dict((k, float(v)) for k, v in losses.items())} -> {path}")


# ----------------------------------------------------------------------------- caller fixtures
# The reference's OWN caller functions — constrained_decode_doc (evaluate.py:87-132), constrained_decode (:45-85),
# constrained_decode_smtid (:134-178) and the two merge steps t5seq_aq_retrieve_docids_2 (:489-526) /
# t5seq_aq_get_qid_to_smtid_rankdata_2 (:600-655) — imported from /root/reference/t5_pretrainer/evaluate.py and run on CPU.
# evaluate.py needs two modules that are not installed (`import faiss`, utils/metrics.py: `from pytrec_eval import
# RelevanceEvaluator`): both are satisfied by EMPTY stub modules — neither is touched by the functions run here. The
# module's `generate_for_constrained_prefix_beam_search` is pointed at the same shimmed wrapper every search fixture uses
# (run_reference: encoder once, repeat_interleave(B), the reference's beam_search_for_constrained_prefix), `evaluate`
# (metrics, pytrec_eval) is stubbed out and torch.cuda.device_count() is made to return the number of rank files the
# merge asserts on (:505, :611). Query shards: torch's DistributedSampler(shuffle=False), as evaluate.py:468 builds it.
CALLER_CASES = {
    # 2 ranks, batch size 3 (a ragged last batch), duplicated smtids (several docids per smtid), one smtid removed from
    # the lookup after the trie was built (the "smtid not in smtid_to_docid" branch), a prefix search at 4 of 8 positions
    "c5_callers_mini": dict(kind="mini", N=600, Q=7, B=4, L=8, Lp=4, V=256, seed=601, world=2, batch_size=3, dup=150,
                            clusters=24),
}


def load_reference_evaluate(gen):
    import importlib
    import importlib.machinery
    try:
        import transformers.trainer  # noqa: F401  (pulls `datasets`, which probes for faiss: must happen before the stub exists)
    except Exception:
        pass
    for name in ("faiss", "pytrec_eval"):
        if name not in sys.modules:
            m_ = types.ModuleType(name)
            m_.__spec__ = importlib.machinery.ModuleSpec(name, None)
            m_.RelevanceEvaluator = object
            sys.modules[name] = m_
    ev = importlib.import_module("t5_pretrainer.evaluate")
    assert os.path.realpath(ev.__file__).startswith(REF + os.sep), ev.__file__
    return ev


def make_caller_case(name, spec, gen, mod, utils, shim):
    import tempfile
    from types import SimpleNamespace
    from torch.utils.data.distributed import DistributedSampler
    ev = load_reference_evaluate(gen)
    N, Q, B, L, Lp, V, seed, W, bs = (spec[k] for k in ("N", "Q", "B", "L", "Lp", "V", "seed", "world", "batch_size"))
    t0 = time.time()
    dims = synth.mini_dims(L=L, V=V)
    sd = synth.make_state_dict(dims, seed=seed)
    model = build_reference_model(mod, dims, sd)
    codes = synth.make_codes(N, L, V, seed=seed)
    codes[:, :Lp] = codes[np.arange(N) % spec["clusters"], :Lp]   # few distinct prefixes: a prefix smtid holds many docids
    codes[N - spec["dup"]:] = codes[: spec["dup"]]           # the last docs repeat the smtids of the first ones
    d2s, lst = reference_trie(gen, codes)
    processor = gen.PrefixConstrainLogitProcessorFastSparse(lst, V)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=20)
    qids = np.arange(Q, dtype=np.int64) * 7 + 1000

    class Out:
        pass

    def wrapper(model_, processor_, input_ids=None, attention_mask=None, max_new_tokens=None, num_beams=None,
                num_return_sequences=None, apply_log_softmax_for_scores=False, **kw):
        assert num_return_sequences == num_beams and kw.get("return_dict_in_generate")
        _enc, seqs, scores, _strs, _steps = run_reference(gen, utils, shim, model_, processor_, input_ids, attention_mask,
                                                          num_beams, max_new_tokens, apply_log_softmax_for_scores)
        o = Out()
        o.sequences, o.sequences_scores = torch.from_numpy(seqs), torch.from_numpy(scores)
        return o

    ev.generate_for_constrained_prefix_beam_search = wrapper
    ev.evaluate = lambda args: None
    ev.tqdm = lambda it, **kw: it
    real_count = torch.cuda.device_count
    torch.cuda.device_count = lambda: W

    def smtid_lookup(n_tok):     # evaluate.py:439-446 (inside t5seq_aq_retrieve_docids: cannot be called on its own)
        out = {}
        for docid, smtids in d2s.items():
            assert smtids[0] == -1
            out.setdefault("_".join(str(x) for x in smtids[1:1 + n_tok]), []).append(docid)
        return out

    def loaders():
        for r in range(W):
            idx = list(DistributedSampler(list(range(Q)), num_replicas=W, rank=r, shuffle=False))
            yield r, idx, [{"id": torch.from_numpy(qids[idx[i:i + bs]]), "input_ids": torch.from_numpy(ids[idx[i:i + bs]]),
                            "attention_mask": torch.from_numpy(mask[idx[i:i + bs]])} for i in range(0, len(idx), bs)]

    res = {}
    try:
        with tempfile.TemporaryDirectory() as tmp:
            full = smtid_lookup(L)
            # first pass to learn what the model returns, then drop the best smtid of the first query from the lookup
            probe = wrapper(model, processor, input_ids=torch.from_numpy(ids[:1]), attention_mask=torch.from_numpy(mask[:1]),
                            max_new_tokens=L, num_beams=B, num_return_sequences=B, return_dict_in_generate=True)
            dropped = utils.convert_ptsmtids_to_strsmtid(probe.sequences.view(-1, B, L + 1), L)[0][0]
            lookup = {k: v for k, v in full.items() if k != dropped}
            for variant, ls in (("doc", False), ("doc_logsoftmax", True)):
                out_dir = os.path.join(tmp, variant, "msmarco_dev")   # get_dataset_name(...) -> "MSMARCO"
                os.makedirs(os.path.join(tmp, variant, "MSMARCO"))
                shards = {}
                for r, idx, batches in loaders():
                    ev.constrained_decode_doc(model, batches, processor, lookup, L, "cpu", os.path.join(tmp, variant, "MSMARCO"), r,
                                              topk=B, apply_log_softmax_for_scores=ls)
                    shards[r] = json.load(open(os.path.join(tmp, variant, "MSMARCO", f"run_{r}.json")))
                    res.setdefault("shard_indices", {})[r] = idx
                ev.t5seq_aq_retrieve_docids_2(SimpleNamespace(q_collection_paths=[json.dumps([out_dir])], out_dir=os.path.join(tmp, variant)))
                res[variant] = dict(shards=shards, merged=json.load(open(os.path.join(tmp, variant, "MSMARCO", "run.json"))))
            # smtid-level outputs of the full-length search (evaluate.py:45-85)
            sm_dir = os.path.join(tmp, "smtid")
            os.makedirs(sm_dir)
            shards = {}
            for r, idx, batches in loaders():
                ev.constrained_decode(model, batches, processor, lookup, L, "cpu", sm_dir, r, topk=B)
                shards[r] = json.load(open(os.path.join(sm_dir, f"qid_to_smtid_{r}.json")))
            res["smtid"] = dict(shards=shards)
            # training-data generation pass: prefix search over the first Lp positions, nested output, merge (:134-178, :600-655)
            pdir = os.path.join(tmp, "prefix")
            os.makedirs(pdir)
            plookup = smtid_lookup(Lp)
            shards = {}
            for r, idx, batches in loaders():
                ev.constrained_decode_smtid(model, batches, processor, plookup, Lp, "cpu", pdir, r, topk=B)
                shards[r] = json.load(open(os.path.join(pdir, f"qid_smtid_rankdata_{r}.json")))
            ev.t5seq_aq_get_qid_to_smtid_rankdata_2(SimpleNamespace(out_dir=pdir))
            res["prefix"] = dict(shards=shards, merged=json.load(open(os.path.join(pdir, "qid_smtid_rankdata.json"))))
    finally:
        torch.cuda.device_count = real_count
    # G6: the index lists of torch's DistributedSampler(shuffle=False) at MSMARCO-dev size
    samp = {str(w): [list(DistributedSampler(list(range(6980)), num_replicas=w, rank=r, shuffle=False))[:4] +
                     list(DistributedSampler(list(range(6980)), num_replicas=w, rank=r, shuffle=False))[-4:]
                     + [len(list(DistributedSampler(list(range(6980)), num_replicas=w, rank=r, shuffle=False)))]
                     for r in range(w)] for w in (1, 2, 3, 4, 8)}
    out = dict(spec=json.dumps(dict(spec, name=name, dims=dims.__dict__)), input_ids=ids, attention_mask=mask, codes=codes,
               qids=qids, dropped_smtid=np.array(dropped), results=np.array(json.dumps(res)),
               sampler_head_tail_len=np.array(json.dumps(samp)))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    n_doc = sum(len(v) for v in res["doc"]["merged"].values())
    print(f"[golden] {name}: {time.time() - t0:.1f}s, merged run holds {len(res['doc']['merged'])} queries / {n_doc} docids, "
          f"dropped smtid {dropped} -> {path} ({os.path.getsize(path) / 1e3:.1f} KB)")


# ----------------------------------------------------------------------------- data side of the fine-tune (row f4)
class WordTokenizer:
    """Whitespace tokenizer with the HF call signature the reference collator uses (no pretrained tokenizer offline): word ->
    3 + crc32 % 97, ``</s>`` = 1 appended, padded with 0 to the longest of the batch, truncated to max_length."""

    def __call__(self, texts, add_special_tokens=True, padding="longest", truncation="longest_first", max_length=64,
                 return_attention_mask=True, return_tensors="pt"):
        import zlib
        assert padding == "longest" and truncation == "longest_first" and add_special_tokens and return_tensors == "pt"
        ids = [[3 + zlib.crc32(w.encode()) % 97 for w in t.split()][: max_length - 1] + [1] for t in texts]
        m = max(len(x) for x in ids)
        return {"input_ids": torch.tensor([x + [0] * (m - len(x)) for x in ids]),
                "attention_mask": torch.tensor([[1] * len(x) + [0] * (m - len(x)) for x in ids])}


def lngknp_files(root, L, n_ex=5, n_cand=4, seed=0):
    """The three inputs of LngKnpMarginMSEforT5SeqAQDataset: examples (jsonl), query collection, docid_to_smtid.json."""
    rng = np.random.RandomState(1234 + L + seed)
    os.makedirs(os.path.join(root, "queries"), exist_ok=True)
    os.makedirs(os.path.join(root, "docs"), exist_ok=True)
    with open(os.path.join(root, "queries", "raw.tsv"), "w") as f:
        for q in range(n_ex):
            f.write(f"{100 + q}\t" + " ".join(f"w{int(x)}" for x in rng.randint(0, 50, size=3 + q)) + " \n")
    with open(os.path.join(root, "docs", "raw.tsv"), "w") as f:
        f.write("0\tunused document text\n")
    d2s, lines = {}, []
    for q in range(n_ex):
        docids = [str(1000 + q * 10 + j) for j in range(n_cand)]
        smtids = []
        for d in docids:
            codes = [int(x) for x in rng.randint(0, 256, size=L)]
            d2s[d] = [-1] + codes
            smtids.append("_".join(str(c) for c in codes))
        ex = {"qid": str(100 + q), "docids": docids, "smtids": smtids, "scores": [float(x) for x in rng.uniform(-5, 30, n_cand).round(3)]}
        for k in (4, 8, 16):
            if k < L:
                ex[f"smtid_{k}_scores"] = [float(x) for x in rng.uniform(-5, 30, n_cand).round(3)]
        lines.append(json.dumps(ex))
    with open(os.path.join(root, "examples.jsonl"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(root, "docid_to_smtid.json"), "w") as f:
        json.dump(d2s, f)
    return dict(queries=open(os.path.join(root, "queries", "raw.tsv")).read(), docs=open(os.path.join(root, "docs", "raw.tsv")).read(),
                examples=open(os.path.join(root, "examples.jsonl")).read(), docid_to_smtid=json.dumps(d2s))


def make_lngknp_data_case(name="c6_lngknp_data"):
    """The reference's OWN dataset and collator (dataset/dataset.py:418-525, dataset/data_collator.py:11-88) on small seeded
    files: items for a fixed index order under random.seed(5) and the collated batch, for smtid lengths 8 / 16 / 32 in both
    lookup modes. AutoTokenizer.from_pretrained is replaced by the whitespace tokenizer above (no tokenizer files offline)."""
    import importlib
    import random
    import tempfile
    ds_mod = importlib.import_module("t5_pretrainer.dataset.dataset")
    dc_mod = importlib.import_module("t5_pretrainer.dataset.data_collator")
    for m_ in (ds_mod, dc_mod):
        assert os.path.realpath(m_.__file__).startswith(REF + os.sep), m_.__file__

    class _AT:
        @staticmethod
        def from_pretrained(_path):
            return WordTokenizer()

    dc_mod.AutoTokenizer = _AT
    cases = {}
    for L in (8, 16, 32):
        for as_docid in (True, False):
            with tempfile.TemporaryDirectory() as root:
                files = lngknp_files(root, L)
                ds = ds_mod.LngKnpMarginMSEforT5SeqAQDataset(
                    dataset_path=os.path.join(root, "examples.jsonl"), document_dir=os.path.join(root, "docs"),
                    query_dir=os.path.join(root, "queries"), docid_to_smtid_path=None if as_docid else os.path.join(root, "docid_to_smtid.json"),
                    smtid_as_docid=as_docid)
                order = [3, 0, 4, 1, 2, 0]
                random.seed(5)
                items = [ds[i] for i in order]
                coll = dc_mod.LngKnpMarginMSEforT5SeqAQCollator("unused", max_length=6)
                batch = coll(items[:4])
                flat = {}
                for k, v in batch.items():
                    if isinstance(v, dict):
                        for kk, vv in v.items():
                            flat[f"{k}.{kk}"] = vv.tolist()
                    else:
                        flat[k] = v.tolist()
                cases[f"L{L}_{'smtid' if as_docid else 'docid'}"] = dict(
                    files=files, order=order, items=[list(it) for it in items], batch=flat, length=len(ds), max_length=6)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, cases=np.array(json.dumps(cases)))
    print(f"[golden] {name}: {len(cases)} cases -> {path} ({os.path.getsize(path) / 1e3:.1f} KB)")


# ----------------------------------------------------------------------------- query front-end + truncate_run (row f3)
def frontend_files(root):
    """A checkpoint directory with an offline-trained SentencePiece tokenizer and a ``raw.tsv`` query collection that
    exercises the reader: tabs inside the text, CRLF line ends, unicode line separators inside a line, trailing blanks,
    a query longer than max_length (256) tokens, numeric ids that are not row numbers, and an empty last line."""
    import random
    import sentencepiece as spm
    ckpt = os.path.join(root, "checkpoint")
    qdir = os.path.join(root, "msmarco_frontend", "dev_queries")
    os.makedirs(ckpt)
    os.makedirs(qdir)
    rnd = random.Random(11)
    words = ["what", "is", "the", "how", "to", "of", "in", "a", "best", "price", "weather", "define"] + [f"w{i}" for i in range(150)]
    corpus = os.path.join(root, "corpus.txt")
    with open(corpus, "w") as f:
        for _ in range(1500):
            f.write(" ".join(rnd.choice(words) for _ in range(rnd.randint(3, 12))) + "\n")
    spm.SentencePieceTrainer.train(input=corpus, model_prefix=os.path.join(ckpt, "spiece"), vocab_size=200, model_type="unigram",
                                   pad_id=0, eos_id=1, unk_id=2, bos_id=-1, pad_piece="<pad>", eos_piece="</s>", unk_piece="<unk>",
                                   hard_vocab_limit=False, minloglevel=2)
    tok_cfg = {"tokenizer_class": "T5Tokenizer", "extra_ids": 0, "model_max_length": 512}
    json.dump(tok_cfg, open(os.path.join(ckpt, "tokenizer_config.json"), "w"))
    sent = lambda n: " ".join(rnd.choice(words) for _ in range(n))
    lines = [f"1048585\t{sent(5)}\n",
             f"2\t{sent(3)}\twith a tab\tand another\n",            # tabs inside the text -> joined by blanks
             f"1102432\t  {sent(4)}  \r\n",                          # CRLF, leading / trailing blanks (kept by id_style row_id)
             f"524332\t{sent(2)}\x0b{sent(2)}\u2028{sent(1)}\n",    # vertical tab / LINE SEPARATOR inside a line -> blanks
             f"87181\t{sent(300)}\n",                                # > 256 tokens: truncated to max_length
             f" 300674 \tUPPER case Query ? unknown-chars \u00e9\u4e2d\n",   # id with blanks; pieces outside the vocabulary
             f"7\t{sent(1)}\n",
             f"915593\t{sent(9)}\n",
             f"264014\t{sent(6)}\n",
             f"1\t{sent(7)}\n",
             f"33\t\n",                                              # empty text: "query: " alone
             "\n"]                                                   # a bare newline (len(line) <= 1) is skipped; only legal at the end
    raw = "".join(lines)
    with open(os.path.join(qdir, "raw.tsv"), "w", newline="") as f:   # newline="": the CRLF reaches the file as written
        f.write(raw)
    return ckpt, qdir, raw, open(os.path.join(ckpt, "spiece.model"), "rb").read(), tok_cfg


def make_frontend_case(name="c8_frontend"):
    """The reference's OWN query front-end (evaluate.py:461-468: CollectionDatasetWithDocIDPreLoad(id_style="row_id",
    add_prefix=True, is_query=True) + CollectionDataWithDocIDLoader(max_length=256, sampler=DistributedSampler(shuffle=False)),
    dataset/dataset.py:266-332, dataset/dataloader.py:62-79) on a small query file, and utils/metrics.py:9-15 ``truncate_run``
    (imported with a stub pytrec_eval: the function is pure Python) on runs with score ties at and around the cut."""
    import importlib
    import importlib.machinery
    import tempfile
    from torch.utils.data.distributed import DistributedSampler
    ds_mod = importlib.import_module("t5_pretrainer.dataset.dataset")
    dl_mod = importlib.import_module("t5_pretrainer.dataset.dataloader")
    if "pytrec_eval" not in sys.modules:
        m_ = types.ModuleType("pytrec_eval")
        m_.__spec__ = importlib.machinery.ModuleSpec("pytrec_eval", None)
        m_.RelevanceEvaluator = object
        sys.modules["pytrec_eval"] = m_
    mt_mod = importlib.import_module("t5_pretrainer.utils.metrics")
    for m_ in (ds_mod, dl_mod, mt_mod):
        assert os.path.realpath(m_.__file__).startswith(REF + os.sep), m_.__file__
    out = {}
    with tempfile.TemporaryDirectory() as root:
        ckpt, qdir, raw, spiece, tok_cfg = frontend_files(root)
        ds = ds_mod.CollectionDatasetWithDocIDPreLoad(data_dir=qdir, id_style="row_id", tid_to_smtid_path=None, add_prefix=True,
                                                      is_query=True)
        out["items"] = [list(ds[i]) for i in range(len(ds))]
        loaders = {}
        for W, bs in ((1, 4), (2, 3), (3, 2)):
            for r in range(W):
                # num_workers=0 and no pinned memory: the collate function is what is pinned here, not the worker plumbing
                real_init = torch.utils.data.DataLoader.__init__
                def _init(self, *a, **k):
                    k["pin_memory"] = False
                    real_init(self, *a, **k)
                torch.utils.data.DataLoader.__init__ = _init
                try:
                    ld = dl_mod.CollectionDataWithDocIDLoader(dataset=ds, tokenizer_type=ckpt, max_length=256, batch_size=bs, num_workers=0,
                                                              sampler=DistributedSampler(ds, num_replicas=W, rank=r, shuffle=False))
                    batches = [{k: v.tolist() for k, v in b.items()} for b in ld]
                finally:
                    torch.utils.data.DataLoader.__init__ = real_init
                loaders[f"w{W}_r{r}_bs{bs}"] = batches
        out["loaders"] = loaders
    # truncate_run: ties inside the top k, a tie straddling the cut (run-file order decides), fewer than k documents,
    # integer and float scores, negative scores, k = 1
    runs = {
        "q_tie_at_cut": {"d1": 3.0, "d2": 5.0, "d3": 3.0, "d4": 3.0, "d5": 1.0},
        "q_all_equal": {"b": 1.5, "a": 1.5, "c": 1.5},
        "q_short": {"x": -2.0, "y": -1.0},
        "q_mixed": {"m1": 2, "m2": 2.0, "m3": 10, "m4": -7.25, "m5": 2},
        "q_empty": {},
    }
    out["truncate"] = {str(k): mt_mod.truncate_run(runs, k) for k in (1, 2, 3, 10)}
    # key ORDER is part of the result (the evaluator and the json writer see it): stored as lists of pairs
    out["truncate_order"] = {k: {q: list(d.items()) for q, d in v.items()} for k, v in out["truncate"].items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, raw_tsv=np.array(raw), spiece_model=np.frombuffer(spiece, dtype=np.uint8),
                        tokenizer_config=np.array(json.dumps(tok_cfg)), runs=np.array(json.dumps(runs)),
                        results=np.array(json.dumps(out)))
    nb = sum(len(v) for v in out["loaders"].values())
    print(f"[golden] {name}: {len(out['items'])} queries, {nb} batches over {len(out['loaders'])} loaders -> {path} "
          f"({os.path.getsize(path) / 1e3:.1f} KB)")


def make_beam_indices_case(gen, mod, utils, shim, name="c7_beam_indices"):
    """``beam_indices`` of the reference's output object (generation.py:521-522, :548-552) for two committed search fixtures
    (the ``scores`` tuple is already stored there as ``step_scores``): per returned slot the L parent indices q*B + slot."""
    out = {}
    for case in ("g1_mini_b4_l8", "g1_mini_b4_l8_logsoftmax", "g1_mini_b10_l32"):
        spec = CASES[case]
        N, Q, B, L, V, seed = spec["N"], spec["Q"], spec["B"], spec["L"], spec["V"], spec["seed"]
        dims = synth.mini_dims(L=L, V=V, shared_output_input_embeds=spec.get("shared", False))
        model = build_reference_model(mod, dims, synth.make_state_dict(dims, seed=seed))
        codes = synth.make_codes(N, L, V, seed=seed)
        _, lst = reference_trie(gen, codes)
        processor = gen.PrefixConstrainLogitProcessorFastSparse(lst, V)
        ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=20)
        run_reference(gen, utils, shim, model, processor, ids, mask, B, L, spec.get("log_softmax", False))
        out[case] = run_reference.last_beam_indices
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"[golden] {name}: {[(k, v.shape) for k, v in out.items()]} -> {path}")


def make_checkpoint_case(mod, name="c9_ref_checkpoint"):
    """A checkpoint DIRECTORY written by the reference's own T5SeqAQEncoder.save_pretrained (t5_generative_retriever.py:850-851:
    self.base_model.save_pretrained -> HF PreTrainedModel.save_pretrained of the installed transformers), zipped. The model is the
    smallest the reference's constructor accepts (12 decoder layers x 12 heads -> d_model 768; everything else minimal) with
    synth.patterned_state_dict values, so the zip is small and the loader under test (ripor_amd T5SeqAQEncoder.from_pretrained)
    is checked file by file, key by key."""
    import tempfile
    import zipfile
    dims = synth.ModelDims(vocab_size=16, d_model=768, d_kv=2, d_ff=4, num_layers=1, num_decoder_layers=12, num_heads=12,
                           decoder_vocab_sizes=[64, 64])
    sd = synth.patterned_state_dict(dims)
    # transformers 5.x refuses to save tensors that share storage unless the class declares the tie (4.17 wrote both names);
    # the reference ties the encoder's embedding to `shared` (:90) — declared here in 5.x's format, part of the version shim
    mod.T5ForDocIDGeneration._tied_weights_keys = {"encoder.embed_tokens.weight": "shared.weight"}

    # ... and 5.x constructs models on the meta device inside from_pretrained, where the reference constructor's
    # `self.start_token_embed.data = <numpy-backed tensor>` (:121) cannot run: 4.17's from_pretrained (build on the CPU, then
    # load the weight file) restated for the wrapper's call at :782
    def from_pretrained_417(cls, path, config=None, **kw):
        from safetensors.torch import load_file
        model = cls(config)
        st = os.path.join(path, "model.safetensors")
        weights = load_file(st) if os.path.exists(st) else torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
        weights.setdefault("encoder.embed_tokens.weight", weights["shared.weight"])
        missing, unexpected = model.load_state_dict(weights, strict=False)
        assert not unexpected and all("decoder.embed_tokens" in m for m in missing), (missing, unexpected)
        return model.eval()

    mod.T5ForDocIDGeneration.from_pretrained = classmethod(from_pretrained_417)
    base = build_reference_model(mod, dims, sd)
    with tempfile.TemporaryDirectory() as tmp:
        first, second = os.path.join(tmp, "a"), os.path.join(tmp, "b")
        base.save_pretrained(first)                                   # a directory the reference's wrapper can be built from
        enc = mod.T5SeqAQEncoder.from_pretrained(first, shared_output_input_embeds=False)   # :853-855 -> __init__ :772-784
        got = enc.base_model.state_dict()
        for k, v in sd.items():                                       # the reference read back what was written
            assert torch.equal(got[k].cpu(), torch.from_numpy(v)), k
        enc.save_pretrained(second)                                   # :850-851
        files = sorted(os.listdir(second))
        out = os.path.join(HERE, name + ".zip")
        with zipfile.ZipFile(out, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as z:
            for f in files:
                z.write(os.path.join(second, f), f)
    print(f"[golden] {name}: {files} -> {os.path.getsize(out)} bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    torch.manual_seed(0)
    gen, mod, utils, shim = load_reference()
    for name, spec in CASES.items():
        if args.only and args.only not in (name, "search"):
            continue
        make_case(name, spec, gen, mod, utils, shim)
    for name, spec in TRAIN_CASES.items():
        if args.only and args.only not in (name, "train"):
            continue
        make_train_case(name, spec, gen, mod, utils, shim)
    for name, spec in CALLER_CASES.items():
        if args.only and args.only not in (name, "callers"):
            continue
        make_caller_case(name, spec, gen, mod, utils, shim)
    if not args.only or args.only in ("c6_lngknp_data", "callers"):
        make_lngknp_data_case()
    if not args.only or args.only in ("c7_beam_indices", "callers"):
        make_beam_indices_case(gen, mod, utils, shim)
    if not args.only or args.only in ("c8_frontend", "callers"):
        make_frontend_case()
    if not args.only or args.only in ("c9_ref_checkpoint", "callers"):
        make_checkpoint_case(mod)


if __name__ == "__main__":
    main()
