"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product path (ripor_amd/*).

CPU (torch, float32) restatement of the T5 encoder/decoder arithmetic that the reference's
``T5ForDocIDGeneration`` executes on the constrained-beam-search path.

Follows, line by line in behaviour (not in code):
  * reference t5_pretrainer/modeling/t5_generative_retriever.py:194-214  (decoder input embeds)
  * reference t5_pretrainer/modeling/t5_generative_retriever.py:250-262  (per-position logits)
  * reference t5_pretrainer/modeling/t5_generative_retriever.py:295-450  (forward: encoder once,
    full-prefix decoder, all-position logits)
  * third-party ``transformers`` T5Stack/T5Block/T5Attention/T5LayerNorm math (pinned 4.17.0 at
    reference requirements.txt:1; not vendored) as summarised in SURVEY.md Appendix B: RMSNorm
    without mean/bias, unscaled QK^T + bucketed relative bias (block-0 table shared by all
    blocks), additive pad/causal masks, ReLU feed-forward, final RMSNorm.

Parity pinning: validated against the *imported* reference (through the compatibility shim in
tests/golden/make_golden.py) — see tests/golden/*.npz and tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

NEG_MASK = torch.finfo(torch.float32).min  # masked keys contribute exp(.)==0 exactly in fp32


def relative_position_bucket(rel: torch.Tensor, bidirectional: bool, num_buckets: int = 32,
                             max_distance: int = 128) -> torch.Tensor:
    """Bucket function of HF ``T5Attention._relative_position_bucket`` (float32 log, truncation).
    ``rel`` = key_pos - query_pos (int64)."""
    buckets = torch.zeros_like(rel)
    if bidirectional:
        num_buckets //= 2
        buckets = buckets + (rel > 0).to(torch.long) * num_buckets
        rel = rel.abs()
    else:
        rel = -torch.minimum(rel, torch.zeros_like(rel))
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (
        torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def bucket_table(bidirectional: bool, max_len: int, num_buckets: int = 32, max_distance: int = 128) -> np.ndarray:
    """Host-side lookup used by tests to pin the device tables: for the decoder
    ``tab[n] = bucket(rel=-n)``, n in [0,max_len); for the encoder ``tab[n + max_len-1] =
    bucket(rel=n)`` for n in (-max_len, max_len)."""
    if bidirectional:
        rel = torch.arange(-(max_len - 1), max_len, dtype=torch.long)
    else:
        rel = -torch.arange(0, max_len, dtype=torch.long)
    return relative_position_bucket(rel, bidirectional, num_buckets, max_distance).numpy().astype(np.int32)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


class T5Ref:
    """Weights are taken from a state dict with the reference checkpoint's key names."""

    def __init__(self, state_dict: Dict[str, np.ndarray], dims):
        self.dims = dims
        self.sd = {k: torch.from_numpy(np.ascontiguousarray(v)).float() for k, v in state_dict.items()}
        self.H, self.dkv, self.eps = dims.num_heads, dims.d_kv, dims.layer_norm_epsilon
        self.L = len(dims.decoder_vocab_sizes)

    # -- attention core (scores are NOT scaled by 1/sqrt(dkv)) ------------------------------
    def _attn(self, prefix: str, x: torch.Tensor, kv: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
        B, Tq, _ = x.shape
        Tk = kv.shape[1]
        sd, H, dkv = self.sd, self.H, self.dkv
        q = (x @ sd[prefix + ".q.weight"].t()).view(B, Tq, H, dkv).transpose(1, 2)
        k = (kv @ sd[prefix + ".k.weight"].t()).view(B, Tk, H, dkv).transpose(1, 2)
        v = (kv @ sd[prefix + ".v.weight"].t()).view(B, Tk, H, dkv).transpose(1, 2)
        scores = q @ k.transpose(2, 3) + bias
        w = torch.softmax(scores.float(), dim=-1)
        o = (w @ v).transpose(1, 2).reshape(B, Tq, H * dkv)
        return o @ sd[prefix + ".o.weight"].t()

    def _ff(self, prefix: str, x: torch.Tensor) -> torch.Tensor:
        sd = self.sd
        return torch.relu(x @ sd[prefix + ".wi.weight"].t()) @ sd[prefix + ".wo.weight"].t()

    # -- encoder (reference: model.encoder(...) once per batch, generation.py:132-137) -------
    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        sd, d = self.sd, self.dims
        x = sd["shared.weight"][input_ids]
        Q, Lq = input_ids.shape
        pos = torch.arange(Lq)
        bucket = relative_position_bucket(pos[None, :] - pos[:, None], True,
                                          d.relative_attention_num_buckets, d.relative_attention_max_distance)
        tab = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        bias = tab[bucket].permute(2, 0, 1)[None]  # [1,H,Lq,Lq]
        pad = torch.where(attention_mask[:, None, None, :] > 0, 0.0, NEG_MASK)
        bias = bias + pad
        for i in range(d.num_layers):
            p = f"encoder.block.{i}.layer"
            h = rmsnorm(x, sd[p + ".0.layer_norm.weight"], self.eps)
            x = x + self._attn(p + ".0.SelfAttention", h, h, bias)
            h = rmsnorm(x, sd[p + ".1.layer_norm.weight"], self.eps)
            x = x + self._ff(p + ".1.DenseReluDense", h)
        return rmsnorm(x, sd["encoder.final_layer_norm.weight"], self.eps)

    # -- decoder input embeddings (reference :194-214) ---------------------------------------
    def decoder_inputs_embeds(self, ids: torch.Tensor) -> torch.Tensor:
        sd = self.sd
        R, T = ids.shape
        parts = [sd["start_token_embed"].expand(R, 1, -1)]
        for i in range(1, T):
            parts.append(sd[f"list_decoder_embeds.{i - 1}.weight"][ids[:, i]].unsqueeze(1))
        return torch.cat(parts, dim=1)

    # -- full-prefix decoder, as the reference runs it every step (no KV cache is consumed) ---
    def decode_full(self, ids: torch.Tensor, enc: torch.Tensor, enc_mask: torch.Tensor) -> torch.Tensor:
        sd, d = self.sd, self.dims
        x = self.decoder_inputs_embeds(ids)
        R, T, _ = x.shape
        pos = torch.arange(T)
        bucket = relative_position_bucket(pos[None, :] - pos[:, None], False,
                                          d.relative_attention_num_buckets, d.relative_attention_max_distance)
        tab = sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        causal = torch.where(pos[None, :] <= pos[:, None], 0.0, NEG_MASK)
        self_bias = tab[bucket].permute(2, 0, 1)[None] + causal[None, None]
        cross_bias = torch.where(enc_mask[:, None, None, :] > 0, 0.0, NEG_MASK)
        for i in range(d.num_decoder_layers):
            p = f"decoder.block.{i}.layer"
            h = rmsnorm(x, sd[p + ".0.layer_norm.weight"], self.eps)
            x = x + self._attn(p + ".0.SelfAttention", h, h, self_bias)
            h = rmsnorm(x, sd[p + ".1.layer_norm.weight"], self.eps)
            x = x + self._attn(p + ".1.EncDecAttention", h, enc, cross_bias)
            h = rmsnorm(x, sd[p + ".2.layer_norm.weight"], self.eps)
            x = x + self._ff(p + ".2.DenseReluDense", h)
        x = rmsnorm(x, sd["decoder.final_layer_norm.weight"], self.eps)
        if d.scaleup_output_hidden:
            x = x * (d.d_model ** -0.5)
        return x

    def out_embed(self, i: int) -> torch.Tensor:
        if self.dims.shared_output_input_embeds:
            return self.sd[f"list_decoder_embeds.{i}.weight"]
        return self.sd[f"list_output_embeds.{i}.weight"]

    def last_logits(self, ids: torch.Tensor, enc: torch.Tensor, enc_mask: torch.Tensor) -> torch.Tensor:
        """``outputs.logits[-1]`` of the reference forward (reference :250-262, generation.py:448)."""
        h = self.decode_full(ids, enc, enc_mask)
        i = ids.shape[1] - 1
        return h[:, i, :] @ self.out_embed(i).t()


class T5RefCached(T5Ref):
    """KV-cached variant of the same arithmetic (mathematically identical because decoder
    self-attention is causal — SURVEY.md §3.2 Quirk A). Used only to make larger oracle cases
    finish in seconds; validated against :class:`T5Ref` in tests/test_oracle_golden.py."""

    def start(self, enc: torch.Tensor, enc_mask: torch.Tensor):
        sd, d, H, dkv = self.sd, self.dims, self.H, self.dkv
        R, Lq, _ = enc.shape
        self.cross = []
        for i in range(d.num_decoder_layers):
            p = f"decoder.block.{i}.layer.1.EncDecAttention"
            k = (enc @ sd[p + ".k.weight"].t()).view(R, Lq, H, dkv).transpose(1, 2)
            v = (enc @ sd[p + ".v.weight"].t()).view(R, Lq, H, dkv).transpose(1, 2)
            self.cross.append((k, v))
        self.cross_bias = torch.where(enc_mask[:, None, None, :] > 0, 0.0, NEG_MASK)
        self.k_cache = [None] * d.num_decoder_layers
        self.v_cache = [None] * d.num_decoder_layers
        self.t = 0

    def reorder(self, beam_idx: torch.Tensor):
        for i in range(len(self.k_cache)):
            self.k_cache[i] = self.k_cache[i].index_select(0, beam_idx)
            self.v_cache[i] = self.v_cache[i].index_select(0, beam_idx)

    def step(self, last_tokens: Optional[torch.Tensor], R: int) -> torch.Tensor:
        """One decoder position; returns logits ``[R, V]`` for position ``t``."""
        sd, d, H, dkv, t = self.sd, self.dims, self.H, self.dkv, self.t
        if t == 0:
            x = sd["start_token_embed"].expand(R, 1, -1)
        else:
            x = sd[f"list_decoder_embeds.{t - 1}.weight"][last_tokens].unsqueeze(1)
        rel = (torch.arange(t + 1) - t)[None, :]
        bucket = relative_position_bucket(rel, False, d.relative_attention_num_buckets,
                                          d.relative_attention_max_distance)
        tab = sd["decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
        self_bias = tab[bucket].permute(2, 0, 1)[None]  # [1,H,1,t+1]
        for i in range(d.num_decoder_layers):
            p = f"decoder.block.{i}.layer"
            h = rmsnorm(x, sd[p + ".0.layer_norm.weight"], self.eps)
            a = p + ".0.SelfAttention"
            q = (h @ sd[a + ".q.weight"].t()).view(R, 1, H, dkv).transpose(1, 2)
            k = (h @ sd[a + ".k.weight"].t()).view(R, 1, H, dkv).transpose(1, 2)
            v = (h @ sd[a + ".v.weight"].t()).view(R, 1, H, dkv).transpose(1, 2)
            self.k_cache[i] = k if t == 0 else torch.cat([self.k_cache[i], k], dim=2)
            self.v_cache[i] = v if t == 0 else torch.cat([self.v_cache[i], v], dim=2)
            w = torch.softmax((q @ self.k_cache[i].transpose(2, 3) + self_bias).float(), dim=-1)
            o = (w @ self.v_cache[i]).transpose(1, 2).reshape(R, 1, H * dkv)
            x = x + o @ sd[a + ".o.weight"].t()
            h = rmsnorm(x, sd[p + ".1.layer_norm.weight"], self.eps)
            a = p + ".1.EncDecAttention"
            q = (h @ sd[a + ".q.weight"].t()).view(R, 1, H, dkv).transpose(1, 2)
            ck, cv = self.cross[i]
            w = torch.softmax((q @ ck.transpose(2, 3) + self.cross_bias).float(), dim=-1)
            o = (w @ cv).transpose(1, 2).reshape(R, 1, H * dkv)
            x = x + o @ sd[a + ".o.weight"].t()
            h = rmsnorm(x, sd[p + ".2.layer_norm.weight"], self.eps)
            x = x + self._ff(p + ".2.DenseReluDense", h)
        x = rmsnorm(x, sd["decoder.final_layer_norm.weight"], self.eps)
        if d.scaleup_output_hidden:
            x = x * (d.d_model ** -0.5)
        self.t += 1
        return x[:, 0, :] @ self.out_embed(t).t()
