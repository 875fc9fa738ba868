"""Host-side mirror of the reference's model objects for the generative-retrieval path.

Same names and call surface as reference t5_pretrainer/modeling/t5_generative_retriever.py
(``T5forDocIDConfig`` :45-67, ``T5ForDocIDGeneration`` :70-512, ``T5SeqAQEncoder`` :772-855) as far as
``evaluate.py``'s retrieve tasks use them: ``from_pretrained`` of an HF checkpoint directory,
``.base_model``, ``.config.decoder_vocab_sizes``, ``.eval()``, ``.to(device)``, ``.device``,
``save_pretrained``. The objects only hold weights; all arithmetic of the search happens in
libripor_hip.so (ripor_amd/engine.py binds the weights to an ``rpr_model``). There is no PyTorch
forward here and no CPU fallback: calling the search without a HIP device raises.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional

import numpy as np
import torch


class T5forDocIDConfig:
    """Fields of the reference config (T5Config + docid extras, reference :45-65)."""

    model_type = "t5"

    def __init__(self, decoder_vocab_sizes: Optional[List[int]] = None, decoding: bool = False,
                 decoder_start_token_path: str = "./t5_decoder_start_token_embeds/t5-base.npy",
                 apply_decoder_t5_stack: bool = False, scaleup_output_hidden: bool = False,
                 shared_output_input_embeds: bool = True, vocab_size: int = 32128, d_model: int = 768,
                 d_kv: int = 64, d_ff: int = 3072, num_layers: int = 12, num_decoder_layers: Optional[int] = None,
                 num_heads: int = 12, relative_attention_num_buckets: int = 32,
                 relative_attention_max_distance: int = 128, layer_norm_epsilon: float = 1e-6,
                 feed_forward_proj: str = "relu", **kwargs):
        self.decoder_vocab_sizes = list(decoder_vocab_sizes) if decoder_vocab_sizes is not None else [256] * 32
        self.decoding = decoding
        self.decoder_start_token_path = decoder_start_token_path
        self.apply_decoder_t5_stack = apply_decoder_t5_stack
        self.scaleup_output_hidden = scaleup_output_hidden
        self.shared_output_input_embeds = shared_output_input_embeds
        self.vocab_size, self.d_model, self.d_kv, self.d_ff = vocab_size, d_model, d_kv, d_ff
        self.num_layers = num_layers
        self.num_decoder_layers = num_decoder_layers if num_decoder_layers is not None else num_layers
        self.num_heads = num_heads
        self.relative_attention_num_buckets = relative_attention_num_buckets
        self.relative_attention_max_distance = relative_attention_max_distance
        self.layer_norm_epsilon = layer_norm_epsilon
        self.feed_forward_proj = feed_forward_proj
        self.tie_word_embeddings = False
        self.max_decoder_length = len(self.decoder_vocab_sizes)
        self.extra = dict(kwargs)
        assert self.apply_decoder_t5_stack == False  # noqa: E712  (reference :67)
        if feed_forward_proj != "relu":
            raise ValueError(f"feed_forward_proj={feed_forward_proj!r}: only the non-gated ReLU T5 v1.0 block is supported")

    @classmethod
    def from_pretrained(cls, path: str) -> "T5forDocIDConfig":
        with open(os.path.join(path, "config.json")) as f:
            d = json.load(f)
        d.pop("model_type", None)
        d.pop("tie_word_embeddings", None)
        d.pop("max_decoder_length", None)
        return cls(**d)

    @classmethod
    def from_dims(cls, dims) -> "T5forDocIDConfig":
        """From ripor_amd.utils.synth.ModelDims (synthetic checkpoints)."""
        return cls(decoder_vocab_sizes=dims.decoder_vocab_sizes, vocab_size=dims.vocab_size, d_model=dims.d_model,
                   d_kv=dims.d_kv, d_ff=dims.d_ff, num_layers=dims.num_layers,
                   num_decoder_layers=dims.num_decoder_layers, num_heads=dims.num_heads,
                   relative_attention_num_buckets=dims.relative_attention_num_buckets,
                   relative_attention_max_distance=dims.relative_attention_max_distance,
                   layer_norm_epsilon=dims.layer_norm_epsilon,
                   shared_output_input_embeds=dims.shared_output_input_embeds,
                   scaleup_output_hidden=dims.scaleup_output_hidden)

    def to_dict(self) -> dict:
        d = {k: v for k, v in self.__dict__.items() if k != "extra"}
        d.update(self.extra)
        d["model_type"] = self.model_type
        return d

    def save_pretrained(self, path: str):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(self.to_dict(), f, indent=2)


_IGNORED_KEYS = ("encoder.embed_tokens.weight", "decoder.embed_tokens.weight", "lm_head.weight",
                 "decoder.block.0.layer.1.EncDecAttention.relative_attention_bias.weight")


def expected_keys(cfg: T5forDocIDConfig) -> List[str]:
    """State-dict keys the search path reads."""
    keys = ["shared.weight", "encoder.final_layer_norm.weight", "decoder.final_layer_norm.weight", "start_token_embed",
            "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight",
            "decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer"
        keys += [f"{p}.0.SelfAttention.{w}.weight" for w in "qkvo"]
        keys += [f"{p}.0.layer_norm.weight", f"{p}.1.layer_norm.weight",
                 f"{p}.1.DenseReluDense.wi.weight", f"{p}.1.DenseReluDense.wo.weight"]
    for i in range(cfg.num_decoder_layers):
        p = f"decoder.block.{i}.layer"
        keys += [f"{p}.0.SelfAttention.{w}.weight" for w in "qkvo"]
        keys += [f"{p}.1.EncDecAttention.{w}.weight" for w in "qkvo"]
        keys += [f"{p}.0.layer_norm.weight", f"{p}.1.layer_norm.weight", f"{p}.2.layer_norm.weight",
                 f"{p}.2.DenseReluDense.wi.weight", f"{p}.2.DenseReluDense.wo.weight"]
    for i in range(len(cfg.decoder_vocab_sizes)):
        keys.append(f"list_decoder_embeds.{i}.weight")
        if not cfg.shared_output_input_embeds:
            keys.append(f"list_output_embeds.{i}.weight")
    return keys


def _load_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(path, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {path}")


class T5ForDocIDGeneration:
    """Weights of the docid-generation T5 (reference :70-135) bound lazily to the HIP engine."""

    def __init__(self, config: T5forDocIDConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        self.config = config
        self._sd: Dict[str, torch.Tensor] = {}
        self._device = torch.device("cpu")
        self._engine_model = None
        if (config.num_decoder_layers, config.num_heads) not in ((12, 12), (24, 16), (24, 32)):
            # t5-base, t5-large, t5-3b (32 heads of d_kv = 128: search and encode only): reference :116-135
            raise ValueError("the model with decoer layers {} is not supported.".format(config.num_decoder_layers))
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # -- weights --
    def load_state_dict(self, state_dict, strict: bool = True):
        sd = {}
        for k, v in state_dict.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            sd[k] = v.detach().to(torch.float32)
        need = expected_keys(self.config)
        missing = [k for k in need if k not in sd]
        unexpected = [k for k in sd if k not in need and k not in _IGNORED_KEYS]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                               f"unexpected {unexpected[:8]}")
        self._sd = {k: sd[k] for k in need if k in sd}
        self._engine_model = None
        return missing, unexpected

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(self._sd)

    @classmethod
    def from_pretrained(cls, model_name_or_path: str, config: Optional[T5forDocIDConfig] = None):
        config = config if config is not None else T5forDocIDConfig.from_pretrained(model_name_or_path)
        return cls(config, _load_checkpoint(model_name_or_path))

    def save_pretrained(self, save_dir: str):
        self.config.save_pretrained(save_dir)
        torch.save({k: v.cpu() for k, v in self._sd.items()}, os.path.join(save_dir, "pytorch_model.bin"))

    # -- module-like surface used by evaluate.py --
    def eval(self):
        return self

    def to(self, device):
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        if device != self._device:
            self._device = device
            self._engine_model = None
        return self

    def cuda(self, device=None):
        return self.to(torch.device("cuda", device if device is not None else torch.cuda.current_device()))

    @property
    def device(self) -> torch.device:
        return self._device

    def engine_model(self):
        """The rpr_model bound to this object's weights on its device (built on first use)."""
        if self._engine_model is None:
            from .. import engine as E
            if self._device.type != "cuda":
                raise E.RiporHipError("T5ForDocIDGeneration is on the CPU: move it to a HIP device with .to(local_rank) "
                                      "(the search path has no CPU fallback)")
            ctx = E.Context.get(self._device)
            self._engine_model = E.DeviceModel(ctx, self._sd, self.config)
        return self._engine_model

    def __call__(self, *a, **k):
        raise NotImplementedError("only the constrained beam search path is built: use "
                                  "ripor_amd.tasks.generation.generate_for_constrained_prefix_beam_search")


class T5SeqAQEncoder:
    """reference :772-855 (inference surface only)."""

    def __init__(self, model_name_or_path, shared_output_input_embeds=None, multi_vocab_sizes=None):
        config = T5forDocIDConfig.from_pretrained(model_name_or_path)
        config.decoding = False
        if shared_output_input_embeds is not None:
            assert shared_output_input_embeds in [False, True]
            config.shared_output_input_embeds = shared_output_input_embeds
        self.base_model = T5ForDocIDGeneration.from_pretrained(model_name_or_path, config=config)
        self.config = config
        self.model_args = None

    @classmethod
    def from_pretrained(cls, model_name_or_path=None, shared_output_input_embeds=None, multi_vocab_sizes=False):
        return cls(model_name_or_path, shared_output_input_embeds, multi_vocab_sizes)

    @classmethod
    def from_synthetic(cls, dims, seed=None):
        """Random-init checkpoint of the given dims (no network / no pretrained weights here)."""
        from ..utils import synth
        obj = cls.__new__(cls)
        obj.config = T5forDocIDConfig.from_dims(dims)
        sd = synth.make_state_dict(dims) if seed is None else synth.make_state_dict(dims, seed=seed)
        obj.base_model = T5ForDocIDGeneration(obj.config, sd)
        obj.model_args = None
        return obj

    def eval(self):
        return self

    def to(self, device):
        self.base_model.to(device)
        return self

    def save_pretrained(self, save_dir):
        self.base_model.save_pretrained(save_dir)


class T5SeqAQEncoderForLngKnpMarginMSE(T5SeqAQEncoder):
    """reference :902-966 — forward of the prefix-oriented ranking fine-tune step, same
    ``forward(**inputs)`` dict interface and return keys (``rank``, ``rank_4``, ``rank_8``, ``rank_16``). The whole
    pass runs in ``rpr_lngknp_forward`` (teacher-forced decoder over all positions of the positive and the negative
    smtid at once, one encoder pass per query). Forward only: no autograd graph is attached to the returned losses."""

    _PREFIXES = {8: [(8, ""), (4, "smtid_4_")], 16: [(16, ""), (4, "smtid_4_"), (8, "smtid_8_")],
                 32: [(32, ""), (4, "smtid_4_"), (8, "smtid_8_"), (16, "smtid_16_")]}

    def forward(self, **inputs):
        from .. import engine as E
        pos_q, neg_q = inputs["pos_tokenized_query"], inputs["neg_tokenized_query"]
        pos_codes, neg_codes = inputs["pos_doc_encoding"], inputs["neg_doc_encoding"]
        L = pos_codes.size(1)
        if L not in self._PREFIXES:
            raise ValueError("not valid length: {}".format(L))
        if not (torch.equal(pos_q["input_ids"], neg_q["input_ids"])):
            # dataset.py:502-503 builds both from the same query text; two different texts would need two encoder passes
            raise ValueError("pos_tokenized_query and neg_tokenized_query must carry the same query tokens")
        for side, q, codes in (("pos", pos_q, pos_codes), ("neg", neg_q, neg_codes)):
            di = q["decoder_input_ids"]
            if not torch.equal(di[:, 1:].to(codes.device), codes[:, :-1]):
                raise ValueError(f"{side}: decoder_input_ids must be the doc encoding shifted right (dataset.py:497-500)")
        names = ["rank" if k == L else f"rank_{k}" for k, _ in self._PREFIXES[L]]
        tp = torch.stack([inputs[p + "teacher_pos_scores"] for _, p in self._PREFIXES[L]]).float()
        tn = torch.stack([inputs[p + "teacher_neg_scores"] for _, p in self._PREFIXES[L]]).float()
        codes = torch.stack([pos_codes, neg_codes], dim=1)
        em = self.base_model.engine_model()
        ctx = em.ctx
        ctx.clear_status_async()
        args = (em, pos_q["input_ids"], pos_q["attention_mask"], codes, tp, tn, [k for k, _ in self._PREFIXES[L]])
        losses, self.last_position_scores = E.lngknp_forward(*args)
        # same guard as the search path (tasks/generation.py): this forward runs on the static-scale f16 planes; an
        # activation outside their range is clamped and flagged, and the pass is then repeated on the exact-fp32 GEMMs
        st = ctx.status(clear=True)
        if st & E._lib.STATUS_EMPTY_QUERY:
            raise ValueError("a query has an all-zero attention_mask (no token to attend to)")
        if (st & E._lib.STATUS_SATURATED) and ctx.get_precision() != "f32":
            import warnings
            warnings.warn("activation outside the f16 plane range of the split-precision GEMMs: repeating this forward "
                          "with exact fp32 MFMA (RPR_PRECISION=f32 avoids the retry)")
            saved = ctx.get_precision()   # "f16x2" or "bf16" (a training ctx): whatever it was comes back
            ctx.set_precision("f32")
            try:
                losses, self.last_position_scores = E.lngknp_forward(*args)
                ctx.status(clear=True)
            finally:
                ctx.set_precision(saved)
        return {n: losses[i] for i, n in enumerate(names)}

    __call__ = forward

    # ---- training (backward pass + optimizer on the device; reference: HF Trainer around this forward) -----------------
    def _batch(self, inputs):
        pos_q, neg_q = inputs["pos_tokenized_query"], inputs["neg_tokenized_query"]
        pos_codes, neg_codes = inputs["pos_doc_encoding"], inputs["neg_doc_encoding"]
        L = pos_codes.size(1)
        if L not in self._PREFIXES:
            raise ValueError("not valid length: {}".format(L))
        if not torch.equal(pos_q["input_ids"], neg_q["input_ids"]):
            raise ValueError("pos_tokenized_query and neg_tokenized_query must carry the same query tokens")
        names = ["rank" if k == L else f"rank_{k}" for k, _ in self._PREFIXES[L]]
        tp = torch.stack([inputs[p + "teacher_pos_scores"] for _, p in self._PREFIXES[L]]).float()
        tn = torch.stack([inputs[p + "teacher_neg_scores"] for _, p in self._PREFIXES[L]]).float()
        return pos_q, torch.stack([pos_codes, neg_codes], dim=1), tp, tn, [k for k, _ in self._PREFIXES[L]], names

    def train_state(self):
        from .. import engine as E
        if getattr(self, "_train_state", None) is None or self._train_state.model is not self.base_model.engine_model():
            self._train_state = E.TrainState(self.base_model.engine_model())
        return self._train_state

    def backward(self, **inputs):
        """loss = sum of the task losses (ln_to_weight 1, reference arguments.py:109-119; tasks/trainer.py:228-240),
        loss.backward(): fills ``train_state().grads``; returns the task losses like ``forward``."""
        from .. import engine as E
        pos_q, codes, tp, tn, prefix_lens, names = self._batch(inputs)
        losses = E.lngknp_backward(self.base_model.engine_model(), self.train_state(), pos_q["input_ids"],
                                   pos_q["attention_mask"], codes, tp, tn, prefix_lens)
        return {n: losses[i] for i, n in enumerate(names)}

    def training_step(self, lr, max_grad_norm=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **inputs):
        """One optimisation step as the reference's trainer performs it: backward with the gradient all-reduce across the
        data-parallel ranks overlapped (RCCL when torch.distributed is initialised; engine.GradExchange), clip_grad_norm_,
        AdamW — weights updated in place on the device. Returns the task losses of the batch (before the update)."""
        from .. import engine as E
        pos_q, codes, tp, tn, prefix_lens, names = self._batch(inputs)
        losses = E.train_step(self.base_model.engine_model(), self.train_state(), pos_q["input_ids"], pos_q["attention_mask"],
                              codes, tp, tn, prefix_lens, lr, betas=betas, eps=eps, weight_decay=weight_decay,
                              max_grad_norm=max_grad_norm)
        losses = {n: losses[i] for i, n in enumerate(names)}
        return losses
