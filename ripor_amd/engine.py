"""Thin Python objects over the C ABI: device context, bound model weights, device trie, search.

torch is used only for device memory (tensors own the buffers whose ``data_ptr()`` crosses the
ABI) and for the current HIP stream handle. All compute happens in ``libripor_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import RiporHipError, check

_contexts: Dict[int, "Context"] = {}


def _stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context:
    """One ``rpr_ctx`` per device (workspaces, KV cache and hipGraphs live in it)."""

    def __init__(self, device_index: int):
        if not torch.cuda.is_available():
            raise RiporHipError("no HIP device visible to torch: the search path has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device_index)
        h = C.c_void_p()
        check(self.lib.rpr_init(device_index, C.byref(h)), "rpr_init")
        self.handle = h

    @staticmethod
    def get(device=None) -> "Context":
        if device is None:
            idx = torch.cuda.current_device() if torch.cuda.is_available() else 0
        else:
            dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
            if dev.type != "cuda":
                raise RiporHipError(f"device {dev} is not a HIP device: the search path has no CPU fallback")
            idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if idx not in _contexts:
            _contexts[idx] = Context(idx)
        return _contexts[idx]

    def set_precision(self, name: str):
        """'f32' = exact fp32 MFMA; 'f16x2' = split-precision f16 MFMA (default)."""
        check(self.lib.rpr_set_precision(self.handle, {"f32": _lib.PREC_F32, "f16x2": _lib.PREC_F16X2, "bf16": _lib.PREC_BF16}[name]),
              "rpr_set_precision")

    def get_precision(self) -> str:
        return {_lib.PREC_F32: "f32", _lib.PREC_F16X2: "f16x2", _lib.PREC_BF16: "bf16"}[int(self.lib.rpr_get_precision(self.handle))]

    def has_bf16(self) -> bool:
        """The bf16 arithmetic of the training GEMMs (RPR_PREC_BF16) is compiled into this library."""
        return True

    def workspace_bytes(self) -> int:
        return int(self.lib.rpr_workspace_bytes(self.handle))

    def set_lane_split(self, min_rows: int):
        """Batches of at least ``min_rows`` decoder rows (queries x beams) run as two halves on two CU-masked streams
        (0 = never)."""
        check(self.lib.rpr_set_lane_split(self.handle, int(min_rows)), "rpr_set_lane_split")

    def lane_split(self) -> int:
        """The threshold in force; 0 when splitting is off or masked streams are unavailable."""
        return int(self.lib.rpr_lane_split(self.handle))

    def set_forced_tail(self, mode):
        """Forced-tail evaluation (``rpr_set_forced_tail``): queries whose beams can no longer be pruned leave the
        step-by-step loop at a fork and get their remaining positions scored in one teacher-forced pass.
        ``False``/0 = off, ``True``/1 = exact (library default), 2 = optimistic: no stage after the last fork; a query
        left unforced there raises ``STATUS_TAIL_LEFTOVER`` and the call must be repeated in mode 1 (see
        :func:`search_checked`)."""
        check(self.lib.rpr_set_forced_tail(self.handle, int(mode)), "rpr_set_forced_tail")

    def forced_tail(self) -> int:
        return int(self.lib.rpr_forced_tail(self.handle))

    def set_fork_depths(self, depths: Optional[Sequence[int]]):
        """Explicit fork depths (ascending, at most two; ``[]`` = never fork); ``None`` = choose from the trie statistics."""
        if depths is None:
            check(self.lib.rpr_set_fork_depths(self.handle, -1, None), "rpr_set_fork_depths")
        else:
            arr = (C.c_int32 * max(1, len(depths)))(*[int(d) for d in depths])
            check(self.lib.rpr_set_fork_depths(self.handle, len(depths), arr), "rpr_set_fork_depths")

    def fork_depths(self, model: "DeviceModel", trie: "DeviceTrie", Q: int, B: int, L: int, log_softmax: bool = False) -> List[int]:
        """The fork depths a search of this shape would use (``rpr_fork_depths``)."""
        out = (C.c_int32 * 2)()
        n = int(self.lib.rpr_fork_depths(self.handle, model.handle, trie.handle, Q, B, L,
                                         _lib.FLAG_LOG_SOFTMAX if log_softmax else 0, out))
        if n < 0:
            check(n, "rpr_fork_depths")
        return [int(out[i]) for i in range(n)]

    def last_fork_stats(self) -> List[dict]:
        """Per fork of the last search: depth, queries forced there, queries that walked on (synchronises the device)."""
        d, f, l = (C.c_int32 * 2)(), (C.c_int32 * 2)(), (C.c_int32 * 2)()
        n = int(self.lib.rpr_last_fork_stats(self.handle, d, f, l))
        if n < 0:
            check(n, "rpr_last_fork_stats")
        return [dict(depth=int(d[i]), forced=int(f[i]), left=int(l[i])) for i in range(n)]

    def status_async(self, clear: bool = True):
        """Flags of the work enqueued so far WITHOUT synchronising (``rpr_status_words_async``): returns a
        :class:`StatusTicket` whose ``flags()`` waits only for the copy enqueued here."""
        words = torch.zeros(4, dtype=torch.int32).pin_memory()
        check(self.lib.rpr_status_words_async(self.handle, _stream_ptr(self.device), words.data_ptr(), 1 if clear else 0),
              "rpr_status_words_async")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        return StatusTicket(words, ev)

    def clear_status_async(self):
        check(self.lib.rpr_status_words_async(self.handle, _stream_ptr(self.device), None, 1), "rpr_status_words_async")

    def status(self, clear: bool = True) -> int:
        """Synchronises the current stream and returns the sticky status flags of the work enqueued so far
        (``_lib.STATUS_SATURATED``: an activation left the f16 plane range in f16x2 mode and was clamped — the results
        since the last clear must be recomputed in 'f32' precision; ``_lib.STATUS_EMPTY_QUERY``)."""
        out = C.c_uint32(0)
        check(self.lib.rpr_get_status(self.handle, _stream_ptr(self.device), C.byref(out), 1 if clear else 0), "rpr_get_status")
        flags = int(out.value) | getattr(self, "_kept_status", 0)
        if clear:
            self._kept_status = 0
        return flags

    def keep_status(self, flags: int):
        """Hand flags back that a caller read with ``status(clear=True)`` but does not own (the device words are gone):
        the next ``status()`` reports them again. Used by side passes that must clear the device words around their own
        run (tasks/generation.py ``_per_step_record``) without swallowing another search's unchecked flags."""
        self._kept_status = getattr(self, "_kept_status", 0) | int(flags)

    # -- profiling (bench.py roofline leg) --
    def profile_enable(self, on: bool):
        check(self.lib.rpr_profile_enable(self.handle, 1 if on else 0), "rpr_profile_enable")

    def profile_reset(self):
        check(self.lib.rpr_profile_reset(self.handle), "rpr_profile_reset")

    def profile_get(self) -> Dict[str, dict]:
        out = {}
        for cls, name in enumerate(_lib.KERNEL_CLASS_NAMES):
            st = _lib.KernelStats()
            check(self.lib.rpr_profile_get(self.handle, cls, C.byref(st)), "rpr_profile_get")
            out[name] = dict(total_ms=st.total_ms, launches=int(st.launches), flops=st.flops, bytes=st.bytes)
        return out

    # -- single operators (kernel-level parity tests) --
    def linear(self, A: torch.Tensor, W: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False):
        assert A.is_cuda and W.is_cuda and A.dtype == torch.float32 and W.dtype == torch.float32
        A, W = A.contiguous(), W.contiguous()
        M, K = A.shape
        N = W.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        res = residual.contiguous() if residual is not None else None
        check(self.lib.rpr_op_linear(self.handle, A.data_ptr(), W.data_ptr(), res.data_ptr() if res is not None else None,
                                     out.data_ptr(), M, N, K, 1 if relu else 0, _stream_ptr(A.device)), "rpr_op_linear")
        return out

    def linear_bf16(self, A: torch.Tensor, W: torch.Tensor, residual: Optional[torch.Tensor] = None, relu: bool = False,
                    n_products: int = 0):
        """The bf16 GEMM kernels of the fine-tune step (rpr_op_linear_bf16). n_products = 0: one product through the step's own
        kernel choice -> [M, N]; n_products >= 1: the grouped weight-gradient launch -> [n_products, M, N] (product i = the
        first M - 256 i rows of A; the rows behind them stay zero)."""
        assert A.is_cuda and W.is_cuda and A.dtype == torch.float32 and W.dtype == torch.float32
        A, W = A.contiguous(), W.contiguous()
        M, K = A.shape
        N = W.shape[0]
        shape = (M, N) if n_products == 0 else (n_products, M, N)
        out = torch.zeros(shape, dtype=torch.float32, device=A.device)
        res = residual.contiguous() if residual is not None else None
        check(self.lib.rpr_op_linear_bf16(self.handle, A.data_ptr(), W.data_ptr(), res.data_ptr() if res is not None else None,
                                          out.data_ptr(), M, N, K, 1 if relu else 0, n_products, _stream_ptr(A.device)),
              "rpr_op_linear_bf16")
        return out

    def rmsnorm(self, x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6):
        x, w = x.contiguous(), w.contiguous()
        out = torch.empty_like(x)
        check(self.lib.rpr_op_rmsnorm(self.handle, x.data_ptr(), w.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                      eps, _stream_ptr(x.device)), "rpr_op_rmsnorm")
        return out


class StatusTicket:
    """Status words of a ctx as of one point of a stream (``Context.status_async``)."""

    def __init__(self, words: torch.Tensor, event):
        self._words, self._event = words, event

    def flags(self) -> int:
        self._event.synchronize()
        w = self._words
        return ((_lib.STATUS_SATURATED if int(w[0]) else 0) | (_lib.STATUS_EMPTY_QUERY if int(w[1]) else 0) |
                (_lib.STATUS_TAIL_LEFTOVER if int(w[2]) else 0))


def rel_bucket(rel: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> int:
    """Host-only table entry exactly as the library computes it (needs no GPU)."""
    return int(_lib.load().rpr_rel_bucket(rel, 1 if bidirectional else 0, num_buckets, max_distance))


def read_docid_to_smtid(path: str):
    """``docid_to_smtid.json`` -> (docids: list[str] in file order, codes: uint16 ``[N, L]`` without the
    leading -1). Streaming C++ reader (``rpr_d2s_*``); host only, needs no GPU. Replaces the reference's
    ``ujson.load`` + per-doc list slicing (evaluate.py:400-402, 439-446)."""
    lib = _lib.load()
    h = C.c_void_p()
    check(lib.rpr_d2s_open(path.encode(), C.byref(h)), "rpr_d2s_open")
    try:
        N, L, kb = C.c_int64(), C.c_int32(), C.c_int64()
        check(lib.rpr_d2s_dims(h, C.byref(N), C.byref(L), C.byref(kb)), "rpr_d2s_dims")
        codes = np.empty((N.value, L.value), dtype=np.uint16)
        keys = C.create_string_buffer(max(1, kb.value))
        check(lib.rpr_d2s_copy(h, codes.ctypes.data_as(C.c_void_p), keys), "rpr_d2s_copy")
        docids = keys.raw[:kb.value].decode("utf-8").split("\n")
    finally:
        lib.rpr_d2s_close(h)
    assert len(docids) == codes.shape[0], (len(docids), codes.shape)
    return docids, codes


def build_trie_file(codes: np.ndarray, V: int, path: str, docids: Optional[Sequence[str]] = None,
                    source_path: Optional[str] = None) -> None:
    """HOST ONLY (no GPU, no ctx): sort the docid code matrix and write the binary trie cache (``rpr_trie_build_file``),
    optionally with the docid strings and the identity (size, mtime) of the JSON the codes came from."""
    import os
    lib = _lib.load()
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    N, L = codes.shape
    keys = "\n".join(docids).encode("utf-8") if docids is not None else b""
    if docids is not None and len(docids) != N:
        raise ValueError("one docid per code row expected")
    size = mtime = 0
    if source_path is not None:
        st = os.stat(source_path)
        size, mtime = st.st_size, st.st_mtime_ns
    check(lib.rpr_trie_build_file(codes.ctypes.data_as(C.c_void_p), N, L, int(V), keys, len(keys), size, mtime,
                                  path.encode()), "rpr_trie_build_file")


def trie_single_frac(codes: np.ndarray, L: Optional[int] = None) -> np.ndarray:
    """HOST ONLY: share of the depth-t trie nodes (t = 0..L) that hold a single distinct L-token sequence
    (``rpr_trie_single_frac``) — the statistic behind the automatic fork depths of the forced-tail search."""
    lib = _lib.load()
    codes = np.ascontiguousarray(codes, dtype=np.uint16)
    N, Lc = codes.shape
    L = Lc if L is None else int(L)
    out = (C.c_double * (L + 1))()
    check(lib.rpr_trie_single_frac(codes.ctypes.data_as(C.c_void_p), N, Lc, L, out), "rpr_trie_single_frac")
    return np.asarray(list(out), dtype=np.float64)


def trie_file_info(path: str) -> dict:
    """Header of a binary trie file (host only): N, L, V, key_bytes, src_size, src_mtime_ns."""
    lib = _lib.load()
    n, l, v, kb, ss, sm = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), C.c_int64()
    check(lib.rpr_trie_file_info(path.encode(), C.byref(n), C.byref(l), C.byref(v), C.byref(kb), C.byref(ss), C.byref(sm)),
          "rpr_trie_file_info")
    return dict(N=n.value, L=l.value, V=v.value, key_bytes=kb.value, src_size=ss.value, src_mtime_ns=sm.value)


def _ptr_array(tensors: Sequence[torch.Tensor]):
    arr = (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class DeviceModel:
    """Binds a ``T5ForDocIDGeneration`` state dict (reference checkpoint key names, SURVEY.md §8
    row a14) to an ``rpr_model``: q/k/v are concatenated, codebooks and cross-attention k/v stacked,
    everything float32 on the device."""

    def __init__(self, ctx: Context, state_dict: Dict[str, torch.Tensor], cfg):
        self.ctx, self.cfg = ctx, cfg
        dev = ctx.device

        def g(name):
            t = state_dict[name]
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(t)
            return t.to(device=dev, dtype=torch.float32).contiguous()

        L = len(cfg.decoder_vocab_sizes)
        V = int(cfg.decoder_vocab_sizes[0])
        if len(set(cfg.decoder_vocab_sizes)) != 1:
            raise ValueError("not valid decoder_vocab_size")  # reference evaluate.py:433-436
        shared_embeds = bool(cfg.shared_output_input_embeds)
        keep: List[torch.Tensor] = []
        # (checkpoint tensor name, packed device tensor, slice of it or None): how the packed tensors map back to names
        self._names: List[tuple] = []

        def K(t):
            keep.append(t)
            return t

        def N(name, t, sl=None):
            self._names.append((name, t, sl))
            return t

        def attn_qkv(prefix):
            t = K(torch.cat([g(prefix + ".q.weight"), g(prefix + ".k.weight"), g(prefix + ".v.weight")], 0).contiguous())
            n = t.shape[0] // 3
            for j, w in enumerate("qkv"):
                N(f"{prefix}.{w}.weight", t, slice(j * n, (j + 1) * n))
            return t

        ne, nd = cfg.num_layers, cfg.num_decoder_layers
        enc = dict(ln0=[], qkv=[], o=[], ln1=[], wi=[], wo=[])
        for i in range(ne):
            p = f"encoder.block.{i}.layer"
            enc["ln0"].append(N(p + ".0.layer_norm.weight", K(g(p + ".0.layer_norm.weight"))))
            enc["qkv"].append(attn_qkv(p + ".0.SelfAttention"))
            enc["o"].append(N(p + ".0.SelfAttention.o.weight", K(g(p + ".0.SelfAttention.o.weight"))))
            enc["ln1"].append(N(p + ".1.layer_norm.weight", K(g(p + ".1.layer_norm.weight"))))
            enc["wi"].append(N(p + ".1.DenseReluDense.wi.weight", K(g(p + ".1.DenseReluDense.wi.weight"))))
            enc["wo"].append(N(p + ".1.DenseReluDense.wo.weight", K(g(p + ".1.DenseReluDense.wo.weight"))))
        dec = dict(ln0=[], qkv=[], o=[], ln1=[], xq=[], xo=[], ln2=[], wi=[], wo=[])
        xkv = []
        for i in range(nd):
            p = f"decoder.block.{i}.layer"
            dec["ln0"].append(N(p + ".0.layer_norm.weight", K(g(p + ".0.layer_norm.weight"))))
            dec["qkv"].append(attn_qkv(p + ".0.SelfAttention"))
            dec["o"].append(N(p + ".0.SelfAttention.o.weight", K(g(p + ".0.SelfAttention.o.weight"))))
            dec["ln1"].append(N(p + ".1.layer_norm.weight", K(g(p + ".1.layer_norm.weight"))))
            dec["xq"].append(N(p + ".1.EncDecAttention.q.weight", K(g(p + ".1.EncDecAttention.q.weight"))))
            xkv += [g(p + ".1.EncDecAttention.k.weight"), g(p + ".1.EncDecAttention.v.weight")]
            dec["xo"].append(N(p + ".1.EncDecAttention.o.weight", K(g(p + ".1.EncDecAttention.o.weight"))))
            dec["ln2"].append(N(p + ".2.layer_norm.weight", K(g(p + ".2.layer_norm.weight"))))
            dec["wi"].append(N(p + ".2.DenseReluDense.wi.weight", K(g(p + ".2.DenseReluDense.wi.weight"))))
            dec["wo"].append(N(p + ".2.DenseReluDense.wo.weight", K(g(p + ".2.DenseReluDense.wo.weight"))))
        self.shared = N("shared.weight", K(g("shared.weight")))
        self.enc_rel = N("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", K(g("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")))
        self.dec_rel = N("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight", K(g("decoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")))
        self.enc_fln = N("encoder.final_layer_norm.weight", K(g("encoder.final_layer_norm.weight")))
        self.dec_fln = N("decoder.final_layer_norm.weight", K(g("decoder.final_layer_norm.weight")))
        self.start = N("start_token_embed", K(g("start_token_embed").reshape(-1).contiguous()))
        self.in_embeds = K(torch.stack([g(f"list_decoder_embeds.{i}.weight") for i in range(L)], 0).contiguous())
        if shared_embeds:
            self.out_embeds = self.in_embeds
        else:
            self.out_embeds = K(torch.stack([g(f"list_output_embeds.{i}.weight") for i in range(L)], 0).contiguous())
        self.dec_xkv = K(torch.cat(xkv, 0).contiguous())
        inner = cfg.num_heads * cfg.d_kv
        for i in range(nd):
            for j, w in enumerate("kv"):
                N(f"decoder.block.{i}.layer.1.EncDecAttention.{w}.weight", self.dec_xkv,
                  slice((2 * i + j) * inner, (2 * i + j + 1) * inner))
        for i in range(L):
            N(f"list_decoder_embeds.{i}.weight", self.in_embeds, i)
            if not shared_embeds:
                N(f"list_output_embeds.{i}.weight", self.out_embeds, i)
        self._keep = keep
        self._arrays = {}
        d = _lib.ModelDesc()
        d.vocab_size, d.d_model, d.d_kv, d.d_ff = cfg.vocab_size, cfg.d_model, cfg.d_kv, cfg.d_ff
        d.num_heads, d.num_layers, d.num_decoder_layers = cfg.num_heads, ne, nd
        d.rel_buckets, d.rel_max_distance = cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance
        d.L, d.V = L, V
        d.scaleup_output_hidden = 1 if cfg.scaleup_output_hidden else 0
        d.layer_norm_eps = float(cfg.layer_norm_epsilon)
        d.shared, d.enc_rel_bias, d.dec_rel_bias = self.shared.data_ptr(), self.enc_rel.data_ptr(), self.dec_rel.data_ptr()
        d.enc_final_ln, d.dec_final_ln, d.start_embed = self.enc_fln.data_ptr(), self.dec_fln.data_ptr(), self.start.data_ptr()
        d.in_embeds, d.out_embeds, d.dec_xkv = self.in_embeds.data_ptr(), self.out_embeds.data_ptr(), self.dec_xkv.data_ptr()
        for k, v in enc.items():
            self._arrays["enc_" + k] = _ptr_array(v)
            setattr(d, "enc_" + k, self._arrays["enc_" + k])
        for k, v in dec.items():
            self._arrays["dec_" + k] = _ptr_array(v)
            setattr(d, "dec_" + k, self._arrays["dec_" + k])
        h = C.c_void_p()
        torch.cuda.synchronize(dev)
        check(ctx.lib.rpr_load_model(ctx.handle, C.byref(d), C.byref(h)), "rpr_load_model")
        self.handle = h
        self.L, self.V, self.d_model = L, V, cfg.d_model
        # a weight outside the range of the f16 planes pins the model to the exact-fp32 kernels (rpr_load_model)
        self.f32_only = bool(ctx.lib.rpr_model_f32_only(h))

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.ctx.lib.rpr_free_model(self.handle)
                self.handle = None
        except Exception:
            pass

    def named_device_params(self):
        """(checkpoint tensor name, packed device tensor, index / slice of it or None) for every tensor of the model."""
        return list(self._names)

    def export_state_dict(self) -> Dict[str, torch.Tensor]:
        """The model's current device weights under the reference checkpoint's names (after training steps)."""
        out = {}
        for name, t, sl in self._names:
            v = t if sl is None else t[sl]
            out[name] = v.reshape(1, 1, -1).clone() if name == "start_token_embed" else v.clone()
        return out

    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor) -> torch.Tensor:
        dev = self.ctx.device
        ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
        mask = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
        Q, Lq = ids.shape
        out = torch.empty((Q, Lq, self.d_model), dtype=torch.float32, device=dev)
        check(self.ctx.lib.rpr_encode(self.ctx.handle, self.handle, ids.data_ptr(), mask.data_ptr(), Q, Lq,
                                      out.data_ptr(), _stream_ptr(dev)), "rpr_encode")
        return out


class DeviceTrie:
    """Sorted-code-matrix trie on the device (replaces the reference's dict/CSR structures)."""

    def __init__(self, ctx: Context, handle, L: int, V: int):
        self.ctx, self.handle, self.L, self.V = ctx, handle, L, V
        self.N = int(ctx.lib.rpr_trie_num_rows(handle))
        p = ctx.lib.rpr_trie_perm(handle)
        self.perm = np.ctypeslib.as_array(p, shape=(self.N,))  # view, valid until free

    @classmethod
    def from_codes(cls, ctx: Context, codes: np.ndarray, V: int) -> "DeviceTrie":
        codes = np.ascontiguousarray(codes, dtype=np.uint16)
        N, L = codes.shape
        h = C.c_void_p()
        check(ctx.lib.rpr_build_trie(ctx.handle, codes.ctypes.data_as(C.c_void_p), N, L, V, C.byref(h)), "rpr_build_trie")
        return cls(ctx, h, L, V)

    @classmethod
    def load(cls, ctx: Context, path: str, L: Optional[int] = None, V: Optional[int] = None) -> "DeviceTrie":
        """Load a binary trie file (``rpr_trie_load`` validates it). The depth is the FILE's; ``V`` (the model's decoder
        vocab size) may widen the file's vocab (a cache is built without knowing the model). ``L`` is accepted for
        backward compatibility and only checked against the file."""
        h = C.c_void_p()
        check(ctx.lib.rpr_trie_load(ctx.handle, path.encode(), C.byref(h)), "rpr_trie_load")
        n, fl, fv, kb = C.c_int64(), C.c_int32(), C.c_int32(), C.c_int64()
        check(ctx.lib.rpr_trie_dims(h, C.byref(n), C.byref(fl), C.byref(fv), C.byref(kb)), "rpr_trie_dims")
        try:
            if L is not None and L > fl.value:
                raise RiporHipError(f"{path} holds {fl.value} code columns, {L} requested")
            if V is not None and V != fv.value:
                check(ctx.lib.rpr_trie_set_vocab(h, V), "rpr_trie_set_vocab")
        except Exception:
            ctx.lib.rpr_free_trie(h)
            raise
        t = cls(ctx, h, fl.value, V if V is not None else fv.value)
        t.key_bytes = kb.value
        return t

    def docids(self) -> Optional[List[str]]:
        """docid strings in original row order when the trie came from a file that stores them, else None."""
        kb = getattr(self, "key_bytes", 0)
        if not kb:
            return None
        buf = C.create_string_buffer(kb)
        check(self.ctx.lib.rpr_trie_keys(self.handle, buf), "rpr_trie_keys")
        return buf.raw[:kb].decode("utf-8").split("\n")

    def save(self, path: str):
        check(self.ctx.lib.rpr_trie_save(self.handle, path.encode()), "rpr_trie_save")

    def mask(self, prefix: np.ndarray) -> np.ndarray:
        prefix = np.ascontiguousarray(prefix, dtype=np.int32)
        R, T = prefix.shape
        out = np.zeros((R, self.V), dtype=np.uint8)
        check(self.ctx.lib.rpr_trie_mask(self.ctx.handle, self.handle, prefix.ctypes.data_as(C.c_void_p), R, T,
                                         out.ctypes.data_as(C.c_void_p)), "rpr_trie_mask")
        return out

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.perm = None
                self.ctx.lib.rpr_free_trie(self.handle)
                self.handle = None
        except Exception:
            pass


@dataclass
class SearchResult:
    tokens: torch.Tensor      # int32 [Q, B, L]
    scores: torch.Tensor      # float32 [Q, B]
    row_lo: torch.Tensor      # int64 [Q, B]
    row_hi: torch.Tensor      # int64 [Q, B]
    taps: Optional[dict] = None


def search(model: DeviceModel, trie: DeviceTrie, input_ids: torch.Tensor, attention_mask: torch.Tensor,
           num_beams: int, max_new_tokens: int, apply_log_softmax_for_scores: bool = False,
           use_graph: bool = True, taps: bool = False) -> SearchResult:
    """One call of the hot path: encoder + L fused decode/select steps. Asynchronous on the
    current torch stream; results are device tensors."""
    ctx = model.ctx
    dev = ctx.device
    ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
    mask = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
    Q, Lq = ids.shape
    if not taps and Lq % 8 and Lq < 256:
        # Bucket the padded length to a multiple of 8 (extra columns: id 0, mask 0 — masked keys, identical
        # results). The encoder runs on packed rows, so the padding is free on the device, and a stream of
        # batches with different longest queries reuses a handful of captured hipGraphs instead of one per length.
        pad = min(256, (Lq + 7) // 8 * 8) - Lq
        ids = torch.nn.functional.pad(ids, (0, pad))
        mask = torch.nn.functional.pad(mask, (0, pad))
        Lq += pad
    B, L = int(num_beams), int(max_new_tokens)
    tokens = torch.empty((Q, B, L), dtype=torch.int32, device=dev)
    scores = torch.empty((Q, B), dtype=torch.float32, device=dev)
    lo = torch.empty((Q, B), dtype=torch.int64, device=dev)
    hi = torch.empty((Q, B), dtype=torch.int64, device=dev)
    flags = (_lib.FLAG_LOG_SOFTMAX if apply_log_softmax_for_scores else 0) | (0 if use_graph else _lib.FLAG_NO_GRAPH)
    tap_struct, tap_out = None, None
    if taps:
        tap_out = dict(
            encoder_out=torch.empty((Q, Lq, model.d_model), dtype=torch.float32, device=dev),
            step_logits=torch.empty((L, Q * B, model.V), dtype=torch.float32, device=dev),
            step_scores=torch.empty((L, Q, B), dtype=torch.float64, device=dev),
            step_tokens=torch.empty((L, Q, B), dtype=torch.int32, device=dev),
            step_parent=torch.empty((L, Q, B), dtype=torch.int32, device=dev),
            # bit (beam*V + token) of word [t, q, (beam*V + token) // 64]: token is a trie child of the beam (uint64 bits)
            step_valid=torch.zeros((L, Q, B * model.V // 64), dtype=torch.int64, device=dev))
        tap_struct = _lib.DebugTaps(*[tap_out[k].data_ptr() for k in
                                      ("encoder_out", "step_logits", "step_scores", "step_tokens", "step_parent",
                                       "step_valid")])
    check(ctx.lib.rpr_search(ctx.handle, model.handle, trie.handle, ids.data_ptr(), mask.data_ptr(), Q, Lq, B, L, flags,
                             tokens.data_ptr(), scores.data_ptr(), lo.data_ptr(), hi.data_ptr(),
                             C.byref(tap_struct) if tap_struct is not None else None, _stream_ptr(dev)), "rpr_search")
    # ids/mask must stay alive until the async D2D staging copies have been enqueued (they have).
    return SearchResult(tokens, scores, lo, hi, tap_out)


class GuardedSearch:
    """A search whose device-side guards are checked LATER (no host synchronisation when it is issued).

    ``result()`` waits for the status words copied right behind the search and, if a guard fired, repeats the batch
    synchronously with the safe settings before returning: ``STATUS_TAIL_LEFTOVER`` (optimistic forced-tail mode: a
    query was still unforced at the last fork) -> exact forced-tail mode; ``STATUS_SATURATED`` (an activation left the
    f16 plane range of the split-precision GEMMs) -> exact fp32 MFMA. ``STATUS_EMPTY_QUERY`` raises ``ValueError``.
    ``repeated`` tells whether the returned result is a second run (its tensors are then new ones)."""

    def __init__(self, model, trie, ids, mask, B, L, log_softmax, res, ticket, optimistic=False):
        self._args = (model, trie, ids, mask, B, L, log_softmax)
        self._res, self._ticket = res, ticket
        self._optimistic = bool(optimistic)      # this call ran in the optimistic forced-tail mode on the caller's behalf
        self.repeated = False
        self._done = False
        self._error: Optional[BaseException] = None

    def result(self) -> SearchResult:
        if self._done:
            return self._res
        if self._error is not None:   # the guard fired and the repeat failed: the first result is known to be invalid
            raise self._error
        try:
            res = self._result()
        except BaseException as e:
            self._error = e
            raise
        self._done = True
        return res

    def _result(self) -> SearchResult:
        st = self._ticket.flags()
        if st & _lib.STATUS_EMPTY_QUERY:
            raise ValueError("a query has an all-zero attention_mask (no token to attend to)")
        if st & (_lib.STATUS_SATURATED | _lib.STATUS_TAIL_LEFTOVER):
            model, trie, ids, mask, B, L, log_softmax = self._args
            ctx = model.ctx
            saved_mode, saved_prec = ctx.forced_tail(), ctx.get_precision()
            if (st & _lib.STATUS_SATURATED) and saved_prec != "f32":
                import warnings
                warnings.warn("activation outside the f16 plane range of the split-precision GEMMs: repeating this batch "
                              "with exact fp32 MFMA (RPR_PRECISION=f32 avoids the retry)")
                ctx.set_precision("f32")
            if saved_mode == 2:
                ctx.set_forced_tail(1)
            if st & _lib.STATUS_TAIL_LEFTOVER:
                _note_optimistic_outcome(ctx, leftover=True)
            try:
                ctx.status(clear=True)
                self._res = search(model, trie, ids, mask, B, L, apply_log_softmax_for_scores=log_softmax)
                ctx.status(clear=True)
            finally:
                ctx.set_precision(saved_prec)
                ctx.set_forced_tail(saved_mode)
            self.repeated = True
        elif self._optimistic:
            _note_optimistic_outcome(self._args[0].ctx, leftover=False)
        return self._res


# Back-off of the optimistic forced-tail mode (search_guarded). The optimistic mode bets that the last fork leaves no query
# behind; when it loses, the batch is searched twice. On a trie whose popular prefixes stay dense for longer than its node
# statistics suggest (real residual-quantiser codes may) the bet would be lost batch after batch: after two lost bets in a row
# the ctx runs the exact mode for the next OPTIMISTIC_BACKOFF calls, then tries again.
OPTIMISTIC_BACKOFF = 20


def _note_optimistic_outcome(ctx: Context, leftover: bool) -> None:
    if leftover:
        ctx._leftover_streak = getattr(ctx, "_leftover_streak", 0) + 1
        if ctx._leftover_streak >= 2:
            ctx._exact_calls_left = OPTIMISTIC_BACKOFF
    else:
        ctx._leftover_streak = 0


def _optimistic_allowed(ctx: Context) -> bool:
    left = getattr(ctx, "_exact_calls_left", 0)
    if left > 0:
        ctx._exact_calls_left = left - 1
        if ctx._exact_calls_left == 0:
            ctx._leftover_streak = 1      # one more lost bet after the pause and the pause starts again
        return False
    return True


def search_guarded(model: DeviceModel, trie: DeviceTrie, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                   num_beams: int, max_new_tokens: int, apply_log_softmax_for_scores: bool = False,
                   optimistic: Optional[bool] = None) -> GuardedSearch:
    """``search`` plus the status guards, checked when ``result()`` is called (see :class:`GuardedSearch`). With
    ``optimistic`` (default: on unless ``RPR_OPTIMISTIC_TAIL=0``) a ctx in the exact forced-tail mode runs this call in
    the optimistic mode — the last, almost always empty, step-by-step stage is not enqueued — and the guard repeats the
    batch exactly in the rare case a query needed it."""
    import os
    ctx = model.ctx
    if optimistic is None:
        optimistic = os.environ.get("RPR_OPTIMISTIC_TAIL", "1") != "0"
    mode = ctx.forced_tail()
    ctx.clear_status_async()
    went_optimistic = bool(optimistic and mode == 1 and _optimistic_allowed(ctx))
    if went_optimistic:
        ctx.set_forced_tail(2)
    try:
        res = search(model, trie, input_ids, attention_mask, num_beams, max_new_tokens,
                     apply_log_softmax_for_scores=apply_log_softmax_for_scores)
    finally:
        ctx.set_forced_tail(mode)
    ticket = ctx.status_async(clear=True)
    return GuardedSearch(model, trie, input_ids, attention_mask, int(num_beams), int(max_new_tokens),
                         bool(apply_log_softmax_for_scores), res, ticket, optimistic=went_optimistic)


def lngknp_forward(model: DeviceModel, input_ids: torch.Tensor, attention_mask: torch.Tensor, doc_codes: torch.Tensor,
                   teacher_pos: Optional[torch.Tensor] = None, teacher_neg: Optional[torch.Tensor] = None,
                   prefix_lens: Optional[Sequence[int]] = None):
    """Forward of the prefix-oriented ranking fine-tune step (``rpr_lngknp_forward``; reference
    T5SeqAQEncoderForLngKnpMarginMSE.forward, modeling/t5_generative_retriever.py:902-966).

    doc_codes ``[bz, n_docs, L]`` (positive first); teacher_pos / teacher_neg ``[n_prefix, bz]`` aligned with
    ``prefix_lens``. Returns ``(losses float32 [n_prefix] or None, position_scores float32 [bz, n_docs, L])``,
    device tensors, asynchronous on the current stream."""
    ctx = model.ctx
    dev = ctx.device
    ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
    mask = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
    codes = doc_codes.to(device=dev, dtype=torch.int32).contiguous()
    bz, Lq = ids.shape
    _, n_docs, L = codes.shape
    if Lq % 8 and Lq < 256:   # same bucketing as search(): masked padding columns, identical results
        pad = min(256, (Lq + 7) // 8 * 8) - Lq
        ids = torch.nn.functional.pad(ids, (0, pad)); mask = torch.nn.functional.pad(mask, (0, pad)); Lq += pad
    pos_scores = torch.empty((bz, n_docs, L), dtype=torch.float32, device=dev)
    n_prefix = 0 if prefix_lens is None else len(prefix_lens)
    losses = tp = tn = pl = None
    if n_prefix:
        tp = teacher_pos.to(device=dev, dtype=torch.float32).contiguous()
        tn = teacher_neg.to(device=dev, dtype=torch.float32).contiguous()
        assert tuple(tp.shape) == (n_prefix, bz) and tuple(tn.shape) == (n_prefix, bz)
        pl = torch.tensor(list(prefix_lens), dtype=torch.int32, device=dev)
        losses = torch.empty((n_prefix,), dtype=torch.float32, device=dev)
    check(ctx.lib.rpr_lngknp_forward(ctx.handle, model.handle, ids.data_ptr(), mask.data_ptr(), bz, Lq, codes.data_ptr(),
                                     n_docs, L, tp.data_ptr() if n_prefix else None, tn.data_ptr() if n_prefix else None,
                                     pl.data_ptr() if n_prefix else None, n_prefix,
                                     losses.data_ptr() if n_prefix else None, pos_scores.data_ptr(), _stream_ptr(dev)),
          "rpr_lngknp_forward")
    return losses, pos_scores


# ---- training step of the ranking fine-tune: flat gradient / optimizer buffers -----------------------
_PARAM_KINDS = ["shared", "enc_rel", "dec_rel", "enc_fln", "dec_fln", "start", "in_emb", "out_emb", "xkv",
                "enc_ln0", "enc_qkv", "enc_o", "enc_ln1", "enc_wi", "enc_wo",
                "dec_ln0", "dec_qkv", "dec_o", "dec_ln1", "dec_xq", "dec_xo", "dec_ln2", "dec_wi", "dec_wo"]


class TrainState:
    """Flat fp32 buffers of one model replica: gradients and the two AdamW moments, laid out as ``rpr_param_info``
    says (the tensors of ``DeviceModel`` in a fixed order). ``grads`` is what data-parallel ranks all-reduce."""

    def __init__(self, model: DeviceModel):
        self.model = model
        lib, h = model.ctx.lib, model.handle
        self.total = int(lib.rpr_param_total(h))
        n = int(lib.rpr_param_count(h))
        self.layout = []   # (device pointer, numel, offset)
        for i in range(n):
            p, ne, off = C.c_void_p(), C.c_int64(), C.c_int64()
            check(lib.rpr_param_info(h, i, C.byref(p), C.byref(ne), C.byref(off)), "rpr_param_info")
            self.layout.append((int(p.value), int(ne.value), int(off.value)))
        dev = model.ctx.device
        self.grads = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.step = 0
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def named_grads(self) -> Dict[str, torch.Tensor]:
        """Gradients under the reference checkpoint's tensor names (q/k/v, the cross k/v and the codebooks un-stacked)."""
        out = {}
        by_ptr = {ptr: (ne, off) for ptr, ne, off in self.layout}
        for name, t, sl in self.model.named_device_params():
            ne, off = by_ptr[t.data_ptr()]
            g = self.grads[off:off + ne].view(t.shape)
            out[name] = g[sl] if sl is not None else g
        return out


class GradExchange:
    """Data-parallel gradient exchange overlapped with the backward pass (the reference wraps the model in
    DistributedDataParallel, tasks/trainer.py:486: bucketed all-reduce running under ``loss.backward()``).

    ``rpr_lngknp_backward_buckets`` hands the flat gradient buffer over in buckets (one per transformer layer in the order
    the backward finishes them, then one for the embeddings / codebooks / cross K/V / norms in front of the first layer):
    for each, ``on_bucket(offset, numel)`` enqueues an asynchronous all-reduce of that slice on a communication stream
    that already waits for the slice's producers. ``finish()`` joins the stream and divides by the world size (DDP's
    averaging). RCCL ("nccl" backend) on GPUs: a bucket is 28-38 MB for t5-base — large messages, as the per-link-bound
    ring on the point-to-point xGMI mesh wants; any torch.distributed backend works (the CPU tests use gloo).
    With one rank (or no process group) nothing is enqueued."""

    def __init__(self, grads: torch.Tensor, dry_run: bool = False, mode: Optional[str] = None):
        """``dry_run`` (tests, one rank): take the buckets and touch each slice on the communication stream instead of
        all-reducing it — the hand-off (callback order, stream dependencies, coverage) without a process group.

        ``mode`` (default: ``RPR_GRAD_EXCHANGE`` or "allreduce"): "allreduce" = one ``all_reduce`` per bucket, the
        algorithm is RCCL's choice; "mesh" = the exchange SURVEY §8 f4 asks for on the point-to-point xGMI mesh: every
        rank owns 1/W of a bucket, receives that shard from all peers at once over the W-1 direct links
        (``all_to_all_single``), sums the W copies in rank order (every rank adds in the same order: the reduced shard has
        one value), and the shards are gathered back (``all_gather_into_tensor``) — reduce-scatter + all-gather in two
        single-hop steps instead of a ring of 2(W-1) per-link-bound steps. Never timed on hardware (no multi-GPU box)."""
        import os
        import torch.distributed as dist
        self.grads = grads
        self.mode = (mode or os.environ.get("RPR_GRAD_EXCHANGE", "allreduce")).lower()
        if self.mode not in ("allreduce", "mesh"):
            raise ValueError(f"unknown gradient exchange mode {self.mode!r} (allreduce | mesh)")
        self._scratch = {}                   # mesh mode: send / receive buffers per padded bucket size
        self.dry_run = bool(dry_run)
        self.active = self.dry_run or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        self.world = dist.get_world_size() if (self.active and not self.dry_run) else 1
        self.stream = torch.cuda.Stream(grads.device) if (self.active and grads.is_cuda) else None
        self.works: List = []
        self.buckets: List[tuple] = []
        self.history: List[tuple] = []       # buckets of the last finished exchange, in hand-over order
        self.error: Optional[BaseException] = None   # first exception raised inside the ctypes callback
        self._cb = _lib.GRAD_BUCKET_CB(self._on_bucket_c)

    def _on_bucket_c(self, _user, offset, numel):
        # ctypes prints and swallows an exception raised inside a callback: the C side would carry on, finish()'s coverage
        # check would pass and AdamW would run on a bucket that was never reduced. Keep the first error; lngknp_backward /
        # finish() re-raise it, and nothing is enqueued after it.
        if self.error is not None:
            return
        try:
            self.on_bucket(int(offset), int(numel))
        except BaseException as e:   # noqa: BLE001 - must not escape into ctypes
            self.error = e

    def on_bucket(self, offset: int, numel: int):
        if not self.active or numel <= 0:
            self.buckets.append((offset, numel))
            return
        sl = self.grads[offset:offset + numel]
        if self.dry_run:
            if self.stream is not None:
                with torch.cuda.stream(self.stream):
                    sl.mul_(1.0)          # reads and rewrites the bucket on the communication stream
        elif self.stream is not None:
            with torch.cuda.stream(self.stream):
                self._exchange(sl)
        else:
            self._exchange(sl)
        self.buckets.append((offset, numel))   # recorded only once its collective has been enqueued

    def raise_pending(self):
        """Re-raise an exception caught inside the bucket callback (and forget the half-done exchange)."""
        if self.error is not None:
            e, self.error = self.error, None
            for w in self.works:   # collectives already in flight must not outlive the buffers they reduce
                try:
                    w.wait()
                except Exception:
                    pass
            if self.stream is not None:
                torch.cuda.current_stream(self.grads.device).wait_stream(self.stream)
            self.works, self.buckets = [], []
            raise RiporHipError(f"gradient exchange failed inside the bucket callback: {e!r}") from e

    def _exchange(self, sl: torch.Tensor):
        import torch.distributed as dist
        if self.mode == "allreduce":
            self.works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, async_op=True))
            return
        W, n = self.world, sl.numel()
        per = (n + W - 1) // W
        if per not in self._scratch:         # a handful of distinct bucket sizes (decoder layer, encoder layer, the rest)
            self._scratch[per] = (torch.zeros(W * per, dtype=sl.dtype, device=sl.device),
                                  torch.empty(W * per, dtype=sl.dtype, device=sl.device),
                                  torch.empty(W * per, dtype=sl.dtype, device=sl.device))
        send, recv, full = self._scratch[per]
        send[:n].copy_(sl)                   # the tail of the last shard stays zero
        dist.all_to_all_single(recv, send, async_op=True).wait()     # recv[i] = rank i's copy of MY shard
        shard = recv.view(W, per)[0].clone()
        for i in range(1, W):                # fixed order: bitwise the same sum on every rank for its shard
            shard += recv.view(W, per)[i]
        dist.all_gather_into_tensor(full, shard, async_op=True).wait()
        sl.copy_(full[:n])

    def comm_stream_ptr(self):
        return C.c_void_p(self.stream.cuda_stream) if self.stream is not None else None

    def callback(self):
        return self._cb if self.active else C.cast(None, _lib.GRAD_BUCKET_CB)

    def finish(self):
        """Join the exchange (the current stream waits for every bucket) and average."""
        self.raise_pending()
        for w in self.works:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream(self.grads.device).wait_stream(self.stream)
        covered = sum(n for _, n in self.buckets)
        if self.active:
            assert covered == self.grads.numel(), (covered, self.grads.numel())   # every element exchanged exactly once
            if self.world > 1:
                self.grads.div_(self.world)
        self.history = list(self.buckets)
        self.works, self.buckets = [], []


def lngknp_backward(model: DeviceModel, state: TrainState, input_ids, attention_mask, doc_codes, teacher_pos, teacher_neg,
                    prefix_lens: Sequence[int], exchange: Optional[GradExchange] = None) -> torch.Tensor:
    """Forward + backward of the sum of the margin-MSE losses (``rpr_lngknp_backward``): fills ``state.grads`` and
    returns the losses ``[n_prefix]`` (device tensor). Asynchronous on the current stream. With ``exchange`` the
    gradient all-reduce of every bucket is enqueued while the backward is still running (``GradExchange``); the caller
    then calls ``exchange.finish()`` before the optimizer step."""
    ctx = model.ctx
    dev = ctx.device
    ids = input_ids.to(device=dev, dtype=torch.int32).contiguous()
    mask = attention_mask.to(device=dev, dtype=torch.int32).contiguous()
    codes = doc_codes.to(device=dev, dtype=torch.int32).contiguous()
    bz, Lq = ids.shape
    assert codes.shape[0] == bz and codes.shape[1] == 2
    L = codes.shape[2]
    n_prefix = len(prefix_lens)
    tp = teacher_pos.to(device=dev, dtype=torch.float32).contiguous()
    tn = teacher_neg.to(device=dev, dtype=torch.float32).contiguous()
    assert tuple(tp.shape) == (n_prefix, bz) and tuple(tn.shape) == (n_prefix, bz)
    pl = torch.tensor(list(prefix_lens), dtype=torch.int32, device=dev)
    losses = torch.empty((n_prefix,), dtype=torch.float32, device=dev)
    if exchange is not None and exchange.active:
        check(ctx.lib.rpr_lngknp_backward_buckets(ctx.handle, model.handle, ids.data_ptr(), mask.data_ptr(), bz, Lq,
                                                  codes.data_ptr(), L, tp.data_ptr(), tn.data_ptr(), pl.data_ptr(), n_prefix,
                                                  losses.data_ptr(), state.grads.data_ptr(), _stream_ptr(dev),
                                                  exchange.comm_stream_ptr(), exchange.callback(), None),
              "rpr_lngknp_backward_buckets")
        exchange.raise_pending()
    else:
        check(ctx.lib.rpr_lngknp_backward(ctx.handle, model.handle, ids.data_ptr(), mask.data_ptr(), bz, Lq, codes.data_ptr(), L,
                                          tp.data_ptr(), tn.data_ptr(), pl.data_ptr(), n_prefix, losses.data_ptr(),
                                          state.grads.data_ptr(), _stream_ptr(dev)), "rpr_lngknp_backward")
    return losses


def allreduce_grads(state: TrainState, bucket_elems: int = 64 << 20) -> None:
    """Serial data-parallel gradient exchange (``RPR_GRAD_OVERLAP=0``; the default is :class:`GradExchange`, overlapped
    with the backward pass): sum ``state.grads`` over the ranks in a few large chunks after the backward and divide by
    the world size (DDP's gradient averaging). RCCL ("nccl" backend) on GPUs; any torch.distributed backend works. The
    MI355X xGMI mesh is point-to-point, so few large messages are what RCCL's ring needs: 256 MB chunks by default."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    g = state.grads
    for s in range(0, g.numel(), bucket_elems):
        dist.all_reduce(g[s:s + bucket_elems], op=dist.ReduceOp.SUM)
    g.div_(world)


def allreduce_mode() -> str:
    import os
    if os.environ.get("RPR_GRAD_OVERLAP", "1") == "0":
        return "serial"
    how = "all_to_all + all_gather per bucket (mesh)" if os.environ.get("RPR_GRAD_EXCHANGE", "allreduce").lower() == "mesh" else "all_reduce per bucket"
    return f"bucketed, overlapped with the backward pass, {how}"


def train_step(model: DeviceModel, state: TrainState, input_ids, attention_mask, doc_codes, teacher_pos, teacher_neg,
               prefix_lens: Sequence[int], lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
               max_grad_norm: float = 1.0) -> torch.Tensor:
    """One optimisation step as the reference's trainer performs it (tasks/trainer.py:203-275 + HF Trainer defaults):
    forward + backward, gradient all-reduce across the data-parallel ranks — bucketed and overlapped with the backward
    (``RPR_GRAD_OVERLAP=0``: one serial pass afterwards) —, clip_grad_norm_, AdamW. Returns the losses before the update."""
    import os
    if os.environ.get("RPR_GRAD_OVERLAP", "1") == "0":
        losses = lngknp_backward(model, state, input_ids, attention_mask, doc_codes, teacher_pos, teacher_neg, prefix_lens)
        allreduce_grads(state)
    else:
        ex = getattr(state, "_exchange", None)
        if ex is None or ex.grads is not state.grads:
            ex = state._exchange = GradExchange(state.grads)
        losses = lngknp_backward(model, state, input_ids, attention_mask, doc_codes, teacher_pos, teacher_neg, prefix_lens,
                                 exchange=ex)
        ex.finish()
    adamw_step(model, state, lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm)
    return losses


def adamw_step(model: DeviceModel, state: TrainState, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
               weight_decay: float = 0.0, max_grad_norm: float = 1.0) -> None:
    """``clip_grad_norm_(max_grad_norm)`` + ``torch.optim.AdamW`` on the model's device tensors, in place
    (``rpr_adamw_step``); ``state.grad_norm`` receives the pre-clip global norm."""
    ctx = model.ctx
    state.step += 1
    check(ctx.lib.rpr_adamw_step(ctx.handle, model.handle, state.grads.data_ptr(), state.exp_avg.data_ptr(),
                                 state.exp_avg_sq.data_ptr(), state.step, lr, betas[0], betas[1], eps, weight_decay,
                                 max_grad_norm, state.grad_norm.data_ptr(), _stream_ptr(ctx.device)), "rpr_adamw_step")
