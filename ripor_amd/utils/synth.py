"""Deterministic synthetic inputs for the constrained-beam-search path.

There is no network in the build/bench environment, so checkpoints, the
MS MARCO ``docid_to_smtid.json`` and the dev queries are replaced by synthetic
data of the same *shape* (BASELINE.json ``configs``).

Everything here comes from a counter-based integer hash (splitmix64) so that the
build container, the GPU box and the golden-fixture generator all produce
bit-identical arrays without depending on any library RNG stream:

    value(name, i) = ((splitmix64(fnv1a64(name) + i) >> 40) - 2**23) / 2**23 * scale

which is exact in float64 and rounds once to float32.

State-dict key names and shapes follow the reference checkpoint layout
(reference: t5_pretrainer/modeling/t5_generative_retriever.py:84-112, HF ``T5Stack``
parameter names).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

SEED = 20240928
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_u64(name: str, n: int, seed: int = SEED, offset: int = 0) -> np.ndarray:
    base = np.uint64((fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + n, dtype=np.uint64) + base
    return splitmix64(ctr)


_CHUNK = 1 << 18          # elements per work item: 2 MB of uint64 — the temporaries of a chunk stay in the cache
_POOL = None


def _pool():
    """Worker threads for the generators below (numpy's ufuncs release the GIL). The value of element i depends on i
    only, so the chunking and the number of threads cannot change a bit of the result."""
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        try:
            n = len(os.sched_getaffinity(0))
        except AttributeError:
            n = os.cpu_count() or 1
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, n)))
    return _POOL


def _hash_chunk(base: np.uint64, s: int, e: int) -> np.ndarray:
    """splitmix64(base + i) for i in [s, e), in place on one buffer."""
    with np.errstate(over="ignore"):
        z = np.arange(s, e, dtype=np.uint64)
        z += base
        z += np.uint64(0x9E3779B97F4A7C15)
        t = z >> np.uint64(30)
        z ^= t
        z *= np.uint64(0xBF58476D1CE4E5B9)
        np.right_shift(z, np.uint64(27), out=t)
        z ^= t
        z *= np.uint64(0x94D049BB133111EB)
        np.right_shift(z, np.uint64(31), out=t)
        z ^= t
    return z


def _name_base(name: str, seed: int) -> np.uint64:
    return np.uint64((fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF)


def uniform_f32(name: str, shape, scale: float, seed: int = SEED) -> np.ndarray:
    """Uniform in [-scale, scale) with 24-bit resolution; exact & reproducible."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    base = _name_base(name, seed)
    mul = scale / float(1 << 23)

    def work(s):
        e = min(n, s + _CHUNK)
        h = _hash_chunk(base, s, e)
        h >>= np.uint64(40)
        v = h.astype(np.int64)
        v -= (1 << 23)
        out[s:e] = (v.astype(np.float64) * mul).astype(np.float32)

    starts = range(0, n, _CHUNK)
    if n <= _CHUNK:
        for s in starts:
            work(s)
    else:
        list(_pool().map(work, starts))
    return out.reshape(shape)


def randint(name: str, shape, lo: int, hi: int, seed: int = SEED) -> np.ndarray:
    """Integers uniform in [lo, hi) (modulo bias is irrelevant for synthetic data)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.int64)
    span = np.uint64(hi - lo)
    base = _name_base(name, seed)

    def work(s):
        e = min(n, s + _CHUNK)
        h = _hash_chunk(base, s, e)
        h >>= np.uint64(11)
        h %= span
        out[s:e] = h.astype(np.int64) + lo

    starts = range(0, n, _CHUNK)
    if n <= _CHUNK:
        for s in starts:
            work(s)
    else:
        list(_pool().map(work, starts))
    return out.reshape(shape)


# --------------------------------------------------------------------------- model dims


@dataclass
class ModelDims:
    """Dimensions of a ``T5ForDocIDGeneration`` checkpoint (T5forDocIDConfig fields)."""

    vocab_size: int = 32128
    d_model: int = 768
    d_kv: int = 64
    d_ff: int = 3072
    num_layers: int = 12
    num_decoder_layers: int = 12
    num_heads: int = 12
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    layer_norm_epsilon: float = 1e-6
    decoder_vocab_sizes: List[int] = field(default_factory=lambda: [256] * 32)
    shared_output_input_embeds: bool = False
    scaleup_output_hidden: bool = False

    @property
    def inner(self) -> int:
        return self.num_heads * self.d_kv


def t5_base_dims(L: int = 32, V: int = 256, **kw) -> ModelDims:
    return ModelDims(decoder_vocab_sizes=[V] * L, **kw)


def t5_large_dims(L: int = 32, V: int = 256, **kw) -> ModelDims:
    return ModelDims(d_model=1024, d_kv=64, d_ff=4096, num_layers=24, num_decoder_layers=24,
                     num_heads=16, decoder_vocab_sizes=[V] * L, **kw)


def mini_dims(L: int = 8, V: int = 256, enc_layers: int = 2, d_ff: int = 256,
              vocab_size: int = 512, **kw) -> ModelDims:
    """Cheap golden model: the reference ctor forces (12 decoder layers, 12 heads, d=768)
    (t5_generative_retriever.py:116-121); encoder depth, d_ff and vocab are free."""
    return ModelDims(vocab_size=vocab_size, d_ff=d_ff, num_layers=enc_layers,
                     decoder_vocab_sizes=[V] * L, **kw)


def make_state_dict(dims: ModelDims, seed: int = SEED, logit_scale: float = 0.35, outliers: float = 0.0) -> Dict[str, np.ndarray]:
    """Seeded weights under the reference's state-dict key names (float32 numpy arrays).

    Scales follow HF's T5 init factors so activations stay O(1); layer-norm weights are
    perturbed around 1 and the relative-bias tables are O(1) so that a kernel that drops
    either is caught by the parity tests. ``logit_scale`` sets the output-codebook spread
    (logits come out O(10), like a trained model's dot-product scores).

    ``outliers`` > 0: the activation statistics trained T5 checkpoints are known for (the reason HF clamps fp16 T5), which the
    N(0, sigma) weights above do not have — (1) three residual-stream channels of about ``outliers`` (e.g. 1e5) in both stacks:
    large embedding columns, and feed-forward output rows of one sign on those channels whose size rises geometrically with
    depth; (2) layer-norm weights that are tiny on the outlier channels and spread up to ~8 elsewhere;
    (3) four hidden units per feed-forward block with 150 x larger input weights (the most the f16 weight planes carry) against
    tiny output weights. The residual stream then carries |x| of 0.4-0.6 x ``outliers`` from the embedding to the last block.
    The model is still an ordinary T5 state dict: the CPU oracle and the exact-fp32 path run it unchanged.
    """
    d, inner, dff = dims.d_model, dims.inner, dims.d_ff
    sd: Dict[str, np.ndarray] = {}
    r3 = 3.0 ** 0.5  # uniform[-a,a) has std a/sqrt(3)

    def u(name, shape, std):
        sd[name] = uniform_f32(name, shape, std * r3, seed)

    def ln(name):
        sd[name] = (1.0 + uniform_f32(name, (d,), 0.2, seed)).astype(np.float32)

    u("shared.weight", (dims.vocab_size, d), 1.0)

    def attn(prefix, has_bias):
        u(prefix + ".q.weight", (inner, d), (d * dims.d_kv) ** -0.5)
        u(prefix + ".k.weight", (inner, d), d ** -0.5)
        u(prefix + ".v.weight", (inner, d), d ** -0.5)
        u(prefix + ".o.weight", (d, inner), inner ** -0.5)
        if has_bias:
            u(prefix + ".relative_attention_bias.weight",
              (dims.relative_attention_num_buckets, dims.num_heads), 0.5)

    for i in range(dims.num_layers):
        p = f"encoder.block.{i}.layer"
        attn(p + ".0.SelfAttention", i == 0)
        ln(p + ".0.layer_norm.weight")
        u(p + ".1.DenseReluDense.wi.weight", (dff, d), d ** -0.5)
        u(p + ".1.DenseReluDense.wo.weight", (d, dff), dff ** -0.5)
        ln(p + ".1.layer_norm.weight")
    ln("encoder.final_layer_norm.weight")

    for i in range(dims.num_decoder_layers):
        p = f"decoder.block.{i}.layer"
        attn(p + ".0.SelfAttention", i == 0)
        ln(p + ".0.layer_norm.weight")
        attn(p + ".1.EncDecAttention", False)
        ln(p + ".1.layer_norm.weight")
        u(p + ".2.DenseReluDense.wi.weight", (dff, d), d ** -0.5)
        u(p + ".2.DenseReluDense.wo.weight", (d, dff), dff ** -0.5)
        ln(p + ".2.layer_norm.weight")
    ln("decoder.final_layer_norm.weight")

    for i, V in enumerate(dims.decoder_vocab_sizes):
        u(f"list_decoder_embeds.{i}.weight", (V, d), 1.0)
        if not dims.shared_output_input_embeds:
            u(f"list_output_embeds.{i}.weight", (V, d), logit_scale)
    u("start_token_embed", (1, 1, d), 1.0)
    if outliers > 0.0:
        _add_outliers(sd, dims, float(outliers), seed)
    return sd


def _add_outliers(sd: Dict[str, np.ndarray], dims: ModelDims, target: float, seed: int) -> None:
    """See make_state_dict(outliers=...). In place."""
    d, dff = dims.d_model, dims.d_ff
    chans = np.array([17 % d, 301 % d, 642 % d])
    base = 0.15 * target                                     # outlier size right after the embedding (massive from the start,
                                                             # as in trained T5 stacks; the blocks below grow it ~4 x)
    for name in ["shared.weight", "start_token_embed"] + [f"list_decoder_embeds.{i}.weight" for i in range(len(dims.decoder_vocab_sizes))]:
        w = sd[name]
        w[..., chans] = np.abs(w[..., chans]) * base + base
    lnw = 0.5 + 7.5 * (uniform_f32("outlier/ln", (d,), 0.5, seed) + 0.5) ** 3      # most weights ~1, a tail up to ~8
    lnw[chans] = 0.02
    for stack, n, ff in (("encoder", dims.num_layers, 1), ("decoder", dims.num_decoder_layers, 2)):
        prev = base * 1.5
        for i in range(n):
            p = f"{stack}.block.{i}.layer"
            for k in range(ff + 1):
                sd[f"{p}.{k}.layer_norm.weight"] = (lnw * (1.0 + 0.1 * uniform_f32(f"outlier/{stack}{i}{k}", (d,), 1.0, seed))).astype(np.float32)
            wi, wo = sd[f"{p}.{ff}.DenseReluDense.wi.weight"], sd[f"{p}.{ff}.DenseReluDense.wo.weight"]
            # residual outliers: this block adds (next - prev) to the outlier channels — its ReLU output u >= 0 meets output
            # rows of one sign; sum_j |wo[c, j]| u_j ~ 0.3 * 0.8 * dff^-0.5 * dff * rms(u) per unit of row scale
            nxt = base * 1.5 * (target / (base * 1.5)) ** ((i + 1) / n)
            g = (nxt - prev) / (0.24 * dff ** 0.5 * 0.12)     # 0.12: measured rms of the ReLU output behind the squashed norm
            wo[chans, :] = np.minimum(np.abs(wo[chans, :]) * g, 200.0)   # (the f16 weight planes carry |w| < 255)
            prev = nxt
            # a few hidden units with large pre-activations and tiny output weights (as large as the f16 weight planes allow:
            # |w x layer-norm weight| < 255, common.h): the FF intermediate's own range is exercised by test_gpu_edges.py
            units = (np.arange(4) * 769 + 31 * i) % dff
            wi[units, :] *= 150.0
            wo[:, units] *= 1e-3
    for name in ("encoder.final_layer_norm.weight", "decoder.final_layer_norm.weight"):
        sd[name] = lnw.astype(np.float32).copy()


# --------------------------------------------------------------------------- docid codes


def zipf_tokens(name: str, shape, V: int, s: float = 1.0, seed: int = SEED) -> np.ndarray:
    """Tokens in [0, V) with P(token = k) proportional to (k + 1)^-s (Zipf over the V symbols; token 0 the most frequent):
    inverse-CDF sampling of 40-bit uniform variates from the counter hash. s = 1.0 on the first three levels is SURVEY.md
    §8(d)'s stand-in for the imbalance of residual-quantiser codes (reference
    aq_preprocess/create_customized_smtid_file.py:33-59 writes whatever the RQ index assigned)."""
    w = 1.0 / np.power(np.arange(1, V + 1, dtype=np.float64), float(s))
    cdf = np.cumsum(w / w.sum())
    cdf[-1] = 1.0
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.int64)
    chunk = 1 << 23
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        u = (hash_u64(name, b - a, seed, offset=a) >> np.uint64(24)).astype(np.float64) / float(1 << 40)
        out[a:b] = np.minimum(np.searchsorted(cdf, u, side="right"), V - 1)
    return out.reshape(shape)


def make_codes(N: int, L: int, V: int, seed: int = SEED, skew: bool = False, zipf: Optional[float] = None) -> np.ndarray:
    """Synthetic ``docid_to_smtid`` code matrix ``[N, L]`` (docid = row index), i.i.d. uniform
    tokens. ``zipf`` = s draws the first three levels from Zipf(s) (§8d's skewed variant, s = 1.0);
    ``skew`` (older, milder) squares the uniform variate on those levels."""
    dt = np.uint8 if V <= 256 else np.uint16
    codes = randint(f"codes/{N}x{L}x{V}", (N, L), 0, V, seed)
    lv = min(3, L)
    if zipf is not None:
        codes[:, :lv] = zipf_tokens(f"codes_zipf/{N}x{V}", (N, lv), V, zipf, seed)
    elif skew:
        u = randint(f"codes_skew/{N}", (N, lv), 0, 1 << 20, seed).astype(np.float64) / float(1 << 20)
        codes[:, :lv] = np.minimum((u * u * V).astype(np.int64), V - 1)
    return codes.astype(dt)


def codes_to_docid_to_smtid(codes: np.ndarray) -> Dict[str, List[int]]:
    """The reference's on-disk format: ``{"docid": [-1, c1, ..., cL]}``
    (aq_preprocess/create_customized_smtid_file.py:47-59)."""
    return {str(i): [-1] + [int(x) for x in row] for i, row in enumerate(codes)}


# --------------------------------------------------------------------------- queries


def make_queries(Q: int, vocab_size: int = 32128, seed: int = SEED, mean_len: float = 12.0,
                 std_len: float = 4.0, min_len: int = 6, max_len: int = 32,
                 fixed_len: Optional[int] = None):
    """MSMARCO-dev-shaped tokenised queries: ``input_ids, attention_mask`` ``[Q, Lq]`` int64,
    padded with 0 to the batch maximum (pad-to-longest like the reference collator,
    dataset/dataloader.py:62-79). First three ids are fixed ("query", ":", "▁"-like), last
    valid id is 1 (``</s>``). Lengths ~ clipped N(mean, std) via an Irwin-Hall(12) variate."""
    if fixed_len is not None:
        lens = np.full(Q, fixed_len, dtype=np.int64)
    else:
        u = randint(f"qlen/{Q}", (Q, 12), 0, 1 << 20, seed).astype(np.float64) / float(1 << 20)
        z = u.sum(axis=1) - 6.0  # ~N(0,1)
        lens = np.clip(np.rint(mean_len + std_len * z), min_len, max_len).astype(np.int64)
    Lq = int(lens.max())
    ids = randint(f"qtok/{Q}", (Q, Lq), 3, min(32000, vocab_size), seed)
    fixed = [min(11417, vocab_size - 1), min(10, vocab_size - 1), min(3, vocab_size - 1)]
    ids[:, 0], ids[:, 1], ids[:, 2] = fixed
    pos = np.arange(Lq)[None, :]
    mask = (pos < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[np.arange(Q), lens - 1] = 1
    return ids, mask


def make_codes_fast(N: int, L: int, V: int, seed: int = SEED, zipf: Optional[float] = None) -> np.ndarray:
    """Large-trie variant of :func:`make_codes` (8.8 M x 32): one hash yields four codes
    (16 bits each, reduced mod V), ~4x fewer hash evaluations. Distribution: i.i.d. uniform; ``zipf`` = s
    redraws the first three levels from Zipf(s) (SURVEY.md §8d's skewed trie)."""
    n = N * L
    n4 = (n + 3) // 4
    out = np.empty(n4 * 4, dtype=np.uint16)
    base = _name_base(f"codes_fast/{N}x{L}x{V}", seed)

    def work(s):
        e = min(n4, s + _CHUNK)
        c = _hash_chunk(base, s, e).view(np.uint16)
        if V < 65536:
            c %= np.uint16(V)
        out[4 * s:4 * e] = c

    list(_pool().map(work, range(0, n4, _CHUNK)))
    out = out[:n].reshape(N, L)
    if zipf is not None:
        lv = min(3, L)
        out[:, :lv] = zipf_tokens(f"codes_fast_zipf/{N}x{V}", (N, lv), V, zipf, seed).astype(np.uint16)
    return out


def patterned_state_dict(dims: ModelDims) -> Dict[str, np.ndarray]:
    """A state dict of the checkpoint's tensor names with low-entropy, exactly representable values: element i of tensor
    `name` = ((crc32(name) % 7) + (i % 11) - 8) / 16. For fixtures that store a checkpoint DIRECTORY (tests/golden/
    c9_ref_checkpoint.zip, written by the reference's save_pretrained): the zip stays small and a reader can be checked
    tensor by tensor against the formula."""
    import zlib
    out = {}
    for name, arr in make_state_dict(dims, seed=1).items():
        n = int(np.prod(arr.shape))
        v = ((zlib.crc32(name.encode()) % 7) + (np.arange(n, dtype=np.int64) % 11) - 8).astype(np.float32) / 16.0
        out[name] = v.reshape(arr.shape)
    return out
