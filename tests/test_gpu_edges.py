"""-m gpu: edge cases of the search path against the CPU oracle (KV-cached variant, itself pinned to the
reference by tests/test_oracle_golden.py): greedy B=1 (BASELINE config 1; the reference's HF scorer refuses
num_beams=1, the oracle treats it as the degenerate case), single query, L=1, long queries (Lq up to 200,
multi-chunk attention paths), masks that are not a prefix (left padding, holes), ragged batches whose
row count is not a multiple of any GEMM tile, exact-fp32 precision mode."""
import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def setup():
    from oracle import beam_ref, t5_ref
    from ripor_amd import engine as E
    from ripor_amd.utils import synth
    L, V, N = 6, 256, 400
    dims = synth.mini_dims(L=L, V=V, enc_layers=2, d_ff=256, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=31)
    codes = synth.make_codes(N, L, V, seed=31)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)

    def oracle(ids, mask, B, Lx):
        m = t5_ref.T5RefCached(sd, dims)
        seqs, sc = beam_ref.beam_search_ref(m, pm, ids, mask, B, Lx, use_kv_cache=True)
        Q = ids.shape[0]
        return seqs.numpy().reshape(Q, B, Lx + 1)[:, :, 1:], sc.numpy().reshape(Q, B)

    def hip(ids, mask, B, Lx, **kw):
        r = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, Lx, **kw)
        torch.cuda.synchronize()
        return r.tokens.cpu().numpy(), r.scores.cpu().numpy()

    yield dict(E=E, ctx=ctx, dims=dims, oracle=oracle, hip=hip, synth=synth, L=L)
    ctx.set_precision("f16x2")


def _check(s, ids, mask, B, Lx, **kw):
    et, es = s["oracle"](ids, mask, B, Lx)
    gt, gs = s["hip"](ids, mask, B, Lx, **kw)
    assert np.array_equal(gt, et), "smtid sequences differ from the oracle"
    np.testing.assert_allclose(gs, es, atol=1e-4, rtol=0)


def test_greedy_single_beam(setup):
    ids, mask = setup["synth"].make_queries(5, vocab_size=512, seed=3, max_len=14)
    _check(setup, ids, mask, 1, setup["L"])


def test_single_query_and_single_step(setup):
    ids, mask = setup["synth"].make_queries(1, vocab_size=512, seed=4, max_len=10)
    _check(setup, ids, mask, 3, setup["L"])
    _check(setup, ids, mask, 4, 1)


def test_ragged_row_counts(setup):
    # Q*B = 7*9 = 63 rows and Q*Lq tokens not multiples of 32/64/128
    ids, mask = setup["synth"].make_queries(7, vocab_size=512, seed=5, max_len=13)
    _check(setup, ids, mask, 9, 4)


def test_long_queries_multi_chunk_attention(setup):
    ids, mask = setup["synth"].make_queries(3, vocab_size=512, seed=6, min_len=70, max_len=200, mean_len=130, std_len=50)
    assert ids.shape[1] > 64
    _check(setup, ids, mask, 4, 3)


def test_non_prefix_masks(setup):
    ids, mask = setup["synth"].make_queries(4, vocab_size=512, seed=7, max_len=16)
    Lq = ids.shape[1]
    # query 0: left padding (valid tokens moved to the right end); query 1: a hole in the middle
    n0 = int(mask[0].sum())
    ids2, mask2 = ids.copy(), mask.copy()
    ids2[0] = 0; mask2[0] = 0
    ids2[0, Lq - n0:] = ids[0, :n0]; mask2[0, Lq - n0:] = 1
    mask2[1, 2] = 0
    _check(setup, ids2, mask2, 4, 4)


def test_exact_fp32_mode_matches_oracle(setup):
    setup["ctx"].set_precision("f32")
    try:
        ids, mask = setup["synth"].make_queries(6, vocab_size=512, seed=8, max_len=15)
        _check(setup, ids, mask, 5, setup["L"])
        _check(setup, ids, mask, 5, setup["L"], use_graph=False)
    finally:
        setup["ctx"].set_precision("f16x2")


def test_argument_errors_cross_the_abi_as_exceptions(setup):
    E = setup["E"]
    ids, mask = setup["synth"].make_queries(2, vocab_size=512, seed=9, max_len=10)
    with pytest.raises(E.RiporHipError, match="exceeds"):
        setup["hip"](ids, mask, 2, setup["L"] + 1)       # L beyond the model's decoder length / trie depth
    with pytest.raises(E.RiporHipError, match="out of range|NULL argument"):
        setup["hip"](ids, mask, 0, 2)                    # B < 1 (empty output buffers)
    big = np.ones((1, 300), dtype=np.int64)
    with pytest.raises(E.RiporHipError, match="Lq out of range"):
        setup["hip"](big, big, 2, 2)


def test_beam_too_large_is_refused_with_a_message(setup):
    E = setup["E"]
    ids = np.array([[5, 6, 1]], dtype=np.int64)
    with pytest.raises(E.RiporHipError, match="too large for the select kernel"):
        setup["hip"](ids, np.ones_like(ids), 4000, setup["L"])


def test_scaleup_output_hidden_and_1024_codebook(setup):
    """config.scaleup_output_hidden (reference t5_generative_retriever.py:427-428) and the 16 x 1024 code
    layout (full_16_1024_scripts): a second model built here, compared with the oracle."""
    from oracle import beam_ref, t5_ref
    E, synth = setup["E"], setup["synth"]
    L, V, N = 5, 1024, 700
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512, scaleup_output_hidden=True)
    sd = synth.make_state_dict(dims, seed=41, logit_scale=8.0)   # logits are scaled by 768**-0.5: keep them O(10)
    codes = synth.make_codes(N, L, V, seed=41)
    ctx = setup["ctx"]
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ids, mask = synth.make_queries(5, vocab_size=512, seed=42, max_len=12)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, 6, L, use_kv_cache=True)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), 6, L)
    torch.cuda.synchronize()
    assert np.array_equal(res.tokens.cpu().numpy(), seqs.numpy().reshape(5, 6, L + 1)[:, :, 1:])
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(5, 6), atol=1e-4, rtol=0)


def test_skewed_trie_with_duplicate_smtids(setup):
    """SURVEY §8d skewed variant: squared-uniform codes on the first three levels (RQ code imbalance) over a
    short code (L=4, few distinct tokens on the last level), so prefixes are shared by many docs, several docs
    collide on the full smtid (range sizes > 1) and some nodes have fewer than B children."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q = 4, 256, 3000, 8, 6
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=53)
    codes = synth.make_codes(N, L, V, seed=53, skew=True)
    codes[:, 0] //= 16; codes[:, 1] %= 6; codes[:, 2] %= 6; codes[:, 3] %= 3  # narrow levels -> many docs per smtid
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    d2s = synth.codes_to_docid_to_smtid(codes)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), V)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=54, max_len=10)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    torch.cuda.synchronize()
    tok = res.tokens.cpu().numpy()
    assert np.array_equal(tok, seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:])
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(Q, B), atol=1e-4, rtol=0)
    # every returned range is exactly the set of docs carrying that smtid, in docid order (evaluate.py:439-446)
    s2d = beam_ref.build_smtid_to_docids(d2s, L)
    perm = trie.perm
    lo, hi = res.row_lo.cpu().numpy(), res.row_hi.cpu().numpy()
    sizes = []
    for q in range(Q):
        for b in range(B):
            key = "_".join(str(int(x)) for x in tok[q, b])
            docs = [str(int(d)) for d in perm[lo[q, b]:hi[q, b]]]
            assert docs == s2d[key]
            sizes.append(len(docs))
    assert max(sizes) > 1, "the skewed fixture is meant to contain smtids shared by several docs"


def test_beam_1000_like_the_reference_retrieval_script(setup):
    """full_scripts/full_evaluate_t5seq_aq_encoder.sh:199 runs the retrieval with --topk=1000. More beams than trie
    children at the first levels (so -1e9 candidates are selected and later die out), beam-chunked cross-attention,
    the global-memory path of the select kernel (B*V logits do not fit LDS)."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q = 3, 256, 6000, 1000, 2
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=61)
    codes = synth.make_codes(N, L, V, seed=61)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=62, max_len=9)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    torch.cuda.synchronize()
    exp_tok = seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:]
    exp_sc = sc.numpy().reshape(Q, B)
    got_tok, got_sc = res.tokens.cpu().numpy(), res.scores.cpu().numpy()
    np.testing.assert_allclose(got_sc, exp_sc, atol=1e-4, rtol=0)
    # ranks whose score is separated from both neighbours by more than the fp32 reorder noise must agree exactly
    for q in range(Q):
        gap = np.minimum(np.abs(np.diff(exp_sc[q], prepend=np.inf)), np.abs(np.diff(exp_sc[q], append=-np.inf)))
        clear = gap > 5e-4   # 1000 beams are dense in score: most, not all, ranks are clearly separated
        assert clear.mean() > 0.5
        assert np.array_equal(got_tok[q][clear], exp_tok[q][clear])
    leaf = (res.row_hi > res.row_lo).cpu().numpy()
    assert leaf.all(), "with 6000 docs every one of the 1000 returned smtids must be a real doc"


def test_ff_intermediate_beyond_f16_range(setup):
    """Real T5 checkpoints are known to overflow fp16 in the FF intermediate relu(h Wi^T). The split-precision GEMM
    carries that tensor as f16 planes scaled by 2^-4 (exact), i.e. up to 1.05e6. A model whose pre-FF layer-norm
    weights are scaled by 256 and wi by 128 (FF activations ~1e5, and a residual stream ~1e4 that the following
    norms bring back) must still match the fp32 oracle."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q = 4, 256, 800, 4, 3
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    sd = dict(synth.make_state_dict(dims, seed=71))
    for k in list(sd):
        enc_ff_ln = k.startswith("encoder.block.") and k.endswith("layer.1.layer_norm.weight")
        dec_ff_ln = k.startswith("decoder.block.") and k.endswith("layer.2.layer_norm.weight")
        if k.endswith("DenseReluDense.wi.weight"):
            sd[k] = (sd[k] * 128.0).astype(np.float32)
        elif enc_ff_ln or dec_ff_ln:
            sd[k] = (sd[k] * 256.0).astype(np.float32)
    codes = synth.make_codes(N, L, V, seed=71)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=72, max_len=10)
    ref = t5_ref.T5RefCached(sd, dims)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    seqs, sc = beam_ref.beam_search_ref(ref, pm, ids, mask, B, L, use_kv_cache=True)
    # the premise of the test: the encoder's FF intermediate really leaves the f16 range
    h = torch.from_numpy(sd["shared.weight"])[torch.from_numpy(ids[0])]
    hn = t5_ref.rmsnorm(h, torch.from_numpy(sd["encoder.block.0.layer.1.layer_norm.weight"]), 1e-6)
    big = float((hn @ torch.from_numpy(sd["encoder.block.0.layer.1.DenseReluDense.wi.weight"]).T).abs().max())
    assert 65504.0 < big < 5e5, big
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    torch.cuda.synchronize()
    assert np.array_equal(res.tokens.cpu().numpy(), seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:])
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(Q, B), atol=1e-4, rtol=0)


@pytest.mark.parametrize("target", [1e4, 1e5, 1e6])
def test_heavy_tailed_residual_stream_needs_no_fp32_fallback(setup, target):
    """trained T5 checkpoints carry a few residual channels of 1e3 .. 1e5 (synth.make_state_dict(outliers=...)
    models them: |x| of 0.4-0.6 x target from the embedding to the last block). The residual planes are scaled by 2^-4 since
    round 5 (|x| < 1.05e6; round 4: unscaled, |x| < 65504 — every batch of such a model was repeated on the exact-fp32 path).
    The split-precision search must (1) not raise the saturation flag, (2) return the oracle's ranking (fp32 CPU) within the
    usual bars, (3) agree with the library's exact-fp32 mode."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q = 8, 256, 1000, 4, 5
    dims = synth.mini_dims(L=L, V=V, enc_layers=2, d_ff=128, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=91, outliers=target, logit_scale=3.0)
    codes = synth.make_codes(N, L, V, seed=91)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=92, max_len=12)
    ref = t5_ref.T5RefCached(sd, dims)
    enc = ref.encode(torch.from_numpy(ids).long(), torch.from_numpy(mask).long())
    x0 = torch.from_numpy(sd["shared.weight"])[torch.from_numpy(ids[0]).long()]
    assert 0.1 * target < float(x0.abs().max()) < 1.05e6, float(x0.abs().max())       # the premise: the stream really is that large
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    seqs, sc = beam_ref.beam_search_ref(ref, pm, ids, mask, B, L, use_kv_cache=True)
    exp_tok, exp_sc = seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:], sc.numpy().reshape(Q, B).astype(np.float64)
    model = E.DeviceModel(ctx, sd, dims)
    assert not model.f32_only
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    out = {}
    for prec in ("f16x2", "f32"):
        ctx.set_precision(prec)
        ctx.status(clear=True)
        res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
        torch.cuda.synchronize()
        flags = ctx.status(clear=True)
        assert not (flags & E._lib.STATUS_SATURATED), f"{prec}: an activation left the plane range at |x| ~ {target:g}"
        out[prec] = (res.tokens.cpu().numpy(), res.scores.cpu().numpy().astype(np.float64))
    ctx.set_precision("f16x2")
    for prec, (tok, scs) in out.items():
        for q in range(Q):
            got = {tuple(t): s for t, s in zip(tok[q].tolist(), scs[q])}
            want = {tuple(t): s for t, s in zip(exp_tok[q].tolist(), exp_sc[q])}
            assert got.keys() == want.keys(), (prec, q)
            for k in want:
                assert abs(got[k] - want[k]) <= 1e-4, (prec, q, got[k], want[k])
    assert np.abs(out["f16x2"][1] - out["f32"][1]).max() <= 5e-5
    print(f"[heavy-tail] target {target:g}: max |score(f16x2) - score(f32)| = {np.abs(out['f16x2'][1] - out['f32'][1]).max():.2e}, "
          f"vs oracle {max(np.abs(out['f16x2'][1] - exp_sc).max(), 0):.2e}")


def test_long_docids_beyond_the_register_attention_path(setup):
    """L = 40 positions: self-attention depths 33..36 run the largest register instantiation and 37..40 the generic
    LDS kernel (the reference's docids have 32 or 16 positions; the library accepts up to 64)."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q = 40, 256, 300, 3, 2
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=81)
    codes = synth.make_codes(N, L, V, seed=81)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=82, max_len=9)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    torch.cuda.synchronize()
    assert np.array_equal(res.tokens.cpu().numpy(), seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:])
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(Q, B), atol=1e-4, rtol=0)


@pytest.mark.parametrize("case", ["short", "long_queries", "longest_queries", "deep_docids", "beam1000"])
def test_128_dim_heads_against_the_kv_cached_oracle(setup, case):
    """d_kv = 128 (the head size of t5-3b, t5_generative_retriever.py:128-133) on a small stack: the generic attention kernels
    (enc_attn_kernel<128>, dec_attn_kernel<., 128>), the d_kv-strided KV cache map of the q/k/v GEMM and the packed encoder
    at ragged query lengths up to 120 tokens — and up to the reference's 256 (evaluate.py:465), where K and V of a head no
    longer fit the LDS together (enc_attn_kernel<128, true>: V rows from global memory); both GEMM modes; the plain loop and
    the forced tail (explicit forks)."""
    from oracle import beam_ref, t5_ref
    E, synth, ctx = setup["E"], setup["synth"], setup["ctx"]
    L, V, N, B, Q, qlen = {"short": (8, 256, 3000, 10, 5, 14), "long_queries": (6, 256, 3000, 4, 3, 120),
                           "longest_queries": (5, 256, 3000, 3, 3, 256),
                           "deep_docids": (40, 256, 300, 3, 2, 9), "beam1000": (4, 256, 60000, 1000, 1, 11)}[case]
    dims = synth.ModelDims(vocab_size=512, d_model=256, d_kv=128, d_ff=128, num_layers=2, num_decoder_layers=3, num_heads=4,
                           decoder_vocab_sizes=[V] * L)
    sd = synth.make_state_dict(dims, seed=91)
    codes = synth.make_codes(N, L, V, seed=91)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=92, max_len=qlen, mean_len=0.75 * qlen, std_len=0.2 * qlen)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)
    exp_tok, exp_sc = seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:], sc.numpy().reshape(Q, B)
    ctx.status(clear=True)
    # the plain step loop, and (round 6) the forced tail at this head size: tail_self_attn_kernel<128> + the chunked block
    # cross-attention; explicit fork depths so that every case really leaves the loop early
    modes = [("f16x2", None), ("f32", None), ("f16x2", [1]), ("f16x2", [2, 3] if L >= 5 else [2]), ("f32", [L - 1])]
    try:
        for precision, forks in modes:
            ctx.set_precision(precision)
            ctx.set_forced_tail(forks is not None)
            ctx.set_fork_depths(forks)
            res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
            torch.cuda.synchronize()
            if forks is not None:
                st = ctx.last_fork_stats()
                assert st and sum(f["forced"] + f["left"] for f in st[:1]) == Q, (case, forks, st)
            got_sc = res.scores.cpu().numpy()
            np.testing.assert_allclose(got_sc, exp_sc, atol=1e-4, rtol=0, err_msg=f"{case} {precision} forks {forks}")
            same = (res.tokens.cpu().numpy() == exp_tok).all(axis=2)
            # a rank may differ from the oracle's only where two candidates tie within the score tolerance
            assert bool((same | (np.abs(got_sc - exp_sc) <= 1e-4)).all()) and same.mean() >= 0.98, (case, precision, forks, same.mean())
    finally:
        ctx.set_precision("f16x2")
        ctx.set_fork_depths(None)
        ctx.set_forced_tail(True)
    assert ctx.status() == 0
    if case == "short":   # beyond the reference's truncation length: a message, not a launch error
        long_ids, long_mask = synth.make_queries(2, vocab_size=512, seed=93, fixed_len=257)
        with pytest.raises(E.RiporHipError, match="Lq out of range"):
            E.search(model, trie, torch.from_numpy(long_ids), torch.from_numpy(long_mask), B, L)


# ---- saturation guard of the split-precision planes -------------------------------------------------
def _sat_world(v_scale=1.0, weight_spike=None, seed=31):
    from ripor_amd.modeling.t5_generative_retriever import T5forDocIDConfig, T5ForDocIDGeneration
    from ripor_amd.tasks.generation import PrefixConstrainLogitProcessorFastSparse
    from ripor_amd.utils import synth
    L, V, B = 6, 256, 4
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=seed)
    # v.weight * 2^k and o.weight * 2^-k of one attention: the same function exactly (powers of two), but the attention
    # output of that layer is 2^k times larger
    a = "decoder.block.1.layer.0.SelfAttention"
    sd[a + ".v.weight"] = (sd[a + ".v.weight"] * v_scale).astype(np.float32)
    sd[a + ".o.weight"] = (sd[a + ".o.weight"] / v_scale).astype(np.float32)
    if weight_spike is not None:
        key, val = weight_spike
        sd[key] = sd[key].copy()
        sd[key][3, 5] = val
    codes = synth.make_codes(300, L, V, seed=seed)
    ids, mask = synth.make_queries(4, vocab_size=dims.vocab_size, seed=seed, max_len=12)
    model = T5ForDocIDGeneration(T5forDocIDConfig.from_dims(dims), sd).to(0)
    proc = PrefixConstrainLogitProcessorFastSparse.from_codes(codes, V)
    return dims, sd, codes, ids, mask, model, proc, B, L, V


def _oracle(dims, sd, codes, ids, mask, B, L, V):
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    return beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)


def test_activation_outside_the_f16_planes_is_flagged_and_recomputed_in_fp32():
    """An attention whose value projection is scaled by 2048 (and its output projection by 1/2048: the same model)
    produces attention outputs of several thousand, beyond the +-4094 range of the activation planes (65504 / 2^4): the
    raw engine call must raise the sticky saturation flag, and the reference-shaped entry point must return the
    exact-fp32 result (== oracle) with a warning instead of rankings computed from clipped tensors."""
    import warnings
    from ripor_amd import _lib, engine as E
    from ripor_amd.tasks.generation import generate_for_constrained_prefix_beam_search
    dims, sd, codes, ids, mask, model, proc, B, L, V = _sat_world(v_scale=2048.0)
    ctx = E.Context.get(0)
    assert ctx.get_precision() == "f16x2" and not model.engine_model().f32_only
    ctx.status(clear=True)
    E.search(model.engine_model(), proc.trie(0), torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    assert ctx.status(clear=True) & _lib.STATUS_SATURATED, "clamped activations were not reported"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = generate_for_constrained_prefix_beam_search(
            model, proc, input_ids=torch.from_numpy(ids).cuda(), attention_mask=torch.from_numpy(mask).cuda(),
            max_new_tokens=L, output_scores=True, return_dict_in_generate=True, num_beams=B, num_return_sequences=B)
    assert any("f16 plane range" in str(x.message) for x in w)
    assert ctx.get_precision() == "f16x2"                       # the retry does not leave the ctx in fp32 mode
    seqs, sc = _oracle(dims, sd, codes, ids, mask, B, L, V)
    assert (out.sequences.cpu().numpy() == seqs.numpy()).all()
    np.testing.assert_allclose(out.sequences_scores.cpu().numpy(), sc.numpy(), atol=1e-4, rtol=0)
    # sane inputs right after: no flag, no warning
    dims2, sd2, codes2, ids2, mask2, model2, proc2, *_ = _sat_world()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        generate_for_constrained_prefix_beam_search(model2, proc2, input_ids=torch.from_numpy(ids2).cuda(),
                                                    attention_mask=torch.from_numpy(mask2).cuda(), max_new_tokens=L,
                                                    num_beams=B, num_return_sequences=B)
    assert not [x for x in w if "f16 plane range" in str(x.message)]


def test_weight_outside_the_f16_planes_pins_the_model_to_fp32():
    """A weight of 300 (x 2^8 plane scale > 65504) cannot be carried by the weight planes: rpr_load_model pins the model
    to the exact-fp32 kernels, and the search still matches the oracle."""
    from ripor_amd import engine as E
    dims, sd, codes, ids, mask, model, proc, B, L, V = _sat_world(
        weight_spike=("decoder.block.2.layer.2.DenseReluDense.wi.weight", 300.0))
    em = model.engine_model()
    assert em.f32_only
    ctx = E.Context.get(0)
    ctx.status(clear=True)
    res = E.search(em, proc.trie(0), torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    torch.cuda.synchronize()
    assert ctx.status() == 0 and ctx.get_precision() == "f16x2"
    seqs, sc = _oracle(dims, sd, codes, ids, mask, B, L, V)
    assert (res.tokens.cpu().numpy() == seqs.numpy().reshape(len(ids), B, L + 1)[:, :, 1:]).all()
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(len(ids), B), atol=1e-4, rtol=0)


def test_query_without_attended_tokens_is_reported():
    """an all-zero attention-mask row has no packed encoder rows; the cross-attention used to read the next
    query's K/V. Now: defined output (zeros), sticky flag, ValueError from the reference-shaped entry point; the other
    queries of the batch are unaffected."""
    from ripor_amd import _lib, engine as E
    from ripor_amd.tasks.generation import generate_for_constrained_prefix_beam_search
    dims, sd, codes, ids, mask, model, proc, B, L, V = _sat_world()
    ctx = E.Context.get(0)
    ref = E.search(model.engine_model(), proc.trie(0), torch.from_numpy(ids), torch.from_numpy(mask), B, L)
    mask2 = mask.copy()
    mask2[1] = 0
    ctx.status(clear=True)
    res = E.search(model.engine_model(), proc.trie(0), torch.from_numpy(ids), torch.from_numpy(mask2), B, L)
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & _lib.STATUS_EMPTY_QUERY
    assert torch.isfinite(res.scores).all()
    keep = [0, 2, 3]
    assert torch.equal(res.tokens[keep], ref.tokens[keep]) and torch.allclose(res.scores[keep], ref.scores[keep], atol=1e-5)
    with pytest.raises(ValueError, match="all-zero attention_mask"):
        generate_for_constrained_prefix_beam_search(model, proc, input_ids=torch.from_numpy(ids).cuda(),
                                                    attention_mask=torch.from_numpy(mask2).cuda(), max_new_tokens=L,
                                                    num_beams=B, num_return_sequences=B)


def test_lane_split_gives_the_results_of_one_call(setup):
    """rpr_set_lane_split: a batch run as two halves on the two CU-masked lane streams (own workspaces, own hipGraphs)
    returns what the unsplit call returns, for an odd batch, in graph and eager mode, repeatedly (graph replay), and the
    caller's stream order holds (results are read right after the call on the same stream).
    Every query is computed on its own rows either way, but a half batch may take another GEMM route than the whole one
    (row thresholds of the wave-split tiles, K splits chosen from the tile count and the CUs of a lane) and the routes sum K
    in different orders: same smtids and row ranges, scores within fp32 summation-order noise (3e-5; the parity bar is
    1e-4) — for a forced split of 81 queries and for 2600 queries at the library's own threshold of 10 240 decoder rows.
    Replays and eager launches of the split call itself agree bit for bit."""
    E, ctx, dims, synth = setup["E"], setup["ctx"], setup["dims"], setup["synth"]
    L, V, B, Q = setup["L"], dims.decoder_vocab_sizes[0], 4, 81
    model = E.DeviceModel(ctx, synth.make_state_dict(dims, seed=31), dims)
    codes = synth.make_codes(3000, L, V, seed=77)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=78)
    ids, mask = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    saved = ctx.lane_split()

    def same(got, want, bits):
        assert torch.equal(got.tokens, want.tokens)
        assert torch.equal(got.row_lo, want.row_lo) and torch.equal(got.row_hi, want.row_hi)
        if bits:
            assert torch.equal(got.scores, want.scores)
        else:
            assert float((got.scores - want.scores).abs().max()) <= 3e-5   # two fp32 summation orders through 12 layers (the parity bar is 1e-4)

    try:
        ctx.set_lane_split(0)
        ref = E.search(model, trie, ids, mask, B, L)
        torch.cuda.synchronize()
        ctx.set_lane_split(2)
        if ctx.lane_split() == 0:
            pytest.skip("CU-masked streams unavailable on this device")
        first = None
        for use_graph in (True, True, False):
            got = E.search(model, trie, ids, mask, B, L, use_graph=use_graph)
            tok = got.tokens.clone()          # same stream: ordered after both lanes
            torch.cuda.synchronize()
            assert torch.equal(tok, ref.tokens)
            same(got, ref, bits=False)
            first = first or got
            assert torch.equal(got.scores, first.scores)      # replay and eager launches of the split call: the same bits
        # the other modes of the call: exact-fp32 GEMMs, log-softmax scores, a prefix shorter than the model's length
        for prec, kw, Lx in (("f32", {}, L), ("f16x2", {"apply_log_softmax_for_scores": True}, L), ("f16x2", {}, L - 2)):
            ctx.set_precision(prec)
            ctx.set_lane_split(0)
            want = E.search(model, trie, ids, mask, B, Lx, **kw)
            torch.cuda.synchronize()
            ctx.set_lane_split(2)
            got = E.search(model, trie, ids, mask, B, Lx, **kw)
            torch.cuda.synchronize()
            same(got, want, bits=False)
        ctx.set_precision("f16x2")
        # the library's own threshold: 2600 queries x 4 beams = 10 400 decoder rows
        ids2, mask2 = synth.make_queries(2600, vocab_size=dims.vocab_size, seed=79)
        ids2, mask2 = torch.from_numpy(ids2).cuda(), torch.from_numpy(mask2).cuda()
        ctx.set_lane_split(0)
        want = E.search(model, trie, ids2, mask2, B, L)
        torch.cuda.synchronize()
        ctx.set_lane_split(10240)
        got = E.search(model, trie, ids2, mask2, B, L)
        torch.cuda.synchronize()
        same(got, want, bits=False)
        again = E.search(model, trie, ids2, mask2, B, L, use_graph=False)
        torch.cuda.synchronize()
        same(again, got, bits=True)
    finally:
        ctx.set_lane_split(saved if saved else 10240)


_LEVELS_SCRIPT = r"""
import hashlib, json, sys, numpy as np, torch
sys.path.insert(0, %r)
from ripor_amd import engine as E
from ripor_amd.utils import synth
ctx = E.Context.get(0)
out = {}
for (N, V, L, B, zipf) in [(300_000, 256, 8, 10, 0.0), (120_000, 200, 6, 4, 1.0), (50_000, 1024, 4, 10, 0.0), (40_000, 2048, 4, 3, 0.0)]:
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    model = E.DeviceModel(ctx, synth.make_state_dict(dims, seed=5), dims)
    codes = synth.make_codes(N, L, V, seed=6, zipf=zipf) if zipf else synth.make_codes(N, L, V, seed=6)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ids, mask = synth.make_queries(7, vocab_size=512, seed=7, max_len=10)
    for forks in (None, [1]):
        ctx.set_fork_depths(forks)
        r = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for x in (r.tokens, r.scores, r.row_lo, r.row_hi):
            h.update(x.cpu().numpy().tobytes())
        out[f"{N}_{V}_{L}_{B}_{forks}"] = h.hexdigest()
        assert bool((r.row_hi > r.row_lo).all())
    ctx.set_fork_depths(None)
print("RESULT " + json.dumps(out))
"""


def test_trie_level_tables_change_nothing():
    """Round 5: the selection kernel reads the child ranges of steps 0 and 1 from the trie's level tables (rpr_trie::lvl0 / lvl1,
    built at upload) instead of binary-searching the code matrix. Same searches with RPR_SELECT_LEVELS = 1 / 0 in two processes:
    tokens, scores and row ranges identical bit for bit — uniform and Zipf codes, a vocab off the 64 grid, V = 1024 (level-1
    table of 1 M entries) and V = 2048 (no level-1 table), a stage that starts at step 1 (explicit fork at depth 1)."""
    import json
    import subprocess
    import sys
    got = {}
    for lv in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", _LEVELS_SCRIPT % REPO], env=dict(os.environ, RPR_SELECT_LEVELS=lv), capture_output=True,
                           text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        got[lv] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert got["1"] == got["0"] and len(got["1"]) == 8
