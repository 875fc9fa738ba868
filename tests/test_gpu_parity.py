"""-m gpu: parity of the HIP path (through the C ABI) against the reference's golden vectors and
against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): ranked smtid sequences bit-exact, beam scores within 1e-4.
fp32 summation order differs between any two implementations, so an integer mismatch is accepted
only for a query whose reference margin (gap between neighbouring candidates inside the top-(B+1)
at some step) is below MARGIN_TOL, i.e. where the reference itself is one rounding away from a
different answer; such queries are counted and must stay rare.
"""
import numpy as np
import pytest
import torch

from conftest import golden_names

SCORE_TOL = 1e-4     # north_star: beam scores within 1e-4
LOGIT_TOL = 2e-3     # fp32 logits O(10..100) through 12-24 layers; reference-vs-KV-cached differs by ~3e-5
MARGIN_TOL = 1e-3

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from ripor_amd import engine as E
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return E


def _build(E, g):
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, g.state_dict, g.dims)
    trie = E.DeviceTrie.from_codes(ctx, g.codes, g.V)
    return ctx, model, trie


def _run(E, g, model, trie, **kw):
    ids = torch.from_numpy(g.input_ids)
    mask = torch.from_numpy(g.attention_mask)
    res = E.search(model, trie, ids, mask, g.B, g.L, apply_log_softmax_for_scores=g.log_softmax, **kw)
    torch.cuda.synchronize()
    return res


def _compare_to_golden(g, tokens, scores):
    Q, B, L = g.Q, g.B, g.L
    exp_tok = g.sequences.reshape(Q, B, L + 1)
    assert (exp_tok[:, :, 0] == 0).all()
    exp_tok = exp_tok[:, :, 1:]
    exp_sc = g.sequences_scores.reshape(Q, B)
    margins = g.step_margins()
    excused = 0
    for q in range(Q):
        same = (tokens[q] == exp_tok[q]).all()
        if same:
            np.testing.assert_allclose(scores[q], exp_sc[q], atol=SCORE_TOL, rtol=0,
                                       err_msg=f"{g.name} query {q}: beam scores differ")
        else:
            assert margins is not None and margins[q] < MARGIN_TOL, (
                f"{g.name} query {q}: smtid sequences differ from the reference although its margin "
                f"is {None if margins is None else margins[q]}")
            excused += 1
    return excused


@pytest.mark.parametrize("name", golden_names())
def test_search_matches_reference_golden(engine, golden_cache, name):
    g = golden_cache(name)
    ctx, model, trie = _build(engine, g)
    res = _run(engine, g, model, trie)
    tokens = res.tokens.cpu().numpy()
    scores = res.scores.cpu().numpy()
    excused = _compare_to_golden(g, tokens, scores)
    assert excused <= max(1, g.Q // 4), f"{excused} of {g.Q} queries needed the near-tie excuse"
    # eager (no graph) launches give the same bits as the graph replay
    res2 = _run(engine, g, model, trie, use_graph=False)
    assert torch.equal(res.tokens, res2.tokens) and torch.equal(res.scores, res2.scores)
    # second replay of the cached graph is deterministic
    res3 = _run(engine, g, model, trie)
    assert torch.equal(res.tokens, res3.tokens) and torch.equal(res.scores, res3.scores)


@pytest.mark.parametrize("name", ["g1_mini_b4_l8", "g1_mini_b10_l8_tiny_trie", "g1_mini_b2_l4_v1024"])
def test_encoder_and_step_logits_match_oracle(engine, golden_cache, name):
    from oracle import beam_ref, t5_ref
    g = golden_cache(name)
    ctx, model, trie = _build(engine, g)
    res = _run(engine, g, model, trie, taps=True)
    enc = res.taps["encoder_out"].cpu().numpy()
    np.testing.assert_allclose(enc, g.z["encoder_out"], atol=2e-4, rtol=1e-4)
    # oracle with per-step records
    d2s = {str(i): [-1] + [int(x) for x in row] for i, row in enumerate(g.codes)}
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), g.V)
    rec = {}
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5Ref(g.state_dict, g.dims), pm, g.input_ids, g.attention_mask,
                                        g.B, g.L, g.log_softmax, record=rec)
    assert (seqs.numpy() == g.sequences).all()  # the oracle itself is pinned to the reference
    tok = res.tokens.cpu().numpy()
    if (tok == g.sequences.reshape(g.Q, g.B, g.L + 1)[:, :, 1:]).all():
        lg = res.taps["step_logits"].cpu().numpy()
        for t in range(g.L):
            ref = rec["steps"][t]["logits"]
            if t == 0:
                np.testing.assert_allclose(lg[t], ref, atol=LOGIT_TOL, rtol=1e-4)
            else:
                # beams of step t are in the same slot order iff the selections matched so far
                sel_tok = res.taps["step_tokens"][t - 1].cpu().numpy().reshape(-1)
                ref_tok = rec["steps"][t - 1]["top_tok"][:, : g.B].reshape(-1)
                if (sel_tok == ref_tok).all():
                    np.testing.assert_allclose(lg[t], ref, atol=LOGIT_TOL, rtol=1e-4)


def test_trie_mask_matches_reference_processor(engine, golden_cache):
    for name in golden_names():
        g = golden_cache(name)
        ctx = engine.Context.get(0)
        trie = engine.DeviceTrie.from_codes(ctx, g.codes, g.V)
        for key in g.z.files:
            if not key.startswith("pm_prefix_T"):
                continue
            T = int(key[len("pm_prefix_T"):])
            prefix = g.z[key]
            expect = np.unpackbits(g.z[f"pm_mask_T{T}"], axis=1)[:, : g.V]
            got = trie.mask(prefix)
            assert (got == expect).all(), f"{name} T={T}"


def test_linear_kernel_against_torch_fp32(engine):
    ctx = engine.Context.get(0)
    torch.manual_seed(0)
    for (M, N, K, relu, resid) in [(1, 256, 768, False, False), (80, 2304, 768, False, False),
                                    (130, 768, 3072, False, True), (257, 3072, 768, True, False),
                                    (640, 256, 768, False, False), (33, 96, 64, True, True)]:
        A = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda") * K ** -0.5
        R = torch.randn(M, N, device="cuda") if resid else None
        out = ctx.linear(A, W, R, relu)
        ref = A.double() @ W.double().t()
        if relu:
            ref = torch.relu(ref)
        if resid:
            ref = ref + R.double()
        err = (out.double() - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (M, N, K, err)


def test_rmsnorm_kernel_against_torch_fp32(engine):
    ctx = engine.Context.get(0)
    torch.manual_seed(1)
    for rows, d in [(1, 768), (77, 768), (640, 1024)]:
        x = torch.randn(rows, d, device="cuda") * 3
        w = torch.rand(d, device="cuda") + 0.5
        out = ctx.rmsnorm(x, w, 1e-6)
        ref = w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6))
        torch.testing.assert_close(out, ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(20480, 768, 768), (5120, 2304, 768), (640, 768, 3072), (10, 768, 768), (2050, 832, 768)])
def test_gemm_kernels_are_repeatable_bitwise(engine, M, N, K):
    """Race screen of the three split-precision GEMM kernels (ping-pong 256x256, LDS-DMA 128-row with deep prefetch,
    skinny): LDS-DMA ordering bugs show up as rare timing-dependent wrong tiles, so the same launch is repeated and
    must reproduce its first result bit for bit (tools/gemm_race_screen.py runs the long version)."""
    ctx = engine.Context.get(0)
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda")
    first = ctx.linear(A, W, R)
    ref = A.double() @ W.double().t() + R.double()
    assert (first.double() - ref).abs().max().item() < 5e-5
    for _ in range(8):
        assert torch.equal(ctx.linear(A, W, R), first)
