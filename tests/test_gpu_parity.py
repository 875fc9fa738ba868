"""-m gpu: parity of the HIP path (through the C ABI) against the reference's golden vectors and
against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): ranked smtid sequences bit-exact, beam scores within 1e-4.
fp32 summation order differs between any two implementations, so what "bit-exact" can mean is fixed by the
reference's own numbers (tests/conftest.py::compare_ranked): the SET of returned sequences must be the reference's
unless the reference itself dropped a candidate by < 1e-3 at some step (no committed fixture does: every pruning
margin is > 2e-3, tests/test_oracle_golden.py::test_fixture_margins_are_comfortable); every sequence's score must
match within 1e-4; the tokens at a rank must be identical whenever the reference's score at that rank is more than
2e-4 away from both neighbours. The counts of checked / near-tie ranks are printed per fixture.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ORDER_TOL, compare_ranked, golden_names

SCORE_TOL = 1e-4     # north_star: beam scores within 1e-4
LOGIT_TOL = 5e-4     # fp32 logits O(10..100) through 12-24 layers; reference-vs-KV-cached differs by ~3e-5, the
                     # split-precision GEMMs by <= ~1e-4 from exact fp32 (tools/precision_probe.py)

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.parametrize("name", golden_names())
def test_search_matches_reference_golden(engine, golden_cache, name):
    """Includes BASELINE config 4 (g3_large_b100_l32: t5-large decoder dims, beam 100, len 32, four queries) and
    config 1 (g4_base_b1_l32_q64: t5-base dims, 1k-doc trie, beam 1, 64 queries)."""
    g = golden_cache(name)
    ctx, model, trie = _build(engine, g)
    ctx.status(clear=True)
    res = _run(engine, g, model, trie)
    assert ctx.status() == 0, "saturation / empty-query flag raised on a golden fixture"
    tokens = res.tokens.cpu().numpy()
    scores = res.scores.cpu().numpy()
    stats = compare_ranked(g, tokens, scores)
    assert stats["boundary_excused"] == 0 and stats["sequences_missing"] == 0
    assert stats["ranks_checked"] >= 0.9 * g.Q * g.B or "tiny_trie" in name, stats
    # eager (no graph) launches give the same bits as the graph replay
    res2 = _run(engine, g, model, trie, use_graph=False)
    assert torch.equal(res.tokens, res2.tokens) and torch.equal(res.scores, res2.scores)
    # second replay of the cached graph is deterministic
    res3 = _run(engine, g, model, trie)
    assert torch.equal(res.tokens, res3.tokens) and torch.equal(res.scores, res3.scores)
    # exact-fp32 GEMM mode: same bar
    ctx.set_precision("f32")
    try:
        res4 = _run(engine, g, model, trie)
        compare_ranked(g, res4.tokens.cpu().numpy(), res4.scores.cpu().numpy(), label=" (exact fp32)")
    finally:
        ctx.set_precision("f16x2")
    # Forced-tail evaluation. The runs above used the fork depths chosen from the trie (printed); here the plain
    # step-by-step loop and explicit forks (first step, two adjacent forks, the last possible depth) face the same bar,
    # and every variant must return the sequences of the plain loop with scores within 0.3 of the tolerance.
    auto = ctx.fork_depths(model, trie, g.Q, g.B, g.L, g.log_softmax)
    print(f"[forced tail] {name}: automatic fork depths {auto}")
    try:
        ctx.set_forced_tail(False)
        plain = _run(engine, g, model, trie)
        compare_ranked(g, plain.tokens.cpu().numpy(), plain.scores.cpu().numpy(), label=" (no forced tail)")
        ctx.set_forced_tail(True)
        for depths in ([1], [2, 3], [g.L - 1], [1, g.L - 1]):
            if any(t >= g.L for t in depths) or len(set(depths)) != len(depths):
                continue
            ctx.set_fork_depths(depths)
            r = _run(engine, g, model, trie)
            compare_ranked(g, r.tokens.cpu().numpy(), r.scores.cpu().numpy(), label=f" (forks {depths})")
            same = (r.tokens == plain.tokens).all(dim=2)          # [Q, B]
            close = (r.scores - plain.scores).abs() <= ORDER_TOL
            assert bool((same | close).all()), f"{name} forks {depths}: sequences differ from the step-by-step loop outside near-ties"
            assert float((r.scores - plain.scores).abs().max()) <= 0.3 * SCORE_TOL
            assert torch.equal(r.row_lo[same], plain.row_lo[same]) and torch.equal(r.row_hi[same], plain.row_hi[same])
    finally:
        ctx.set_fork_depths(None)
        ctx.set_forced_tail(True)


def _unpack_valid(words, B, V):
    """[Q, B*V/64] int64 (uint64 bit patterns) -> bool [Q, B, V]."""
    w = words.astype(np.uint64)
    bits = ((w[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).astype(bool)
    return bits.reshape(w.shape[0], B, V)


def _check_select_taps(g, res, pm, label=""):
    """The selection kernel against the reference's per-step record and the reference processor's masks:
      * phase A (trie child mask): the bitmap select_kernel worked from at step t must equal the processor's mask of
        the beams' own prefixes (reconstructed from the tapped tokens/parents — independent of the oracle's path);
      * phases B-D (float64 combine, top-B, beam expand): wherever the reference's sorted top-(B+1) of a step has no
        gap below ORDER_TOL, the (parent, token) of every new slot must equal the reference's candidate of that rank
        and the cumulative float64 score must match to 1e-3 (fp32 logits summed over <= L steps)."""
    Q, B, L, V = g.Q, g.B, g.L, g.V
    tok = res.taps["step_tokens"].cpu().numpy()      # [L, Q, B]
    par = res.taps["step_parent"].cpu().numpy()
    ssc = res.taps["step_scores"].cpu().numpy()
    val = res.taps["step_valid"].cpu().numpy()       # [L, Q, B*V/64]
    ts, ti = g.z["top_scores"], g.z["top_idx"]
    prefixes = np.zeros((Q, B, 1), dtype=np.int64)   # column 0 = start id
    strict_steps = 0
    follow = np.ones(Q, dtype=bool)                  # slot order still equals the reference's
    for t in range(L):
        # ---- phase A
        got = _unpack_valid(val[t], B, V)
        exp = pm(prefixes.reshape(Q * B, t + 1)).reshape(Q, B, V) > 0
        if t == 0:   # beams 1..B-1 start dead (-1e9) on the same root prefix: the mask is the root's for all
            assert (exp == exp[:, :1]).all()
        assert (got == exp).all(), f"{g.name}{label} step {t}: select_kernel child bitmap differs from the processor mask"
        # ---- phases B-D
        K = ts.shape[2]
        for q in range(Q):
            if not follow[q]:
                continue
            live = ts[t, q] > -1e8
            gaps = ts[t, q, :-1] - ts[t, q, 1:]
            gaps = np.where(live[:-1] & live[1:], gaps, np.inf)
            nb = min(B, K)
            if gaps[:nb].min() > ORDER_TOL and live[:nb].all():
                rb, rt = ti[t, q, :nb] // V, ti[t, q, :nb] % V
                assert (par[t, q, :nb] == rb).all() and (tok[t, q, :nb] == rt).all(), (
                    f"{g.name}{label} step {t} query {q}: selected (parent, token) differ from the reference")
                np.testing.assert_allclose(ssc[t, q, :nb], ts[t, q, :nb], atol=1e-3, rtol=0)
                strict_steps += 1
            else:
                follow[q] = False   # a near-tie (or dead candidates, whose order is a tie rule) may permute slots
        prefixes = np.concatenate([np.take_along_axis(prefixes, par[t][:, :, None].astype(np.int64), axis=1),
                                   tok[t][:, :, None].astype(np.int64)], axis=2)
    print(f"[select] {g.name}{label}: {strict_steps} of {L * Q} (step, query) selections checked slot by slot")
    return strict_steps


# (debug taps expose [R, V] rows and V / 64 bitmap words per beam: the fixtures whose vocab is not a multiple of 64 — the
# library pads the token axis for them — are covered by the ranked comparison above, in every mode and with every fork)
@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith("g4_") and "_v100" not in n and "_v200" not in n])
def test_select_kernel_mask_and_choices_match_reference(engine, golden_cache, name):
    from oracle import beam_ref
    g = golden_cache(name)
    ctx, model, trie = _build(engine, g)
    res = _run(engine, g, model, trie, taps=True)
    d2s = {str(i): [-1] + [int(x) for x in row] for i, row in enumerate(g.codes)}
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), g.V)
    strict = _check_select_taps(g, res, pm)
    assert strict >= (g.L * g.Q) // 2 or "tiny_trie" in name or g.B >= 100, (strict, g.L * g.Q)


def test_select_mask_at_the_narrow_wide_crossover(engine):
    """select_kernel enumerates the children of ranges of <= 32 rows and binary-searches wider ones; V = 1024 takes
    the 16-word bitmaps. A constructed trie puts beams on ranges of exactly 1, 31, 32, 33 and 34 rows at the same
    step; bitmaps vs the oracle processor, results vs the oracle search."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    E = engine
    L, V, B = 4, 1024, 6
    sizes = {3: 32, 500: 33, 1000: 31, 7: 34, 900: 1, 64: 40}      # first token -> docs below it
    rows = []
    for first, n in sizes.items():
        sub = synth.randint(f"cross/{first}", (n, L - 1), 0, V, seed=5)
        sub[:, 0] = (np.arange(n) * 29 + first) % V                  # distinct second tokens: every child is one doc
        for r in sub:
            rows.append([first] + [int(x) for x in r])
    codes = np.asarray(rows, dtype=np.uint16)
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=9)
    ids, mask = synth.make_queries(5, vocab_size=dims.vocab_size, seed=9, max_len=14)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L, taps=True)
    torch.cuda.synchronize()
    d2s = {str(i): [-1] + [int(x) for x in row] for i, row in enumerate(codes)}
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), V)
    tok = res.taps["step_tokens"].cpu().numpy(); par = res.taps["step_parent"].cpu().numpy()
    val = res.taps["step_valid"].cpu().numpy()
    Q = ids.shape[0]
    prefixes = np.zeros((Q, B, 1), dtype=np.int64)
    range_sizes = set()
    for t in range(L):
        got = _unpack_valid(val[t], B, V)
        exp = pm(prefixes.reshape(Q * B, t + 1)).reshape(Q, B, V) > 0
        assert (got == exp).all(), f"step {t}"
        prefixes = np.concatenate([np.take_along_axis(prefixes, par[t][:, :, None].astype(np.int64), axis=1),
                                   tok[t][:, :, None].astype(np.int64)], axis=2)
        if t == 0:
            range_sizes = {sizes[int(x)] for x in tok[0].reshape(-1)}
    assert range_sizes == set(sizes.values()), "every constructed range size must be on a beam at step 1"
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, use_kv_cache=True)
    assert (res.tokens.cpu().numpy() == seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:]).all()
    np.testing.assert_allclose(res.scores.cpu().numpy(), sc.numpy().reshape(Q, B), atol=SCORE_TOL, rtol=0)
    lo, hi = res.row_lo.cpu().numpy(), res.row_hi.cpu().numpy()
    assert (hi - lo == 1).all()


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
    # step 0 unconditionally; step t > 0 whenever the beams are still in the reference's slot order, i.e. every selection
    # so far matched the reference's (a near-tie may permute slots: the rows of later steps then belong to other beams)
    lg = res.taps["step_logits"].cpu().numpy()
    checked = 0
    in_order = True
    for t in range(g.L):
        if t > 0:
            sel_tok = res.taps["step_tokens"][t - 1].cpu().numpy().reshape(-1)
            sel_par = res.taps["step_parent"][t - 1].cpu().numpy().reshape(-1)
            ref_tok = rec["steps"][t - 1]["top_tok"][:, : g.B].reshape(-1)
            ref_par = rec["steps"][t - 1]["top_beam"][:, : g.B].reshape(-1) if "top_beam" in rec["steps"][t - 1] else sel_par
            in_order = in_order and bool((sel_tok == ref_tok).all()) and bool((sel_par == ref_par).all())
        if not in_order:
            break
        ref = rec["steps"][t]["logits"]
        live = np.isfinite(ref) & (np.abs(ref) < 1e8)
        np.testing.assert_allclose(lg[t][live], ref[live], atol=LOGIT_TOL, rtol=1e-5)
        checked += 1
    assert checked >= 1 and (checked >= g.L // 2 or "tiny_trie" in name), (name, checked)
    print(f"[logits] {name}: {checked} of {g.L} steps compared with the oracle's logits at {LOGIT_TOL}")


def test_trie_mask_matches_reference_processor(engine, golden_cache):
    """rpr_trie_mask (the stand-alone processor entry, prefix_mask_kernel). The mask the search itself uses is
    checked by test_select_kernel_mask_and_choices_match_reference above."""
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
                                    (640, 256, 768, False, False), (33, 96, 64, True, True),
                                    # t5-large shapes (d = 1024, d_ff = 4096): 128-row, skinny and 256x256 kernels
                                    (300, 1024, 1024, False, True), (1000, 4096, 1024, True, False),
                                    (640, 1024, 4096, False, True), (12800, 3072, 1024, False, False),
                                    (12800, 1024, 4096, False, True), (12800, 4096, 1024, True, False),
                                    # a little more than a whole number of rounds of 256x256 tiles (270 / 792 / 258 tiles)
                                    (23040, 768, 768, False, True), (22500, 2304, 768, False, False), (22016, 768, 3072, True, True)]:
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


@pytest.mark.parametrize("M,N,K", [(20480, 768, 768), (5120, 2304, 768), (640, 768, 3072), (10, 768, 768), (2050, 832, 768),
                                   (12800, 1024, 4096), (12800, 4096, 1024), (23040, 768, 768), (22016, 768, 3072)])
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


_WSPLIT_SCRIPT = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
from ripor_amd import engine as E
ctx = E.Context.get(0)
out = {}
for (M, N, K, relu, resid) in [(40, 768, 768, False, True), (280, 2304, 768, False, False), (333, 256, 768, True, False),
                               (640, 768, 3072, False, True), (70, 96, 64, True, True), (129, 1024, 1024, False, True), (700, 3072, 768, True, False)]:
    torch.manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda") if resid else None
    first = ctx.linear(A, W, R, relu)
    for _ in range(3):
        assert torch.equal(ctx.linear(A, W, R, relu), first), "not repeatable"
    ref = A.double() @ W.double().t()
    ref = torch.relu(ref) if relu else ref
    ref = ref + R.double() if resid else ref
    out[f"{M}x{N}x{K}"] = {"err": (first.double() - ref).abs().max().item(), "scale": max(1.0, ref.abs().max().item()),
                           "sha": hashlib.sha256(first.cpu().numpy().tobytes()).hexdigest()}
print("RESULT " + json.dumps(out))
"""


def _wsplit_run(env):
    import json
    import subprocess
    import sys
    e = dict(os.environ, **{"RPR_DEV_LIB": "1", **env})   # development switches: live only in libripor_hip_dev.so
    p = subprocess.run([sys.executable, "-c", _WSPLIT_SCRIPT % REPO], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_wave_split_gemm_tiles_agree_with_each_other_and_with_fp64():
    """Round 5: gemm_h2_wsplit_kernel (64 x 32 and 64 x 64 tiles next to the 32 x 32 skinny tile, optional K split over blocks
    with the fused reduction launch) for 33 .. 768 rows. Forced through RPR_WSPLIT_CFG / RPR_WSPLIT_KS in subprocesses (the
    route is read once per process): every shape within the split-precision bar of the fp64 product and repeatable; without a
    K split over blocks the three tile shapes sum every output in the same order and must agree BIT FOR BIT (ragged M / N,
    residual, ReLU included); the automatic choice must be one of them."""
    base = _wsplit_run({"RPR_WSPLIT_CFG": "0", "RPR_WSPLIT_KS": "1"})
    for cfg in ("1", "2"):
        got = _wsplit_run({"RPR_WSPLIT_CFG": cfg, "RPR_WSPLIT_KS": "1"})
        for k, v in got.items():
            assert v["err"] < 2e-5 * v["scale"], (cfg, k, v)
            assert v["sha"] == base[k]["sha"], f"tile shape {cfg} differs from the 32 x 32 tile on {k}"
    for cfg, ks in (("0", "2"), ("1", "2"), ("2", "4")):
        got = _wsplit_run({"RPR_WSPLIT_CFG": cfg, "RPR_WSPLIT_KS": ks})
        for k, v in got.items():
            assert v["err"] < 2e-5 * v["scale"], (cfg, ks, k, v)
    auto = _wsplit_run({})
    assert auto == _wsplit_run({"RPR_DEV_LIB": "0"}), "the product library differs from the development build on its default route"
    old = _wsplit_run({"RPR_GEMM_WSPLIT_MAX": "0"})
    for k, v in auto.items():
        assert v["err"] < 2e-5 * v["scale"] and old[k]["err"] < 2e-5 * old[k]["scale"], (k, v, old[k])


_SUPERTILE_SCRIPT = r"""
import hashlib, json, sys, torch
sys.path.insert(0, %r)
from ripor_amd import engine as E
ctx = E.Context.get(0)
out = {}
# more tiles than CUs and more than four column tiles: the persistent 256 x 256 kernel in super-tile order; a ragged last row
# panel and a ragged last column tile, bands that do not divide the row panels, a narrower last column group
for (M, N, K, relu, resid) in [(24576, 3072, 768, True, False), (20000 + 37, 2304, 768, False, False), (16384, 1280 + 64, 768, False, True),
                               (23040, 4096, 256, False, False), (17000, 768, 768, False, True)]:
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * K ** -0.5
    R = torch.randn(M, N, device="cuda") if resid else None
    first = ctx.linear(A, W, R, relu)
    assert torch.equal(ctx.linear(A, W, R, relu), first), "not repeatable"
    ref = A.double() @ W.double().t()
    ref = torch.relu(ref) if relu else ref
    ref = ref + R.double() if resid else ref
    out[f"{M}x{N}x{K}"] = {"err": (first.double() - ref).abs().max().item(), "scale": max(1.0, ref.abs().max().item()),
                           "sha": hashlib.sha256(first.cpu().numpy().tobytes()).hexdigest()}
print("RESULT " + json.dumps(out))
"""


def test_super_tile_order_of_the_persistent_gemm_changes_no_bit():
    """Round 6: gemm_h2_pp_kernel walks products wider than four column tiles band by band, column group by column group
    (GemmH2Args::tile_cw / tile_rb). Same tiles, same arithmetic per tile: the result must be the row-major walk's bit for bit
    (RPR_PP_SUPERTILE=0, development build), every output within the split-precision bar of the fp64 product, also with column
    groups of 2 and 5 tiles; the product library takes the development build's default."""
    import json
    import subprocess
    import sys

    def run(env):
        e = dict(os.environ, **{"RPR_DEV_LIB": "1", **env})
        p = subprocess.run([sys.executable, "-c", _SUPERTILE_SCRIPT % REPO], env=e, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        return json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])

    base = run({"RPR_PP_SUPERTILE": "0"})
    for k, v in base.items():
        assert v["err"] < 2e-5 * v["scale"], (k, v)
    for sup in ("1", "2", "5"):
        got = run({"RPR_PP_SUPERTILE": sup})
        assert {k: v["sha"] for k, v in got.items()} == {k: v["sha"] for k, v in base.items()}, f"super-tile order {sup} changes the result"
    assert run({"RPR_DEV_LIB": "0"}) == run({})


def test_vocab_sizes_off_the_64_grid(engine):
    """Decoder vocab sizes that are not multiples of 64 (rpr_load_model pads every output codebook with zero rows up to the
    next multiple of 64; the selection never picks a padding column; log_softmax runs over the real columns): V = 100, 65,
    200 against the CPU oracle — raw and log-softmax scores, split-precision and exact fp32, forced tail on and off, a
    beam count large enough that every real candidate of the first steps is needed."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    E = engine
    ctx = E.Context.get(0)
    for V, B, L, N, lsm in ((100, 4, 8, 2000, False), (65, 10, 6, 500, True), (200, 32, 6, 3000, False)):
        codes = synth.make_codes(N, L, V, seed=V)
        dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
        sd = synth.make_state_dict(dims, seed=V + 1)
        ids, mask = synth.make_queries(4, vocab_size=dims.vocab_size, seed=V, max_len=12)
        model = E.DeviceModel(ctx, sd, dims)
        trie = E.DeviceTrie.from_codes(ctx, codes, V)
        pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
        seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, apply_log_softmax_for_scores=lsm,
                                            use_kv_cache=True)
        ref_tok, ref_sc = seqs.numpy().reshape(4, B, L + 1)[:, :, 1:], sc.numpy().reshape(4, B)
        live = ref_sc > -1e6
        near = np.zeros((4, B), dtype=bool)
        near[:, 1:] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
        near[:, :-1] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
        try:
            for prec in ("f16x2", "f32"):
                ctx.set_precision(prec)
                for ft in (False, True):
                    ctx.set_forced_tail(ft)
                    ctx.set_fork_depths([2] if ft else None)
                    r = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L, apply_log_softmax_for_scores=lsm)
                    torch.cuda.synchronize()
                    got_tok, got_sc = r.tokens.cpu().numpy(), r.scores.cpu().numpy()
                    assert (got_tok[live] < V).all(), (V, prec, ft, "a padding column was selected")
                    assert ((got_tok == ref_tok).all(axis=2) | near | ~live).all(), (V, prec, ft)
                    assert np.abs((got_sc - ref_sc) * live).max() <= SCORE_TOL, (V, prec, ft)
        finally:
            ctx.set_precision("f16x2"); ctx.set_forced_tail(True); ctx.set_fork_depths(None)
        with pytest.raises(Exception):
            E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L, taps=True)   # taps need V % 64 == 0
