"""-m gpu: the forced-tail evaluation (rpr_search's forks + teacher-forced tail passes, ripor_amd/csrc/api.hip,
tail_kernels.hip) against the plain step-by-step loop and against the CPU oracle.

The step-by-step loop is what the reference does (tasks/generation.py:423-540, one model call + mask + top-k per
position) and is itself pinned to the reference's golden vectors (test_gpu_parity.py, which also runs every golden with
forks). Here: shapes the goldens do not reach — tries dense enough that most queries are forced by the automatic
depths while some are not, duplicated smtids (ranges of many rows holding one sequence), prefix search, skewed codes,
beam 1, the exact-fp32 mode, and the lane split.
"""
import numpy as np
import pytest
import torch

from conftest import ORDER_TOL, SCORE_TOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from ripor_amd import engine
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return engine


def _setup(E, codes, L_model, V, seed=3, Q=24):
    from ripor_amd.utils import synth
    dims = synth.mini_dims(L=L_model, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=seed)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=14)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    return ctx, model, trie, sd, dims, ids, mask


def _same_as_plain(ctx, E, model, trie, ids, mask, B, L, label):
    """forced-tail result == step-by-step result: identical sequences and row ranges at every rank outside score
    near-ties, scores within 0.3 of the parity tolerance (the step loop takes a logit from the split-precision GEMM, the tail from an exact fp32 dot product)."""
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_forced_tail(False)
    plain = E.search(model, trie, ti, tm, B, L)
    ctx.set_forced_tail(True)
    res = E.search(model, trie, ti, tm, B, L)
    torch.cuda.synchronize()
    stats = ctx.last_fork_stats()
    same = (res.tokens == plain.tokens).all(dim=2)
    close = (res.scores - plain.scores).abs() <= ORDER_TOL
    live = plain.scores > -1e6                   # dead beams (-1e9) carry arbitrary masked tokens in both paths
    assert bool((same | close | ~live).all()), f"{label}: sequences differ from the step-by-step loop"
    err = float(((res.scores - plain.scores).abs() * live).max())
    assert err <= 0.3 * SCORE_TOL, (label, err)
    both = same & live
    assert torch.equal(res.row_lo[both], plain.row_lo[both]) and torch.equal(res.row_hi[both], plain.row_hi[both])
    print(f"[forced tail] {label}: forks {stats}, max score diff {err:.2e}, {int(same.sum())}/{same.numel()} ranks identical")
    return res, plain, stats


def test_dense_trie_most_queries_forced_some_not(E):
    """60k docs under 256^2 = 65k depth-2 prefixes: the automatic first fork takes most queries, the rest walk on
    through the second stage; both populations must reproduce the plain loop, and the oracle."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    L, V, B, N = 12, 256, 10, 60_000      # automatic forks need >= 8 positions left after the first one
    codes = synth.make_codes(N, L, V, seed=11)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=48)
    depths = ctx.fork_depths(model, trie, ids.shape[0], B, L)
    assert len(depths) >= 1 and depths[0] in (2, 3, 4), depths
    # thousands of decoder rows in flight: a tail of fewer than 8 positions is not worth a fork; a few hundred rows (steps bound
    # by the launch chain) fork down to two remaining positions (round 6)
    assert ctx.fork_depths(model, trie, 1000, B, 8) == []
    assert ctx.fork_depths(model, trie, ids.shape[0], B, 8) != [] and ctx.fork_depths(model, trie, ids.shape[0], B, 8)[0] <= 6
    res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "dense 60k")
    assert stats[0]["forced"] + stats[0]["left"] == ids.shape[0]
    assert stats[0]["forced"] > 0, "the fork took no query: the test does not exercise the tail pass"
    # explicit early fork: a good share of the queries is NOT forced at depth 2 and takes the compacted stages
    ctx.set_fork_depths([2, 3])
    try:
        res2, _, st2 = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "dense 60k forks [2,3]")
        assert st2[0]["left"] > 0 and st2[1]["forced"] > 0, st2
    finally:
        ctx.set_fork_depths(None)
    # oracle on a few queries (the KV-cached CPU restatement of the reference loop)
    nq = 6
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids[:nq], mask[:nq], B, L, use_kv_cache=True)
    ref_tok = seqs.numpy().reshape(nq, B, L + 1)[:, :, 1:]
    ref_sc = sc.numpy().reshape(nq, B)
    got_tok, got_sc = res.tokens[:nq].cpu().numpy(), res.scores[:nq].cpu().numpy()
    near = np.zeros((nq, B), dtype=bool)
    near[:, 1:] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    near[:, :-1] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    assert ((got_tok == ref_tok).all(axis=2) | near).all(), "forced-tail sequences differ from the oracle"
    np.testing.assert_allclose(got_sc, ref_sc, atol=SCORE_TOL, rtol=0)


def test_duplicated_smtids_are_single_sequences(E):
    """Many docs per smtid (evaluate.py:439-446 appends every docid of an smtid): a beam's range holds several ROWS
    but one distinct sequence, so it is forced; the returned ranges must cover all duplicates."""
    from ripor_amd.utils import synth
    L, V, B = 6, 256, 8
    base = synth.make_codes(3000, L, V, seed=21)
    rep = 1 + (synth.randint("dup", (3000,), 0, 5, seed=21))
    codes = np.repeat(base, rep, axis=0)
    perm = synth.randint("dupperm", (codes.shape[0],), 0, 1 << 30, seed=21).argsort(kind="stable")
    codes = codes[perm]
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V)
    ctx.set_fork_depths([2, 3])
    try:
        res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "duplicated smtids")
    finally:
        ctx.set_fork_depths(None)
    assert stats and stats[0]["forced"] + stats[1]["forced"] > 0
    n = (res.row_hi - res.row_lo).cpu().numpy()
    assert n.max() > 1 and n.min() >= 1, "ranges of duplicated smtids must hold all their docs"


def test_prefix_search_shorter_than_the_trie(E):
    """max_new_token < trie depth (evaluate.py:134-178, the training-data generation callers): 'one distinct
    sequence' is judged on the first L columns only, the final ranges hold every doc sharing the L-prefix."""
    from ripor_amd.utils import synth
    Lc, L, V, B = 8, 3, 256, 10
    codes = synth.make_codes(30_000, Lc, V, seed=31)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, Lc, V)
    ctx.set_fork_depths([1])
    try:
        res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "prefix search L=3 of 8, fork [1]")
    finally:
        ctx.set_fork_depths(None)
    res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "prefix search L=3 of 8, automatic")
    L = 5
    ctx.set_fork_depths([2, 3])
    try:
        res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "prefix search L=5 of 8, forks [2, 3]")
    finally:
        ctx.set_fork_depths(None)
    assert stats and stats[0]["forced"] + stats[1]["forced"] > 0, stats


def test_skewed_codes_and_beam_one(E):
    from ripor_amd.utils import synth
    L, V = 8, 256
    codes = synth.make_codes(40_000, L, V, seed=41, skew=True)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=32)
    ctx.set_fork_depths([3, 4])
    try:
        for B in (1, 4, 10):
            _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"skewed 40k beam {B} forks [3, 4]")
    finally:
        ctx.set_fork_depths(None)
    ctx.set_fork_depths([2, 4])
    try:
        _, _, st = _same_as_plain(ctx, E, model, trie, ids, mask, 10, L, "skewed 40k forks [2,4]")
        assert st[0]["left"] > 0
    finally:
        ctx.set_fork_depths(None)


def test_exact_fp32_mode_and_lane_split(E):
    from ripor_amd.utils import synth
    L, V, B = 8, 256, 10
    codes = synth.make_codes(50_000, L, V, seed=51)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=40)
    ctx.set_fork_depths([3, 4])
    ctx.set_precision("f32")
    try:
        _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "exact fp32")
    finally:
        ctx.set_precision("f16x2")
    saved = ctx.lane_split()
    ctx.set_lane_split(64)           # 40 queries x 10 beams >= 64 rows: two lanes of 20 queries
    try:
        if ctx.lane_split() > 0:
            res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, "two lanes")
            assert sum(s["forced"] + s["left"] for s in stats[:1]) == ids.shape[0]
    finally:
        ctx.set_lane_split(saved)
        ctx.set_fork_depths(None)


def test_large_beam_forced_tail(E):
    """beam 100 and beam 300 (> 256 takes the sorted select path) on a trie with enough leaves."""
    from ripor_amd.utils import synth
    L, V = 6, 256
    codes = synth.make_codes(200_000, L, V, seed=61)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=6)
    ctx.set_fork_depths([3, 4])
    try:
        for B in (100, 300):
            _, _, st = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"200k docs beam {B} forks [3, 4]")
            assert st[0]["forced"] + st[1]["forced"] > 0
    finally:
        ctx.set_fork_depths(None)
    ctx.set_fork_depths([3])
    try:
        _, _, st = _same_as_plain(ctx, E, model, trie, ids, mask, 100, L, "200k docs beam 100 fork [3]")
        assert st[0]["forced"] > 0
    finally:
        ctx.set_fork_depths(None)


def test_log_softmax_forks_too_and_taps_never_fork(E):
    """apply_log_softmax_for_scores (generation.py:453-455): the tail pass computes the V exact-fp32 logits of every forced
    position (one GEMM per position) and adds log_softmax at the token — same sequences as the step-by-step loop, scores within
    0.3 of the tolerance; the oracle agrees. Debug taps still switch the forks off."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    L, V, B = 12, 256, 4
    codes = synth.make_codes(50_000, L, V, seed=71)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=8)
    assert ctx.fork_depths(model, trie, 8, B, L) != []
    assert ctx.fork_depths(model, trie, 8, B, L, log_softmax=True) != []
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_forced_tail(False)
    plain = E.search(model, trie, ti, tm, B, L, apply_log_softmax_for_scores=True)
    ctx.set_forced_tail(True)
    try:
        for depths in (None, [2], [1, 3], [3, 4]):
            ctx.set_fork_depths(depths)
            a = E.search(model, trie, ti, tm, B, L, apply_log_softmax_for_scores=True)
            torch.cuda.synchronize()
            st = ctx.last_fork_stats()
            assert st and (depths == [2] or sum(f["forced"] for f in st) > 0), (depths, st)   # depth 2 alone is too early here
            same = (a.tokens == plain.tokens).all(dim=2)
            close = (a.scores - plain.scores).abs() <= ORDER_TOL
            assert bool((same | close).all()), depths
            assert float((a.scores - plain.scores).abs().max()) <= 0.3 * SCORE_TOL, depths
            assert torch.equal(a.row_lo[same], plain.row_lo[same])
    finally:
        ctx.set_fork_depths(None)
    # the oracle (log-softmax scores) on two queries
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    a = E.search(model, trie, ti, tm, B, L, apply_log_softmax_for_scores=True)
    nq = 3
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids[:nq], mask[:nq], B, L,
                                        apply_log_softmax_for_scores=True, use_kv_cache=True)
    ref_tok, ref_sc = seqs.numpy().reshape(nq, B, L + 1)[:, :, 1:], sc.numpy().reshape(nq, B)
    near = np.zeros((nq, B), dtype=bool)
    near[:, 1:] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    near[:, :-1] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    assert ((a.tokens[:nq].cpu().numpy() == ref_tok).all(axis=2) | near).all(), "log-softmax forced tail differs from the oracle"
    np.testing.assert_allclose(a.scores[:nq].cpu().numpy(), ref_sc, atol=SCORE_TOL, rtol=0)
    b = E.search(model, trie, ti, tm, B, L, taps=True)
    torch.cuda.synchronize()
    assert ctx.last_fork_stats() == []
    c = E.search(model, trie, ti, tm, B, L)
    torch.cuda.synchronize()
    assert len(ctx.last_fork_stats()) >= 1
    assert torch.equal(b.tokens, c.tokens)


def test_optimistic_mode_flags_leftovers_and_the_guard_repeats(E):
    """Mode 2 does not enqueue the stage after the last fork; a query that is still unforced there raises
    STATUS_TAIL_LEFTOVER (its outputs are unspecified) and E.search_guarded repeats the batch in the exact mode."""
    from ripor_amd.utils import synth
    L, V, B = 8, 256, 10
    codes = synth.make_codes(60_000, L, V, seed=11)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=48)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_forced_tail(False)
    plain = E.search(model, trie, ti, tm, B, L)
    torch.cuda.synchronize()
    try:
        # forks [1, 2]: at depth 2 many queries of this trie are not forced yet -> leftovers
        ctx.set_fork_depths([1, 2])
        ctx.set_forced_tail(2)
        ctx.status(clear=True)
        E.search(model, trie, ti, tm, B, L)
        torch.cuda.synchronize()
        st = ctx.last_fork_stats()
        assert st[-1]["left"] > 0, st
        assert ctx.status(clear=True) & E._lib.STATUS_TAIL_LEFTOVER
        # the guard: optimistic first, exact repeat
        ctx.set_forced_tail(1)
        g = E.search_guarded(model, trie, ti, tm, B, L, optimistic=True)
        r = g.result()
        assert g.repeated and ctx.forced_tail() == 1
        same = (r.tokens == plain.tokens).all(dim=2) | ((r.scores - plain.scores).abs() <= ORDER_TOL)
        assert bool(same.all()) and float((r.scores - plain.scores).abs().max()) <= 0.3 * SCORE_TOL
        # automatic depths: the statistics promise an empty last stage -> no flag, no repeat, same results
        ctx.set_fork_depths(None)
        g = E.search_guarded(model, trie, ti, tm, B, L, optimistic=True)
        r = g.result()
        st = ctx.last_fork_stats()
        if len(st) == 2 and st[-1]["left"] == 0:
            assert not g.repeated
        same = (r.tokens == plain.tokens).all(dim=2) | ((r.scores - plain.scores).abs() <= ORDER_TOL)
        assert bool(same.all())
        assert ctx.status(clear=True) == 0
    finally:
        ctx.set_fork_depths(None)
        ctx.set_forced_tail(True)


def test_rank_replay_path_equals_single_ranking_pass(E, monkeypatch):
    """tail_rank_kernel ranks the summed scores once and replays the L - T selection steps only when two final scores
    are exactly equal; RPR_TAIL_RANK_REPLAY=1 forces the replay: both must give the same bits (beams 10 and 300; the
    1024-thread launch of large beams included)."""
    from ripor_amd.utils import synth
    L, V, N = 12, 256, 40_000
    codes = synth.make_codes(N, L, V, seed=5)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=6)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_forced_tail(True)
    for B in (10, 300):
        monkeypatch.setenv("RPR_TAIL_RANK_REPLAY", "0")
        a = E.search(model, trie, ti, tm, B, L)
        stats = ctx.last_fork_stats()
        assert stats and stats[0]["forced"] > 0, stats
        monkeypatch.setenv("RPR_TAIL_RANK_REPLAY", "1")
        b = E.search(model, trie, ti, tm, B, L)
        torch.cuda.synchronize()
        assert torch.equal(a.tokens, b.tokens) and torch.equal(a.scores, b.scores)
        assert torch.equal(a.row_lo, b.row_lo) and torch.equal(a.row_hi, b.row_hi)


def test_many_beams_few_queries_radix_selection_equals_the_single_block(E, monkeypatch):
    """Beam 512 / 1000 with 2 queries on a 300 000-doc trie: the radix selection (the default from 32 beams on,
    select_radix.hip) returns the bits of the single-block select_kernel (RPR_SELECT_RADIX=0), with and without the forced tail."""
    from ripor_amd.utils import synth
    L, V, N = 12, 256, 300_000
    codes = synth.make_codes(N, L, V, seed=9)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=2)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    try:
        for B in (512, 1000):
            for ft in (False, True):
                ctx.set_forced_tail(ft)
                monkeypatch.setenv("RPR_SELECT_RADIX", "0")
                ref = E.search(model, trie, ti, tm, B, L)
                monkeypatch.delenv("RPR_SELECT_RADIX")
                radix = E.search(model, trie, ti, tm, B, L)
                torch.cuda.synchronize()
                assert torch.equal(radix.tokens, ref.tokens) and torch.equal(radix.scores, ref.scores), (B, ft)
                assert torch.equal(radix.row_lo, ref.row_lo) and torch.equal(radix.row_hi, ref.row_hi), (B, ft)
    finally:
        ctx.set_forced_tail(True)


def test_exact_score_ties_resolve_identically_on_every_path(E, monkeypatch):
    """Exact float64 ties between candidates are decided by slot / candidate index (select_kernel: ascending flat index;
    finalize: reverse slot order). Output codebooks with pairwise identical rows (tokens 2k and 2k+1 score the same at every
    position: ties inside the steps) and, in a second model, all-zero codebooks from position 1 on (every candidate under the
    best first token ties, up to the final ranking) force those rules
    through every implementation of them: the step-by-step loop, the forced tail with its single ranking pass (which must
    detect the ties and replay the steps), the forced replay, and the radix selection — all must return the same bits."""
    from ripor_amd.utils import synth
    L, V, N, B = 12, 256, 40_000, 12
    codes = synth.make_codes(N, L, V, seed=21)
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    ids, mask = synth.make_queries(6, vocab_size=dims.vocab_size, seed=4, max_len=14)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx = E.Context.get(0)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    for variant in ("paired rows", "zero codebooks"):
        sd = synth.make_state_dict(dims, seed=6)
        for p in range(L):
            w = sd[f"list_output_embeds.{p}.weight"]
            if variant == "paired rows":
                w[1::2] = w[0::2]
            elif p >= 1:
                w[...] = 0.0
        model = E.DeviceModel(ctx, sd, dims)
        results = {}
        try:
            for name, ft, env in (("plain", False, {}), ("forced", True, {}), ("forced+replay", True, {"RPR_TAIL_RANK_REPLAY": "1"}),
                                  ("radix", False, {"RPR_SELECT_RADIX": "1"}), ("radix+forced", True, {"RPR_SELECT_RADIX": "1"})):
                for k in ("RPR_TAIL_RANK_REPLAY", "RPR_SELECT_RADIX"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                ctx.set_forced_tail(ft)
                r = E.search(model, trie, ti, tm, B, L)
                torch.cuda.synchronize()
                if ft:
                    st = ctx.last_fork_stats()
                    assert st and st[0]["forced"] > 0, (variant, name, st)
                results[name] = r
        finally:
            ctx.set_forced_tail(True)
        ref = results["plain"]
        live = ref.scores > -1e6
        tied = int(((ref.scores[:, 1:] == ref.scores[:, :-1]) & live[:, 1:]).sum())
        if variant == "zero codebooks":
            assert tied > 0, "no exact ties among the returned scores — the test does not exercise the final tie rule"
        # The step loop takes its logits from the split-precision GEMM, the tail pass from an exact fp32 dot product: with
        # non-zero codebooks the two families agree to ~1e-5, not to the bit, so bits are compared inside a family; with the
        # zero codebooks every tail logit is exactly 0 on both sides and all five paths must agree bit for bit.
        families = ([("plain", "radix"), ("forced", "forced+replay", "radix+forced")] if variant == "paired rows"
                    else [tuple(results)])
        for fam in families:
            base = results[fam[0]]
            for name in fam[1:]:
                r = results[name]
                assert torch.equal(r.tokens[live], base.tokens[live]), (variant, name)
                assert torch.equal(r.scores[live], base.scores[live]), (variant, name)
                assert torch.equal(r.row_lo[live], base.row_lo[live]) and torch.equal(r.row_hi[live], base.row_hi[live]), (variant, name)
        assert float((results["forced"].scores - ref.scores).abs().max()) <= 0.3 * SCORE_TOL
        print(f"[ties] {variant}: {tied} exactly tied neighbours among {int(live.sum())} returned beams; "
              f"families compared bit for bit: {families}")


def test_zipf_codes_of_survey_8d(E):
    """SURVEY.md §8(d)'s skewed trie: Zipf s = 1.0 on the first three levels (the imbalance of residual-quantiser codes,
    reference aq_preprocess/create_customized_smtid_file.py:33-59). Popular prefixes are dense, so the automatic forks
    come later than on uniform codes and a share of the queries walks on after the first one."""
    from ripor_amd.utils import synth
    L, V, B, N = 16, 256, 10, 400_000
    codes = synth.make_codes(N, L, V, seed=71, zipf=1.0)
    uni = synth.make_codes(N, L, V, seed=71)
    f_z, f_u = E.trie_single_frac(codes, L), E.trie_single_frac(uni, L)
    # the popular prefixes stay dense for longer (the many rare prefixes are single-sequence nodes early on, which is why
    # depth 2 goes the other way): the depth at which nearly every node holds one sequence moves from 3 to 4-5
    assert f_z[3] < f_u[3] and f_z[4] < f_u[4], (f_z[:6], f_u[:6])
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=64)
    d_z = ctx.fork_depths(model, trie, 64, B, L)
    trie_u = E.DeviceTrie.from_codes(ctx, uni, V)
    d_u = ctx.fork_depths(model, trie_u, 64, B, L)
    assert d_z and d_u and d_z[0] >= d_u[0], (d_z, d_u)
    res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"zipf 400k, forks {d_z} (uniform: {d_u})")
    assert stats[0]["forced"] > 0
    for mode in (2,):   # optimistic mode with its guard
        ctx.set_forced_tail(mode)
        try:
            g = E.search_guarded(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L).result()
        finally:
            ctx.set_forced_tail(True)
        assert torch.equal(g.tokens, res.tokens)


def test_late_first_fork(E):
    """A trie whose first automatic fork is >= 8: nine levels over a 3-symbol alphabet (19 683 prefixes holding ~6 docs
    each), uniform codes below. Ten sequential steps with partly dead beams (3 and 9 live candidates at depths 1 and 2),
    then the tail pass over the remaining 14 positions; also at an explicit later pair of forks."""
    from ripor_amd.utils import synth
    L, V, B, N = 24, 256, 6, 120_000
    codes = synth.make_codes(N, L, V, seed=81)
    codes[:, :9] = synth.randint("late_fork", (N, 9), 0, 3, seed=81).astype(codes.dtype)
    f = E.trie_single_frac(codes, L)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=40)
    depths = ctx.fork_depths(model, trie, 40, B, L)
    assert depths and depths[0] >= 8, (depths, f[:14])
    res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"late fork {depths}")
    assert stats[0]["forced"] > 0, stats
    ctx.set_fork_depths([depths[0] + 1, depths[0] + 3])
    try:
        _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"late fork, explicit {[depths[0] + 1, depths[0] + 3]}")
    finally:
        ctx.set_fork_depths(None)


def test_optimistic_mode_backs_off_after_repeated_leftovers(E):
    """search_guarded bets on the optimistic forced tail; on a trie where the last fork keeps leaving queries behind (a dense
    60k-doc trie with early explicit forks) the bet is lost every time and each batch would be searched twice. After two
    lost bets in a row the ctx runs the exact mode for the next OPTIMISTIC_BACKOFF calls: same results, no repeat."""
    from ripor_amd.utils import synth
    L, V, B, N = 12, 256, 10, 60_000
    codes = synth.make_codes(N, L, V, seed=11)
    ctx, model, trie, sd, dims, ids, mask = _setup(E, codes, L, V, Q=48)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_forced_tail(True)
    exact = E.search(model, trie, ti, tm, B, L)
    ctx._leftover_streak, ctx._exact_calls_left = 0, 0
    ctx.set_fork_depths([1, 2])          # depth 2 of a 60k-doc trie: most beams still have several leaves below them
    try:
        ref = E.search(model, trie, ti, tm, B, L)
        runs, res = [], []
        for _ in range(4):              # (evaluate.py consults a guard one batch later; here: before the next call)
            runs.append(E.search_guarded(model, trie, ti, tm, B, L))
            res.append(runs[-1].result())
        assert [r.repeated for r in runs] == [True, True, False, False], [r.repeated for r in runs]
        assert ctx._exact_calls_left == E.OPTIMISTIC_BACKOFF - 2
        for r in res:
            assert torch.equal(r.tokens, ref.tokens) and torch.equal(r.scores, ref.scores)
        assert ctx.forced_tail() == 1, "the ctx mode must be what it was"
    finally:
        ctx.set_fork_depths(None)
        ctx._leftover_streak, ctx._exact_calls_left = 0, 0
    same = (exact.tokens == ref.tokens).all(dim=2) | ((exact.scores - ref.scores).abs() <= ORDER_TOL)
    assert bool(same.all())


@pytest.mark.parametrize("lens", [(20, 60), (70, 150)])
def test_tail_pass_with_long_queries(E, lens):
    """Cross-attention of the tail pass over long queries: 33-64 encoder keys take the two-key-tile MFMA kernel, more than 64
    the block kernel of the sequential steps (tail_kernels.hip::launch_tail_cross_attn). Forced tail vs the plain loop, and
    the CPU oracle on two queries."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    L, V, B, N = 10, 256, 6, 20_000
    codes = synth.make_codes(N, L, V, seed=91)
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=9)
    ids, mask = synth.make_queries(10, vocab_size=dims.vocab_size, seed=9, min_len=lens[0], max_len=lens[1],
                                   mean_len=(lens[0] + lens[1]) / 2, std_len=(lens[1] - lens[0]) / 4)
    assert lens[0] <= int(mask.sum(1).min()) and ids.shape[1] > (64 if lens[0] >= 70 else 32)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ctx.set_fork_depths([2, 3])
    try:
        res, plain, stats = _same_as_plain(ctx, E, model, trie, ids, mask, B, L, f"long queries {lens}, forks [2, 3]")
    finally:
        ctx.set_fork_depths(None)
    assert stats and stats[0]["forced"] + stats[1]["forced"] > 0, stats
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids[:2], mask[:2], B, L, use_kv_cache=True)
    ref_tok, ref_sc = seqs.numpy().reshape(2, B, L + 1)[:, :, 1:], sc.numpy().reshape(2, B)
    got_tok, got_sc = res.tokens[:2].cpu().numpy(), res.scores[:2].cpu().numpy()
    near = np.zeros((2, B), dtype=bool)
    near[:, 1:] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    near[:, :-1] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
    assert ((got_tok == ref_tok).all(axis=2) | near).all()
    np.testing.assert_allclose(got_sc, ref_sc, atol=SCORE_TOL, rtol=0)
