"""-m gpu: the BASELINE-size configuration (t5-base dims, 8 841 823-doc trie, beam 10, len 32) checked
through size-independent properties — the oracle cannot run at this size:
  * every returned smtid is a leaf of the trie: its sorted-row range is non-empty and the code rows
    under it equal the returned tokens (mask/trie walk/beam expand are consistent over 32 levels);
  * ranked best-first, scores finite;
  * batch invariance: a query searched alone returns exactly what it returns inside a batch
    (no cross-query leakage through the KV cache, ancestry tables or GEMM tiling);
  * graph replay is bit-deterministic;
  * the split-precision GEMM path agrees with the exact-fp32 MFMA path: identical smtids wherever the
    fp32 run's neighbouring scores are >1e-3 apart, scores within 1e-4.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

N_DOCS, L, V, B = 8_841_823, 32, 256, 10


@pytest.fixture(scope="module")
def world():
    from ripor_amd import engine as E
    from ripor_amd.utils import synth
    ctx = E.Context.get(0)
    dims = synth.t5_base_dims(L=L, V=V)
    model = E.DeviceModel(ctx, synth.make_state_dict(dims), dims)
    codes = synth.make_codes_fast(N_DOCS, L, V)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ids, mask = synth.make_queries(48, vocab_size=dims.vocab_size, seed=11)
    yield E, ctx, model, trie, codes, torch.from_numpy(ids), torch.from_numpy(mask)
    ctx.set_precision("f16x2")


def test_fullsize_properties(world):
    E, ctx, model, trie, codes, ids, mask = world
    ctx.set_precision("f16x2")
    res = E.search(model, trie, ids, mask, B, L)
    torch.cuda.synchronize()
    tok = res.tokens.cpu().numpy()
    sc = res.scores.cpu().numpy()
    lo, hi = res.row_lo.cpu().numpy(), res.row_hi.cpu().numpy()
    assert np.isfinite(sc).all() and (sc > -1e6).all()
    assert (np.diff(sc, axis=1) <= 0).all(), "beams are not ranked best-first"
    assert (hi > lo).all(), "a returned smtid is not a trie leaf"
    for q in range(tok.shape[0]):
        for b in range(B):
            rows = trie.perm[lo[q, b]:hi[q, b]]
            assert (codes[rows] == tok[q, b][None, :]).all()
        assert len({tuple(t) for t in tok[q]}) == B, "duplicate smtid among the beams of a query"
    # determinism of the cached graph
    res2 = E.search(model, trie, ids, mask, B, L)
    assert torch.equal(res.tokens, res2.tokens) and torch.equal(res.scores, res2.scores)
    # batch invariance (alone, padded to its own length bucket or to the batch's: same answer)
    for q in (0, 7, 31):
        n = int(mask[q].sum())
        for width in (n, ids.shape[1]):
            r1 = E.search(model, trie, ids[q:q + 1, :width], mask[q:q + 1, :width], B, L)
            assert np.array_equal(r1.tokens.cpu().numpy()[0], tok[q]), f"query {q} differs when searched alone"
            np.testing.assert_allclose(r1.scores.cpu().numpy()[0], sc[q], atol=1e-4, rtol=0)


def test_split_precision_agrees_with_exact_fp32(world):
    E, ctx, model, trie, codes, ids, mask = world
    ctx.set_precision("f32")
    ref = E.search(model, trie, ids, mask, B, L, taps=True)
    torch.cuda.synchronize()
    ctx.set_precision("f16x2")
    res = E.search(model, trie, ids, mask, B, L)
    torch.cuda.synchronize()
    rt, rs = ref.tokens.cpu().numpy(), ref.scores.cpu().numpy()
    tt, ts = res.tokens.cpu().numpy(), res.scores.cpu().numpy()
    step_scores = ref.taps["step_scores"].cpu().numpy()  # [L, Q, B] float64, sorted desc per step
    excused = 0
    for q in range(rt.shape[0]):
        if np.array_equal(rt[q], tt[q]):
            np.testing.assert_allclose(ts[q], rs[q], atol=1e-4, rtol=0)
        else:
            gaps = -np.diff(step_scores[:, q, :], axis=1)
            assert gaps.min() < 1e-3, f"query {q}: smtids differ although the fp32 run has no near-tie"
            excused += 1
    assert excused <= rt.shape[0] // 8, excused
