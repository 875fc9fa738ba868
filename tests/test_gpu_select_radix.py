"""-m gpu: the radix selection (ripor_amd/csrc/select_radix.hip) — the top-B of a step for many beams per query, the
reference script's --topk=1000 (full_evaluate_t5seq_aq_encoder.sh:191-199; tasks/generation.py:453-503) — against the
reference goldens, the CPU oracle and the single-block select_kernel, whose bits it must reproduce.

The library takes the radix path from 32 beams on; RPR_SELECT_RADIX=1 puts every selection on it (the goldens have
2 .. 100 beams), RPR_SELECT_RADIX=0 none.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import compare_ranked, golden_names

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def E():
    from ripor_amd import engine
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return engine


def _equal(a, b):
    return (torch.equal(a.tokens, b.tokens) and torch.equal(a.scores, b.scores) and torch.equal(a.row_lo, b.row_lo)
            and torch.equal(a.row_hi, b.row_hi))


@pytest.mark.parametrize("name", golden_names())
def test_radix_selection_on_the_reference_goldens(E, golden_cache, name, monkeypatch):
    """Every golden with every selection on the radix path: the reference's ranking (same bar as the default path) and the
    bits of the single-block kernel — forced tail on and off, hipGraph replay and eager launches."""
    from test_gpu_parity import _build, _run
    g = golden_cache(name)
    ctx, model, trie = _build(E, g)
    try:
        for ft in (False, True):
            ctx.set_forced_tail(ft)
            monkeypatch.setenv("RPR_SELECT_RADIX", "0")
            ref = _run(E, g, model, trie)
            monkeypatch.setenv("RPR_SELECT_RADIX", "1")
            got = _run(E, g, model, trie)
            compare_ranked(g, got.tokens.cpu().numpy(), got.scores.cpu().numpy(), label=f" (radix selection, forced tail {ft})")
            assert _equal(got, ref), f"{name}: radix selection differs from select_kernel (forced tail {ft})"
            eager = _run(E, g, model, trie, use_graph=False)
            assert _equal(eager, ref)
    finally:
        ctx.set_forced_tail(True)


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith("g4_") and "_v100" not in n and "_v200" not in n])
def test_radix_selection_taps_match_the_reference_steps(E, golden_cache, name, monkeypatch):
    """The child bitmap of rs_mask_kernel against the reference processor's mask and the (parent, token, score) of every
    new slot against the reference's per-step top-(B+1), step by step (the check of select_kernel in test_gpu_parity.py)."""
    from oracle import beam_ref
    from test_gpu_parity import _build, _check_select_taps, _run
    g = golden_cache(name)
    ctx, model, trie = _build(E, g)
    monkeypatch.setenv("RPR_SELECT_RADIX", "1")
    res = _run(E, g, model, trie, taps=True)
    d2s = {str(i): [-1] + [int(x) for x in row] for i, row in enumerate(g.codes)}
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), g.V)
    strict = _check_select_taps(g, res, pm, label=" (radix)")
    assert strict >= (g.L * g.Q) // 2 or "tiny_trie" in name or g.B >= 100, (strict, g.L * g.Q)


def test_radix_selection_many_beams_against_the_oracle(E):
    """Beam 300 and 1000 on a 6000-doc trie (more beams than children at the first levels: masked candidates are selected
    and die out later), raw and log-softmax scores, against the KV-cached CPU oracle; the default path IS the radix one."""
    from oracle import beam_ref, t5_ref
    from ripor_amd.utils import synth
    L, V, N, Q = 3, 256, 6000, 2
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    sd = synth.make_state_dict(dims, seed=61)
    codes = synth.make_codes(N, L, V, seed=61)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
    ids, mask = synth.make_queries(Q, vocab_size=512, seed=62, max_len=9)
    for B, lsm in ((300, False), (1000, True)):
        seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, apply_log_softmax_for_scores=lsm,
                                            use_kv_cache=True)
        res = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L, apply_log_softmax_for_scores=lsm)
        torch.cuda.synchronize()
        exp_tok, exp_sc = seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:], sc.numpy().reshape(Q, B)
        got_tok, got_sc = res.tokens.cpu().numpy(), res.scores.cpu().numpy()
        live = exp_sc > -1e6
        np.testing.assert_allclose(got_sc[live], exp_sc[live], atol=1e-4, rtol=0)
        for q in range(Q):
            gap = np.minimum(np.abs(np.diff(exp_sc[q], prepend=np.inf)), np.abs(np.diff(exp_sc[q], append=-np.inf)))
            clear = (gap > 5e-4) & live[q]
            assert clear.mean() > 0.3
            assert np.array_equal(got_tok[q][clear], exp_tok[q][clear]), (B, lsm, q)


_DEEP_SCRIPT = r"""
import hashlib, json, sys, numpy as np, torch
sys.path.insert(0, %r)
from ripor_amd import engine as E
from ripor_amd.utils import synth
ctx = E.Context.get(0)
out = {}
# (N, V, L, alphabet of the first positions, beams): few distinct codes at the first positions make trie nodes of thousands
# of rows down to depth 4 (CSR levels 2 .. 4; the levels behind level 2 are entered through the 64-ary search); V = 2048 has
# no level-1 table (two binary searches per token at step 1); V = 200 is off the 64 grid
for (N, V, L, alpha, B) in [(300_000, 256, 8, 8, 300), (300_000, 256, 8, 8, 16), (400_000, 64, 6, 64, 400), (40_000, 2048, 4, 2048, 260),
                            (120_000, 200, 6, 200, 256)]:
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, vocab_size=512)
    model = E.DeviceModel(ctx, synth.make_state_dict(dims, seed=5), dims)
    codes = synth.make_codes(N, L, V, seed=6)
    if alpha < V:
        codes[:, :5] %%= alpha
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ids, mask = synth.make_queries(3, vocab_size=512, seed=7, max_len=10)
    for ft in (False, True):
        ctx.set_forced_tail(ft)
        r = E.search(model, trie, torch.from_numpy(ids), torch.from_numpy(mask), B, L)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for x in (r.tokens, r.scores, r.row_lo, r.row_hi):
            h.update(x.cpu().numpy().tobytes())
        out[f"{N}_{V}_{L}_{alpha}_{B}_{ft}"] = h.hexdigest()
        assert bool((r.row_hi > r.row_lo).all())
ctx.set_forced_tail(True)
print("RESULT " + json.dumps(out))
"""


def test_radix_selection_reads_the_child_arrays_of_every_level():
    """Nodes of more than 64 rows at depths 2 .. 4 (CSR child arrays, trie.h ChildLevels), a vocab without a level-1 table and
    one off the 64 grid: the radix selection with the child arrays, the radix selection without them (RPR_SELECT_LEVELS=0:
    binary searches) and the single-block kernel return the same bits."""
    got = {}
    for tag, env in (("radix", {"RPR_SELECT_RADIX": "1"}), ("radix, no child arrays", {"RPR_SELECT_RADIX": "1", "RPR_SELECT_LEVELS": "0"}),
                     ("single block", {"RPR_SELECT_RADIX": "0"})):
        p = subprocess.run([sys.executable, "-c", _DEEP_SCRIPT % REPO], env=dict(os.environ, **env), capture_output=True, text=True,
                           timeout=1200)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
        got[tag] = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert len(got["radix"]) == 10
    assert got["radix"] == got["single block"], "radix selection differs from select_kernel"
    assert got["radix, no child arrays"] == got["single block"]


def test_radix_selection_when_whole_codebooks_tie(E, monkeypatch):
    """All-zero output codebooks: every logit is 0, so at step 0 the 256 candidates of beam 0 tie and so do the (B - 1) * 256
    candidates of the dead beams, and every later step is one big tie: far more ties on the threshold than the finish kernel's
    sort holds, which takes its own radix select on (key, candidate index) first. Ties are decided by ascending candidate
    index — the bits of select_kernel — for few and many beams, with and without the forced tail."""
    from ripor_amd.utils import synth
    L, V, N = 10, 256, 40_000
    codes = synth.make_codes(N, L, V, seed=21)
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=6)
    for p in range(L):
        sd[f"list_output_embeds.{p}.weight"][...] = 0.0
    ids, mask = synth.make_queries(3, vocab_size=dims.vocab_size, seed=4, max_len=14)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx = E.Context.get(0)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    model = E.DeviceModel(ctx, sd, dims)
    try:
        for B in (12, 300, 1000):
            for ft in (False, True):
                ctx.set_forced_tail(ft)
                monkeypatch.setenv("RPR_SELECT_RADIX", "0")
                ref = E.search(model, trie, ti, tm, B, L)
                monkeypatch.setenv("RPR_SELECT_RADIX", "1")
                got = E.search(model, trie, ti, tm, B, L)
                torch.cuda.synchronize()
                live = ref.scores > -1e6
                assert torch.equal(got.tokens[live], ref.tokens[live]) and torch.equal(got.scores, ref.scores), (B, ft)
                assert torch.equal(got.row_lo[live], ref.row_lo[live]) and torch.equal(got.row_hi[live], ref.row_hi[live]), (B, ft)
    finally:
        ctx.set_forced_tail(True)


def test_radix_selection_in_compacted_stages(E, monkeypatch):
    """Explicit forks at depths 2 and 3 on a dense trie: most queries leave at the first fork, the others step on in a
    compacted stage whose live query count only the device knows — the radix kernels' blocks past it must exit and the
    histograms of the queries that left must stay untouched. Same bits as the single-block kernel."""
    from ripor_amd.utils import synth
    L, V, N, B = 12, 256, 60_000, 256
    codes = synth.make_codes(N, L, V, seed=11)
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128)
    sd = synth.make_state_dict(dims, seed=3)
    ids, mask = synth.make_queries(6, vocab_size=dims.vocab_size, seed=3, max_len=14)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    ctx = E.Context.get(0)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    ctx.set_forced_tail(True)
    ctx.set_fork_depths([2, 3])
    try:
        monkeypatch.setenv("RPR_SELECT_RADIX", "0")
        ref = E.search(model, trie, ti, tm, B, L)
        st_ref = ctx.last_fork_stats()
        monkeypatch.setenv("RPR_SELECT_RADIX", "1")
        got = E.search(model, trie, ti, tm, B, L)
        torch.cuda.synchronize()
        st = ctx.last_fork_stats()
        assert st == st_ref and st[0]["left"] > 0, (st, st_ref)
        assert _equal(got, ref)
    finally:
        ctx.set_fork_depths(None)
