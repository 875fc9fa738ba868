#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (diagnostic; `python tools/fuzz_parity.py [n_cases] [seed]`): mini-dims models over
random tries (sizes 300 .. 300 k docs, uniform / skewed codes, duplicated smtids), lengths 4..16, beams 1..1000 (a few cases
with many beams and 1-2 queries, a few with enough rows for the lane split), V 256 / 1024 / 100 / 200,
raw-logit and log-softmax scores. Every case compares

  * the forced tail (automatic depths, then random explicit forks, exact and optimistic mode) with the step-by-step loop:
    same sequences and row ranges outside score near-ties, scores within 0.3e-4;
  * the radix selection (forced onto every step, RPR_SELECT_RADIX=1) with the single-block selection: identical bits;
  * small cases with the CPU oracle (KV-cached restatement of the reference loop): ranked comparison at 1e-4.

Prints one line per case and a summary; exits non-zero on the first mismatch."""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ripor_amd import engine as E
from ripor_amd.utils import synth
from oracle import beam_ref, t5_ref

SCORE_TOL, ORDER_TOL = 1e-4, 2e-4
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
ctx = E.Context.get(0)


BOUNDARY_TOL = 1e-3     # as tests/conftest.py::compare_ranked: a candidate the reference itself dropped / kept by less than this
CASE = {}               # inputs of the running case (for the pruning-margin check below)
n_excused = 0


def pruning_margin(q):
    """Smallest gap, over the steps, between the B-th and (B+1)-th candidate of query q in the CPU oracle's loop: two
    implementations whose logits differ in the last bits may legitimately prune differently below BOUNDARY_TOL."""
    c = CASE
    rec = {}
    beam_ref.beam_search_ref(t5_ref.T5RefCached(c["sd"], c["dims"]), c["pm"](), c["ids"][q:q + 1], c["mask"][q:q + 1], c["B"], c["L"],
                             apply_log_softmax_for_scores=c["lsm"], use_kv_cache=True, record=rec)
    gaps = []
    for st in rec["steps"]:
        ts = st["top_scores"][0]
        if ts.shape[0] > c["B"] and ts[c["B"]] > -1e8:
            gaps.append(float(ts[c["B"] - 1] - ts[c["B"]]))
    return min(gaps) if gaps else float("inf")


def same_as(a, b, label, bits=False):
    live = b.scores > -1e6
    if bits:
        ok = torch.equal(a.tokens[live], b.tokens[live]) and torch.equal(a.scores[live], b.scores[live]) and \
            torch.equal(a.row_lo[live], b.row_lo[live]) and torch.equal(a.row_hi[live], b.row_hi[live])
        assert ok, f"{label}: not bit-identical"
        return
    same = (a.tokens == b.tokens).all(dim=2)
    close = (a.scores - b.scores).abs() <= ORDER_TOL
    global n_excused
    bad_q = sorted(set((~(same | close | ~live)).nonzero()[:, 0].tolist()))
    for q in bad_q:          # a different pruning decision at a boundary the reference itself decides by < BOUNDARY_TOL?
        m = pruning_margin(q)
        if m < BOUNDARY_TOL:
            print(f"  {label}: query {q} differs, excused: the oracle's closest pruning margin is {m:.2e}")
            live[q] = False      # the query is left out of the comparison
            n_excused += 1
    if not bool((same | close | ~live).all()):
        bad = (~(same | close | ~live)).nonzero()
        q = int(bad[0, 0])
        print(f"{label}: query {q}, ranks {bad[bad[:, 0] == q][:, 1].tolist()}")
        print("  scores a:", [round(float(x), 6) for x in a.scores[q]])
        print("  scores b:", [round(float(x), 6) for x in b.scores[q]])
        sa = {tuple(t.tolist()) for t in a.tokens[q]}; sb = {tuple(t.tolist()) for t in b.tokens[q]}
        print(f"  sequences only in a: {len(sa - sb)}, only in b: {len(sb - sa)}")
    assert bool((same | close | ~live).all()), f"{label}: sequences differ outside near-ties"
    err = float(((a.scores - b.scores).abs() * live).max())
    assert err <= 0.3 * SCORE_TOL, (label, err)
    both = same & live
    assert torch.equal(a.row_lo[both], b.row_lo[both]) and torch.equal(a.row_hi[both], b.row_hi[both]), label


only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None     # re-run one case of a sweep (same seed)
params = []
for case in range(n_cases):                 # every random draw up front, so that a single case can be replayed
    N = rng.choice([300, 3000, 20_000, 60_000, 300_000])
    L = rng.choice([4, 6, 8, 10, 12, 16])
    V = rng.choice([256, 256, 256, 1024, 100, 200])
    B = rng.choice([1, 2, 4, 10, 10, 32, 100])
    Q = rng.randint(1, 24)
    skew, dup, lsm = rng.random() < 0.4, rng.random() < 0.3, rng.random() < 0.3
    seed = rng.randint(1, 10_000)
    shape = rng.random()
    if shape < 0.08:        # few queries x many beams: 1024-thread tail ranking
        B, Q, N, V = rng.choice([256, 500, 1000]), rng.randint(1, 2), rng.choice([60_000, 300_000]), 256
    elif shape < 0.12:      # >= 10 240 decoder rows: the call runs as two lanes
        B, Q = 10, rng.randint(1030, 1300)
    d0 = rng.randint(1, max(1, L - 2))
    depths = [d0] + ([rng.randint(d0 + 1, L - 1)] if rng.random() < 0.6 and d0 + 1 <= L - 1 else [])
    params.append((N, L, V, B, Q, skew, dup, lsm, seed, depths, rng.random() < 0.1))   # last: 128-dim heads (6 of them)

n_oracle = 0
for case, (N, L, V, B, Q, skew, dup, lsm, seed, depths, wide) in enumerate(params):
    if only is not None and case != only:
        continue
    codes = synth.make_codes(N, L, V, seed=seed, skew=skew)
    if dup:
        k = max(1, N // 8)
        codes[-k:] = codes[:k]
    dims = synth.mini_dims(L=L, V=V, enc_layers=1, d_ff=128, **(dict(d_kv=128, num_heads=6) if wide else {}))
    sd = synth.make_state_dict(dims, seed=seed)
    ids, mask = synth.make_queries(Q, vocab_size=dims.vocab_size, seed=seed, max_len=14)
    ti, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    model = E.DeviceModel(ctx, sd, dims)
    trie = E.DeviceTrie.from_codes(ctx, codes, V)
    kw = dict(apply_log_softmax_for_scores=lsm)
    CASE.update(sd=sd, dims=dims, ids=ids, mask=mask, B=B, L=L, lsm=lsm,
                pm=lambda codes=codes, V=V: beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V))
    os.environ["RPR_SELECT_RADIX"] = "0"          # the single-block selection first; the radix selection is compared with it below
    ctx.set_fork_depths(None)
    ctx.set_forced_tail(0)
    plain = E.search(model, trie, ti, tm, B, L, **kw)
    notes = []
    for mode in (1, 2):
        ctx.set_forced_tail(mode)
        forced = E.search_guarded(model, trie, ti, tm, B, L, **kw).result()
        same_as(forced, plain, f"case {case} forced mode {mode}")
    notes.append(f"auto forks {ctx.fork_depths(model, trie, Q, B, L, lsm)}")
    ctx.set_forced_tail(1)
    if L >= 4:
        ctx.set_fork_depths(depths)
        forced = E.search(model, trie, ti, tm, B, L, **kw)
        same_as(forced, plain, f"case {case} forks {depths}")
        notes.append(f"forks {depths} -> {[(f['forced'], f['left']) for f in ctx.last_fork_stats()]}")
        ctx.set_fork_depths(None)
    if True:                                      # every selection on the radix path (select_radix.hip): the single block's bits
        os.environ["RPR_SELECT_RADIX"] = "1"
        ctx.set_forced_tail(0)
        radix = E.search(model, trie, ti, tm, B, L, **kw)
        same_as(radix, plain, f"case {case} radix selection", bits=True)
        notes.append("radix == single block")
        os.environ["RPR_SELECT_RADIX"] = "0"
    if Q <= 6 and B <= 10 and N <= 60_000:
        pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(synth.codes_to_docid_to_smtid(codes)), V)
        seqs, sc = beam_ref.beam_search_ref(t5_ref.T5RefCached(sd, dims), pm, ids, mask, B, L, apply_log_softmax_for_scores=lsm,
                                            use_kv_cache=True)
        ref_tok, ref_sc = seqs.numpy().reshape(Q, B, L + 1)[:, :, 1:], sc.numpy().reshape(Q, B)
        ctx.set_forced_tail(1)
        got = E.search(model, trie, ti, tm, B, L, **kw)
        live = ref_sc > -1e6
        near = np.zeros((Q, B), dtype=bool)
        near[:, 1:] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
        near[:, :-1] |= (ref_sc[:, :-1] - ref_sc[:, 1:]) <= ORDER_TOL
        assert ((got.tokens.cpu().numpy() == ref_tok).all(axis=2) | near | ~live).all(), f"case {case}: differs from the oracle"
        assert np.abs((got.scores.cpu().numpy() - ref_sc) * live).max() <= SCORE_TOL, f"case {case}: scores differ from the oracle"
        notes.append("oracle ok")
        n_oracle += 1
    torch.cuda.synchronize()
    assert ctx.status(clear=True) & 1 == 0, f"case {case}: saturation flag"
    print(f"case {case:3d}: N={N} L={L} V={V} B={B} Q={Q} skew={int(skew)} dup={int(dup)} logsm={int(lsm)} dkv={dims.d_kv}: " + "; ".join(notes), flush=True)
    del model, trie
ctx.set_forced_tail(1)
os.environ.pop("RPR_SELECT_RADIX", None)
print(f"{n_cases} cases passed ({n_oracle} also against the CPU oracle; {n_excused} queries excused at a pruning margin below {BOUNDARY_TOL})")
