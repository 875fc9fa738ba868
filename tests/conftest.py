import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """Search fixtures (g*). The training-step fixtures (f4_*) are listed by train_golden_names()."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("g"))


def train_golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz") and f.startswith("f4_"))


class Golden:
    """A committed fixture: inputs are regenerated from the stored seed/dims with
    ripor_amd.utils.synth and cross-checked against the stored copies; expected outputs are the
    reference's own outputs (tests/golden/make_golden.py)."""

    def __init__(self, name):
        from ripor_amd.utils import synth
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.z = z
        self.spec = json.loads(str(z["spec"]))
        self.dims = synth.ModelDims(**self.spec["dims"])
        s = self.spec
        self.N, self.Q, self.B, self.L, self.V, self.seed = s["N"], s["Q"], s["B"], s["L"], s["V"], s["seed"]
        self.log_softmax = bool(s.get("log_softmax", False))
        self.codes = synth.make_codes(self.N, self.L, self.V, seed=self.seed)
        assert (self.codes == z["codes"]).all(), "synthetic code generator drifted from the fixture"
        self.input_ids, self.attention_mask = synth.make_queries(self.Q, vocab_size=self.dims.vocab_size,
                                                                 seed=self.seed, max_len=20)
        assert (self.input_ids == z["input_ids"]).all(), "synthetic query generator drifted from the fixture"
        assert (self.attention_mask == z["attention_mask"]).all()
        self._sd = None

    @property
    def state_dict(self):
        from ripor_amd.utils import synth
        if self._sd is None:
            self._sd = synth.make_state_dict(self.dims, seed=self.seed)
        return self._sd

    @property
    def sequences(self):
        return self.z["sequences"]

    @property
    def sequences_scores(self):
        return self.z["sequences_scores"]


SCORE_TOL = 1e-4          # north_star: beam scores within 1e-4
ORDER_TOL = 2 * SCORE_TOL  # two final scores closer than this may legitimately come out in either order
PRUNE_TOL = 1e-3          # cumulative float64 scores at the pruning boundary (rank B-1 vs rank B of a step)


def prune_margins(g):
    """Per query: (smallest gap over the steps between the last kept candidate (rank B-1) and the first dropped
    one (rank B) of the reference's sorted candidates, number of steps whose gap is below PRUNE_TOL). Only these
    gaps decide WHICH sequences survive; gaps between kept candidates only permute slots. Dead candidates
    (-1e9 mask / initial beam scores) are not competitors."""
    ts = g.z["top_scores"]                      # [L, Q, min(B+1, B*V)] float64, sorted desc
    B = g.B
    if ts.shape[2] <= B:
        return np.full(g.Q, np.inf), np.zeros(g.Q, dtype=int)
    gap = ts[:, :, B - 1] - ts[:, :, B]
    gap = np.where(ts[:, :, B] > -1e8, gap, np.inf)
    return gap.min(axis=0), (gap < PRUNE_TOL).sum(axis=0)


def compare_ranked(g, tokens, scores, label=""):
    """Parity of one search result ([Q,B,L] tokens, [Q,B] float32 scores) with the reference's golden output.

    For every query:
      * the SET of returned smtid sequences must equal the reference's, unless the reference itself dropped a
        candidate by less than PRUNE_TOL at some step (then at most that many sequences may differ);
      * every sequence present in both carries the reference's score within SCORE_TOL;
      * at every rank whose reference score is more than ORDER_TOL away from both neighbours the token sequence
        must be identical (ranks inside a near-tie may swap).
    Returns a dict of counts; raises AssertionError with the offending query/rank. A fixture whose queries are ALL
    boundary-excused fails: such a fixture pins nothing."""
    Q, B, L = g.Q, g.B, g.L
    exp_tok = g.sequences.reshape(Q, B, L + 1)
    assert (exp_tok[:, :, 0] == 0).all()
    exp_tok = exp_tok[:, :, 1:]
    exp_sc = g.sequences_scores.reshape(Q, B).astype(np.float64)
    margins, near = prune_margins(g)
    stats = dict(queries=Q, boundary_excused=0, ranks_checked=0, ranks_in_near_tie=0, sequences_missing=0,
                 max_score_err=0.0)
    for q in range(Q):
        ref = {tuple(int(x) for x in exp_tok[q, r]): r for r in range(B)}
        got = {tuple(int(x) for x in tokens[q, r]): r for r in range(B)}
        assert len(got) == B or len(ref) < B, f"{g.name}{label} query {q}: duplicate sequences among the beams"
        missing = [s for s in ref if s not in got]
        allowed = int(near[q])
        if allowed:
            stats["boundary_excused"] += 1
        assert len(missing) <= allowed, (
            f"{g.name}{label} query {q}: {len(missing)} reference sequences missing from the result, "
            f"{allowed} allowed (pruning margin {margins[q]:.3g})")
        stats["sequences_missing"] += len(missing)
        for s, r in ref.items():
            if s in got:
                err = abs(float(scores[q, got[s]]) - exp_sc[q, r])
                stats["max_score_err"] = max(stats["max_score_err"], err)
                assert err <= SCORE_TOL, f"{g.name}{label} query {q} ref rank {r}: score differs by {err:.3g}"
        if missing:
            continue   # ranks are shifted by the replaced sequences; the set/score checks above still held
        for r in range(B):
            up = exp_sc[q, r - 1] - exp_sc[q, r] if r > 0 else np.inf
            dn = exp_sc[q, r] - exp_sc[q, r + 1] if r + 1 < B else np.inf
            if min(up, dn) > ORDER_TOL:
                stats["ranks_checked"] += 1
                assert (tokens[q, r] == exp_tok[q, r]).all(), (
                    f"{g.name}{label} query {q} rank {r}: tokens differ although the reference score is "
                    f"{min(up, dn):.3g} away from its neighbours")
            else:
                stats["ranks_in_near_tie"] += 1
    assert stats["boundary_excused"] < Q or Q == 0, (
        f"{g.name}{label}: every query sits on a pruning near-tie; the fixture pins nothing — regenerate it")
    print(f"[parity] {g.name}{label}: {stats}")
    return stats


@pytest.fixture(scope="session")
def golden_cache():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get
