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
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


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

    def step_margins(self):
        """Per query: the smallest gap between consecutive candidates among the sorted top-(B+1)
        cumulative scores over all steps, recomputed from the reference's per-step processed scores.
        A tiny margin means fp32 summation-order noise may legitimately reorder/replace beams."""
        if "step_scores" not in self.z.files:
            return None
        ss = self.z["step_scores"]  # [L, Q*B, V] float64 = logits + (1-mask)*(-1e9)
        Q, B, V = self.Q, self.B, self.V
        beam = np.zeros((Q, B), dtype=np.float64)
        beam[:, 1:] = np.float32(-1e9)
        margins = np.full(Q, np.inf)
        for t in range(ss.shape[0]):
            cand = (ss[t].reshape(Q, B, V) + beam[:, :, None]).reshape(Q, B * V)
            order = np.argsort(-cand, axis=1, kind="stable")[:, : B + 1]
            top = np.take_along_axis(cand, order, axis=1)
            gaps = top[:, :-1] - top[:, 1:]
            live = top[:, :-1] > -1e8  # gaps among dead (-1e9) candidates are irrelevant
            gaps = np.where(live, gaps, np.inf)
            margins = np.minimum(margins, gaps.min(axis=1))
            beam = top[:, :B]
        return margins


@pytest.fixture(scope="session")
def golden_cache():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get
