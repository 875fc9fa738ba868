"""not-gpu: the CPU oracle (oracle/) pinned against the golden vectors produced by the imported
reference (tests/golden/make_golden.py). The full-prefix oracle restates the reference loop with the
same torch ops, so it is expected to reproduce the reference's integers exactly and its float32
scores to ~1e-6; the KV-cached oracle variant is the same math in a different summation order."""
import numpy as np
import pytest
import torch

from conftest import compare_ranked, golden_names, prune_margins
from oracle import beam_ref, t5_ref
from ripor_amd.utils import synth

# The full-prefix oracle (the reference's cost profile: every step recomputes the whole decoder prefix) runs on the
# mini and t5-base fixtures; the t5-large B=100 and the 64-query config-1 fixtures take minutes that way and are
# covered by the KV-cached oracle (same arithmetic, different summation order) through the ranked comparison.
FULL = [n for n in golden_names() if n.startswith(("g1_", "g2_"))]
FAST = [n for n in golden_names() if not n.startswith("g2_")]


def _mask_fn(g):
    d2s = synth.codes_to_docid_to_smtid(g.codes)
    return beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(d2s), g.V)


@pytest.mark.parametrize("name", FULL)
def test_oracle_reproduces_reference(golden_cache, name):
    g = golden_cache(name)
    torch.set_num_threads(8)
    rec = {}
    seqs, scores = beam_ref.beam_search_ref(t5_ref.T5Ref(g.state_dict, g.dims), _mask_fn(g), g.input_ids,
                                            g.attention_mask, g.B, g.L, g.log_softmax, record=rec)
    np.testing.assert_allclose(rec["encoder_out"], g.z["encoder_out"], atol=1e-5, rtol=1e-5)
    assert (seqs.numpy() == g.sequences).all(), "oracle smtid sequences differ from the reference"
    np.testing.assert_allclose(scores.numpy(), g.sequences_scores, atol=1e-6, rtol=0)
    assert scores.dtype == torch.float32 and seqs.dtype == torch.int64
    # smtid strings as the reference's convert_ptsmtids_to_strsmtid produced them
    strs = beam_ref.smtid_strings(seqs, g.B, g.L)
    assert (np.array(strs) == g.z["smtid_strings"]).all()
    # per-step logits: reference stores logits + (1-mask)*(-1e9); compare on valid entries
    if "step_scores" in g.z.files and not g.log_softmax:
        ss = g.z["step_scores"]
        for t in range(g.L):
            valid = ss[t] > -1e8
            np.testing.assert_allclose(rec["steps"][t]["logits"][valid], ss[t][valid], atol=1e-5, rtol=1e-6)


@pytest.mark.parametrize("name", FAST)
def test_kv_cached_oracle_equals_full_prefix(golden_cache, name):
    g = golden_cache(name)
    torch.set_num_threads(8)
    seqs, scores = beam_ref.beam_search_ref(t5_ref.T5RefCached(g.state_dict, g.dims), _mask_fn(g), g.input_ids,
                                            g.attention_mask, g.B, g.L, g.log_softmax, use_kv_cache=True)
    got = seqs.numpy().reshape(g.Q, g.B, g.L + 1)
    assert (got[:, :, 0] == 0).all()
    # the ranked comparison the GPU parity tests use (tests/conftest.py): set of sequences, per-sequence scores,
    # exact tokens at every rank outside a score near-tie
    stats = compare_ranked(g, got[:, :, 1:], scores.numpy().reshape(g.Q, g.B), label=" (KV-cached oracle)")
    assert stats["boundary_excused"] == 0 and stats["sequences_missing"] == 0


@pytest.mark.parametrize("name", golden_names())
def test_prefix_mask_oracle_matches_reference_processor(golden_cache, name):
    g = golden_cache(name)
    pm = _mask_fn(g)
    seen = 0
    for key in g.z.files:
        if key.startswith("pm_prefix_T"):
            T = int(key[len("pm_prefix_T"):])
            expect = np.unpackbits(g.z[f"pm_mask_T{T}"], axis=1)[:, : g.V]
            got = pm(g.z[key])
            assert got.dtype == np.float64
            assert (got.astype(np.uint8) == expect).all(), (name, T)
            seen += 1
    assert seen >= 2


@pytest.mark.parametrize("name", [n for n in golden_names() if not n.startswith("g6_")])   # (g6: V = 1024, two bytes per code)
def test_sorted_matrix_mask_equals_the_dict_mask(golden_cache, name):
    """SortedPrefixMaskRef (the oracle's mask for corpora whose dicts do not fit host RAM: tests/test_gpu_fullsize.py) against
    the masks the REFERENCE's processor returned for the prefixes of the golden searches, against the dict mask on the same
    prefixes, and against the dict mask on random prefixes — known ones, unknown ones, every length."""
    g = golden_cache(name)
    if int(np.asarray(g.codes).max()) >= 256:
        pytest.skip("one byte per code")
    pm, sm = _mask_fn(g), beam_ref.SortedPrefixMaskRef(g.codes, g.V)
    seen = 0
    for key in g.z.files:
        if key.startswith("pm_prefix_T"):
            T = int(key[len("pm_prefix_T"):])
            expect = np.unpackbits(g.z[f"pm_mask_T{T}"], axis=1)[:, : g.V]
            got = sm(g.z[key])
            assert got.dtype == np.float64 and (got.astype(np.uint8) == expect).all(), (name, T)
            seen += 1
    assert seen >= 2
    rng = np.random.default_rng(7)
    codes = np.asarray(g.codes)
    N, Lc = codes.shape
    for T in range(1, Lc + 1):
        known = codes[rng.integers(0, N, 16), : T - 1]
        junk = rng.integers(0, g.V, (8, T - 1))
        half = known.copy()
        if T > 1:
            half[:, -1] = rng.integers(0, g.V, len(half))          # a known parent with a random last token
        pre = np.concatenate([known, junk, half], 0)
        ids = np.concatenate([np.full((len(pre), 1), -1, dtype=np.int64), pre.astype(np.int64)], 1)
        assert (pm(ids) == sm(ids)).all(), (name, T)


def test_relative_position_buckets_known_answers():
    """SURVEY.md Appendix B known answers of HF's bucket function."""
    dec = t5_ref.bucket_table(False, 40)
    assert dec.tolist() == list(range(16)) + [16, 16, 16, 17, 17, 18, 18, 18, 19, 19, 19, 20, 20, 20, 20,
                                              21, 21, 21, 21, 22, 22, 22, 22, 22]
    enc = t5_ref.bucket_table(True, 256)
    neg = [int(enc[255 - n]) for n in range(40)]
    assert neg == [0, 1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10,
                   11, 11, 11, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 12, 12, 12]
    pos = {n: int(enc[255 + n]) for n in (40, 45, 50, 57, 64, 72, 80, 90, 91, 100, 128, 255)}
    assert list(pos.values()) == [28, 28, 29, 29, 30, 30, 30, 30, 31, 31, 31, 31]
    assert [int(enc[255 + n]) for n in range(1, 8)] == [17, 18, 19, 20, 21, 22, 23]


def test_fixture_margins_are_comfortable(golden_cache):
    """The bit-exact GPU claims rest on pruning margins >> fp32 noise: no query of any fixture may sit closer than
    PRUNE_TOL to a pruning decision (the checker would excuse it), and every fixture must check most of its ranks."""
    total = 0
    for name in golden_names():
        g = golden_cache(name)
        m, near = prune_margins(g)
        assert (near == 0).all(), (name, m)
        total += len(m)
        sc = g.sequences_scores.reshape(g.Q, g.B).astype(np.float64)
        if g.B > 1:
            ties = int(((-np.diff(sc, axis=1)) < 2e-4).sum())
            assert ties <= max(3, g.Q * g.B // 20) or "tiny_trie" in name, (name, ties)
    assert total >= 90


# ---- forward of the prefix-oriented ranking fine-tune step ---------------------------------------
class TrainGolden:
    def __init__(self, name):
        import json
        import os
        from conftest import GOLDEN_DIR
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.spec = json.loads(str(self.z["spec"]))
        self.dims = synth.ModelDims(**self.spec["dims"])
        self.bz, self.L, self.V, self.seed = self.spec["bz"], self.spec["L"], self.spec["V"], self.spec["seed"]
        self.teacher = {k: self.z[k] for k in self.z.files if k.endswith("_scores") and "teacher" in k}
        self.losses = dict(zip([str(x) for x in self.z["loss_names"]], self.z["losses"].tolist()))

    @property
    def state_dict(self):
        return synth.make_state_dict(self.dims, seed=self.seed)


def grad_sample_indices(name, shape, n=48):
    """The seeded sample positions tests/golden/make_golden.py::grad_samples used for this tensor."""
    numel = int(np.prod(shape))
    if numel <= 4096:
        return np.arange(numel, dtype=np.int64)
    return np.sort(synth.randint(f"gradidx/{name}", (n,), 0, numel, seed=7)).astype(np.int64)


def check_grads_against_fixture(g, grads, rel=2e-4, label=""):
    """grads: name -> numpy array (reference state-dict names). Every gradient tensor of the reference must match in
    Frobenius norm and at the sampled positions; the tolerance is relative to the tensor's largest sampled magnitude."""
    names = [str(x) for x in g.z["grad_names"]]
    norms, counts, samples = g.z["grad_norms"], g.z["grad_sample_counts"], g.z["grad_samples"]
    off, worst = 0, (0.0, None)
    for n, nr, c in zip(names, norms, counts):
        ref = samples[off:off + c]
        off += c
        key = "shared.weight" if n == "encoder.embed_tokens.weight" else n
        arr = np.asarray(grads[key], dtype=np.float64)
        got = arr.reshape(-1)[grad_sample_indices(n, arr.shape)]
        scale = max(np.abs(ref).max(), nr / np.sqrt(arr.size), 1e-12)
        err = np.abs(got - ref).max() / scale
        worst = max(worst, (err, n))
        assert err <= rel * 10, f"{g.name}{label} gradient of {n}: sampled entries differ by {err:.2e} of its scale"
        gnr = np.sqrt((arr ** 2).sum())
        assert abs(gnr - nr) <= rel * max(nr, 1e-12), f"{g.name}{label} gradient norm of {n}: {gnr} vs {nr}"
    return worst


def test_train_step_oracle_reproduces_reference_gradients():
    """Backward + one AdamW step: autograd through the oracle's own forward vs the gradients of the
    imported reference module, the clipped AdamW update and the losses after the step."""
    from conftest import train_golden_names
    from oracle import train_ref
    for name in [n for n in train_golden_names() if "mini" in n]:
        g = TrainGolden(name)
        assert "grad_names" in g.z.files
        torch.set_num_threads(8)
        model = t5_ref.T5Ref(g.state_dict, g.dims)
        before = {k: v.clone() for k, v in model.sd.items()}
        lr = float(g.z["step_lr"])
        losses, total, grads, gnorm = train_ref.train_step(model, g.z["input_ids"], g.z["attention_mask"],
                                                           g.z["pos_doc_encoding"], g.z["neg_doc_encoding"], g.teacher, lr=lr)
        assert abs(total - float(g.z["total_loss"])) <= 1e-5 * abs(total)
        assert abs(gnorm - float(g.z["grad_global_norm"])) <= 1e-4 * gnorm
        worst = check_grads_against_fixture(g, {k: v.numpy() for k, v in grads.items()}, label=" (oracle)")
        # parameter update of the clipped AdamW step at the sampled positions
        names = [str(x) for x in g.z["grad_names"]]
        off = 0
        for n, c in zip(names, g.z["grad_sample_counts"]):
            ref = g.z["param_delta_samples"][off:off + c]
            off += c
            key = "shared.weight" if n == "encoder.embed_tokens.weight" else n
            d = (model.sd[key] - before[key]).numpy().reshape(-1)[grad_sample_indices(n, before[key].shape)]
            np.testing.assert_allclose(d, ref, atol=2e-2 * lr, rtol=0, err_msg=f"{name} update of {n}")
        with torch.no_grad():
            after = train_ref.lng_knp_margin_mse(model, g.z["input_ids"], g.z["attention_mask"], g.z["pos_doc_encoding"],
                                                 g.z["neg_doc_encoding"], g.teacher)
        for k, v in zip(sorted(g.losses), g.z["losses_after_step"]):
            assert abs(float(after[k]) - v) <= 2e-3 * max(1.0, abs(v)), (name, k, float(after[k]), v)
        print(f"[train-oracle] {name}: worst sampled-gradient error {worst[0]:.2e} of scale ({worst[1]})")


def test_train_forward_oracle_reproduces_reference_losses():
    from conftest import train_golden_names
    from oracle import train_ref
    names = train_golden_names()
    assert len(names) >= 3
    for name in names:
        g = TrainGolden(name)
        torch.set_num_threads(8)
        with torch.no_grad():
            out = train_ref.lng_knp_margin_mse(t5_ref.T5Ref(g.state_dict, g.dims), g.z["input_ids"], g.z["attention_mask"],
                                               g.z["pos_doc_encoding"], g.z["neg_doc_encoding"], g.teacher)
        assert set(g.losses) == set(train_ref.LOSS_NAMES[g.L])
        np.testing.assert_allclose(out["pos_position_scores"].numpy(), g.z["pos_position_scores"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(out["neg_position_scores"].numpy(), g.z["neg_position_scores"], atol=2e-5, rtol=1e-5)
        for k, v in g.losses.items():
            assert abs(float(out[k]) - v) <= 1e-5 * max(1.0, abs(v)), (name, k, float(out[k]), v)
