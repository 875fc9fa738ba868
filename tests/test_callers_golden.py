"""not-gpu: the caller functions either side of the search (fixtures G5 / G6) pinned to what the
reference's OWN functions produced. tests/golden/c5_callers_mini.npz was written by tests/golden/make_golden.py running
/root/reference/t5_pretrainer/evaluate.py — constrained_decode_doc (:87-132), constrained_decode (:45-85),
constrained_decode_smtid (:134-178), t5seq_aq_retrieve_docids_2 (:489-526), t5seq_aq_get_qid_to_smtid_rankdata_2 (:600-655)
— on two DistributedSampler(shuffle=False) shards of a seeded mini model: duplicated smtids (several docids per smtid), an
smtid missing from the lookup, a ragged last batch, a wrap-around duplicate query, a prefix search at 4 of 8 positions.

Here: (1) the oracle (beam search + caller restatement) reproduces the reference's per-rank files; (2) the product's host
logic (ripor_amd.evaluate: the three callers with the search replaced by the oracle's outputs, and the two merges) writes
the same files; (3) query sharding equals torch's sampler lists as recorded. tests/test_gpu_api.py repeats (2) with the
HIP search underneath."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import beam_ref, t5_ref
from ripor_amd import evaluate as EV
from ripor_amd.dataset.sharding import shard_indices
from ripor_amd.tasks import generation as GEN
from ripor_amd.utils import synth

SCORE_TOL = 1e-5   # per position; run scores are multiplied by the number of positions


class CallerFixture:
    def __init__(self, name="c5_callers_mini"):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.spec = json.loads(str(z["spec"]))
        s = self.spec
        self.N, self.Q, self.B, self.L, self.Lp, self.V = s["N"], s["Q"], s["B"], s["L"], s["Lp"], s["V"]
        self.world, self.batch_size, self.seed = s["world"], s["batch_size"], s["seed"]
        self.dims = synth.ModelDims(**s["dims"])
        self.codes, self.qids = z["codes"], z["qids"]
        self.input_ids, self.attention_mask = synth.make_queries(self.Q, vocab_size=self.dims.vocab_size, seed=self.seed, max_len=20)
        assert (self.input_ids == z["input_ids"]).all() and (self.attention_mask == z["attention_mask"]).all()
        self.dropped = str(z["dropped_smtid"])
        self.res = json.loads(str(z["results"]))
        self.sampler = json.loads(str(z["sampler_head_tail_len"]))
        self.docids = [str(i) for i in range(self.N)]
        self.d2s = synth.codes_to_docid_to_smtid(self.codes)
        assert list(self.d2s.keys()) == self.docids
        self._sd = None

    @property
    def state_dict(self):
        if self._sd is None:
            self._sd = synth.make_state_dict(self.dims, seed=self.seed)
        return self._sd

    def lookup(self, n_tok, drop=True):
        d = EV.build_smtid_to_docids(self.d2s, n_tok)
        assert d == beam_ref.build_smtid_to_docids(self.d2s, n_tok)
        if drop:
            d.pop(self.dropped, None)
        return d

    def batches(self, rank):
        idx = self.res["shard_indices"][str(rank)]
        bs = self.batch_size
        return [{"id": torch.from_numpy(self.qids[idx[i:i + bs]]), "input_ids": torch.from_numpy(self.input_ids[idx[i:i + bs]]),
                 "attention_mask": torch.from_numpy(self.attention_mask[idx[i:i + bs]])} for i in range(0, len(idx), bs)]


@pytest.fixture(scope="module")
def fx():
    return CallerFixture()


def _same_nested(got, ref, tol, path=""):
    """Same keys at every level (as strings, the way json stores them), leaf scores within tol."""
    if isinstance(ref, dict):
        assert isinstance(got, dict), path
        g = {str(k): v for k, v in got.items()}
        assert list(g.keys()) == list(ref.keys()), f"{path}: keys / insertion order differ: {list(g)[:6]} vs {list(ref)[:6]}"
        for k in ref:
            _same_nested(g[k], ref[k], tol, f"{path}/{k}")
    else:
        assert abs(float(got) - float(ref)) <= tol, f"{path}: {got} vs {ref}"


def oracle_generate(fx, cache=None):
    """A stand-in for generate_for_constrained_prefix_beam_search that answers with the ORACLE's search (full-prefix
    T5Ref: the reference's arithmetic) — sequences, float32 scores and the sorted-row ranges the product's range path uses."""
    torch.set_num_threads(8)
    model = t5_ref.T5Ref(fx.state_dict, fx.dims)
    pm = beam_ref.PrefixMaskRef(beam_ref.build_list_smtid_to_nextids(fx.d2s), fx.V)
    order = np.lexsort(fx.codes.T[::-1])
    sorted_codes = fx.codes[order]

    def fn(model_, processor_, input_ids=None, attention_mask=None, max_new_tokens=None, num_beams=None,
           apply_log_softmax_for_scores=False, **kw):
        key = (input_ids.numpy().tobytes(), max_new_tokens, num_beams, bool(apply_log_softmax_for_scores))
        if cache is not None and key in cache:
            return cache[key]
        seqs, scores = beam_ref.beam_search_ref(model, pm, input_ids.numpy(), attention_mask.numpy(), num_beams, max_new_tokens,
                                                apply_log_softmax_for_scores)
        lo, hi = [], []
        for s in seqs.numpy():
            m = (sorted_codes[:, :max_new_tokens] == s[1:][None, :]).all(1)
            hit = np.flatnonzero(m)
            lo.append(int(hit[0]) if hit.size else 0)
            hi.append(int(hit[-1]) + 1 if hit.size else 0)
        out = GEN.BeamSearchEncoderDecoderOutput(sequences=seqs, sequences_scores=scores, row_lo=torch.tensor(lo),
                                                 row_hi=torch.tensor(hi))
        if cache is not None:
            cache[key] = out
        return out

    class Trie:
        perm = order.astype(np.int64)

    class Proc:
        def trie(self, device):
            return Trie()

    return fn, Proc()


def test_fixture_exercises_the_edge_cases(fx):
    r = fx.res
    assert sorted(r["shard_indices"]) == ["0", "1"] and r["shard_indices"]["1"][-1] == 0, "wrap-around duplicate query expected"
    merged = r["doc"]["merged"]
    assert len(merged) == fx.Q
    assert max(len(v) for v in merged.values()) > fx.B, "no query returns an smtid with several docids"
    full = fx.lookup(fx.L, drop=False)
    assert fx.dropped in full and any(len(v) > 1 for v in full.values())
    # the dropped smtid is returned by a search (the reference printed and skipped it): its docids are missing from the run
    assert not any(d in v for v in merged.values() for d in full[fx.dropped])
    assert max(len(d) for v in r["prefix"]["merged"].values() for d in v.values()) >= 20   # many docids per prefix smtid


def test_oracle_callers_reproduce_the_reference_files(fx):
    gen, _ = oracle_generate(fx)
    for variant, ls in (("doc", False), ("doc_logsoftmax", True)):
        for rank in range(fx.world):
            got = {}
            for b in fx.batches(rank):
                o = gen(None, None, input_ids=b["input_ids"], attention_mask=b["attention_mask"], max_new_tokens=fx.L,
                        num_beams=fx.B, apply_log_softmax_for_scores=ls)
                got.update(beam_ref.constrained_decode_doc_ref(b["id"].tolist(), o.sequences, o.sequences_scores, fx.lookup(fx.L),
                                                               fx.B, fx.L, ls))
            _same_nested(got, fx.res[variant]["shards"][str(rank)], SCORE_TOL * fx.L, f"{variant}/run_{rank}")


def test_product_callers_on_oracle_outputs_write_the_reference_files(fx, monkeypatch, tmp_path):
    cache = {}
    gen, proc = oracle_generate(fx, cache)
    monkeypatch.setattr(EV, "generate_for_constrained_prefix_beam_search", gen)
    table = EV.DocidTable(fx.docids)
    for variant, ls in (("doc", False), ("doc_logsoftmax", True)):
        d = tmp_path / variant
        d.mkdir()
        for rank in range(fx.world):
            ref = fx.res[variant]["shards"][str(rank)]
            run = EV.constrained_decode_doc(None, fx.batches(rank), proc, fx.lookup(fx.L), fx.L, "cpu", str(d), rank, topk=fx.B,
                                            apply_log_softmax_for_scores=ls)
            _same_nested(run, ref, SCORE_TOL * fx.L, f"{variant}/run_{rank} (dict lookup)")
            _same_nested(json.load(open(d / f"run_{rank}.json")), ref, SCORE_TOL * fx.L, f"{variant}/run_{rank}.json")
            # the sorted-row range lookup has no "missing smtid": it answers from the trie, so the docids of the dropped
            # smtid are present there and everything else is identical
            rng = EV.constrained_decode_doc(None, fx.batches(rank), proc, table, fx.L, "cpu", str(d), rank, topk=fx.B,
                                            apply_log_softmax_for_scores=ls, write=False)
            extra = set(fx.lookup(fx.L, drop=False)[fx.dropped])
            for q, docs in ref.items():
                g = {k: v for k, v in rng[int(q)].items() if k not in extra}
                assert set(g) == set(docs), (variant, rank, q)
                for k in docs:
                    assert abs(g[k] - docs[k]) <= SCORE_TOL * fx.L
        merged = EV.merge_runs(str(d), expected_files=fx.world)
        ref = fx.res[variant]["merged"]
        assert set(merged) == set(ref)
        for q in ref:   # the merge order of the rank files is os.listdir's: compare as sets of (docid, score)
            assert set(merged[q]) == set(ref[q]), (variant, q)
            for k in ref[q]:
                assert abs(merged[q][k] - ref[q][k]) <= SCORE_TOL * fx.L
        assert os.listdir(d) == ["run.json"]
    # smtid-level file of the full-length search (evaluate.py:45-85): raw scores, unknown smtids skipped
    d = tmp_path / "smtid"
    d.mkdir()
    for rank in range(fx.world):
        out = EV.constrained_decode(None, fx.batches(rank), proc, fx.lookup(fx.L), fx.L, "cpu", str(d), rank, topk=fx.B)
        _same_nested(out, fx.res["smtid"]["shards"][str(rank)], SCORE_TOL, f"qid_to_smtid_{rank}")
    # training-data generation pass (evaluate.py:134-178): nested output of the prefix search, dict and range lookups
    d = tmp_path / "prefix"
    d.mkdir()
    plookup = fx.lookup(fx.Lp, drop=False)
    for rank in range(fx.world):
        ref = fx.res["prefix"]["shards"][str(rank)]
        out = EV.constrained_decode_smtid(None, fx.batches(rank), proc, plookup, fx.Lp, "cpu", str(d), rank, topk=fx.B)
        _same_nested(out, ref, SCORE_TOL * fx.Lp, f"qid_smtid_rankdata_{rank}")
        rng = EV.constrained_decode_smtid(None, fx.batches(rank), proc, table, fx.Lp, "cpu", str(d), rank, topk=fx.B, write=False)
        for q in ref:
            assert list(rng[int(q)]) == list(ref[q])
            for s in ref[q]:
                assert set(rng[int(q)][s]) == set(ref[q][s]), (q, s)
    merged = EV.merge_qid_smtid_rankdata(str(d), expected_files=fx.world)
    ref = fx.res["prefix"]["merged"]
    assert set(merged) == set(ref)
    for q in ref:
        assert set(merged[q]) == set(ref[q])
        for s in ref[q]:
            assert set(merged[q][s]) == set(ref[q][s])
            for k in ref[q][s]:
                assert abs(merged[q][s][k] - ref[q][s][k]) <= SCORE_TOL * fx.Lp


def test_query_shards_equal_the_recorded_sampler_lists(fx):
    for rank in range(fx.world):
        assert shard_indices(fx.Q, fx.world, rank) == fx.res["shard_indices"][str(rank)]
    for w, per_rank in fx.sampler.items():
        for rank, rec in enumerate(per_rank):
            idx = shard_indices(6980, int(w), rank)
            assert idx[:4] + idx[-4:] + [len(idx)] == rec, (w, rank)
