"""not-gpu: the data side and the schedule of the prefix-oriented ranking fine-tune.

* ``LngKnpMarginMSEforT5SeqAQDataset`` / ``...Collator`` (ripor_amd/dataset/lng_knp.py) against what the REFERENCE's own
  classes (dataset/dataset.py:418-525, dataset/data_collator.py:11-88) produced on the same files
  (tests/golden/c6_lngknp_data.npz, written by tests/golden/make_golden.py::make_lngknp_data_case): items under
  ``random.seed(5)``, collated batches, smtid lengths 8 / 16 / 32, both lookup modes.
* the learning-rate schedule against HF's ``get_linear_schedule_with_warmup`` driven the way ``Trainer`` drives it.
* the loop (LngKnpTrainer) with the device step replaced by a recorder: learning rates, step counts, sampler sharding,
  checkpoint rotation. tests/test_gpu_train.py runs the loop on the device."""
import json
import math
import os
import random
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from ripor_amd.dataset.lng_knp import LngKnpMarginMSEforT5SeqAQCollator, LngKnpMarginMSEforT5SeqAQDataset
from ripor_amd.tasks import trainer as T


class WordTokenizer:   # the tokenizer the fixture was generated with (make_golden.WordTokenizer)
    def __call__(self, texts, add_special_tokens=True, padding="longest", truncation="longest_first", max_length=64,
                 return_attention_mask=True, return_tensors="pt"):
        ids = [[3 + zlib.crc32(w.encode()) % 97 for w in t.split()][: max_length - 1] + [1] for t in texts]
        m = max(len(x) for x in ids)
        return {"input_ids": torch.tensor([x + [0] * (m - len(x)) for x in ids]),
                "attention_mask": torch.tensor([[1] * len(x) + [0] * (m - len(x)) for x in ids])}

    def save_pretrained(self, path):
        open(os.path.join(path, "tokenizer.txt"), "w").write("word tokenizer\n")


def _write(root, files):
    os.makedirs(root / "queries")
    os.makedirs(root / "docs")
    (root / "queries" / "raw.tsv").write_text(files["queries"])
    (root / "docs" / "raw.tsv").write_text(files["docs"])
    (root / "examples.jsonl").write_text(files["examples"])
    (root / "docid_to_smtid.json").write_text(files["docid_to_smtid"])


CASES = json.loads(str(np.load(os.path.join(GOLDEN_DIR, "c6_lngknp_data.npz"))["cases"]))


@pytest.mark.parametrize("name", sorted(CASES))
def test_dataset_and_collator_equal_the_reference_run(name, tmp_path):
    c = CASES[name]
    _write(tmp_path, c["files"])
    as_docid = name.endswith("smtid")
    ds = LngKnpMarginMSEforT5SeqAQDataset(dataset_path=str(tmp_path / "examples.jsonl"), document_dir=str(tmp_path / "docs"),
                                          query_dir=str(tmp_path / "queries"),
                                          docid_to_smtid_path=None if as_docid else str(tmp_path / "docid_to_smtid.json"),
                                          smtid_as_docid=as_docid)
    assert len(ds) == c["length"]
    random.seed(5)
    items = [list(ds[i]) for i in c["order"]]
    assert items == c["items"], "items differ from the reference dataset's (same files, same random seed)"
    batch = LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), max_length=c["max_length"])([tuple(it) for it in items[:4]])
    flat = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            flat.update({f"{k}.{kk}": vv for kk, vv in v.items()})
        else:
            flat[k] = v
    assert sorted(flat) == sorted(c["batch"])
    for k, ref in c["batch"].items():
        got = flat[k]
        assert got.dtype == (torch.float32 if "scores" in k else torch.int64), (k, got.dtype)
        assert got.tolist() == ref, k
    # doc encoding shifted right = decoder_input_ids (dataset.py:488-500), the same query on both sides
    assert torch.equal(batch["pos_tokenized_query"]["decoder_input_ids"][:, 1:], batch["pos_doc_encoding"][:, :-1])
    assert (batch["pos_tokenized_query"]["decoder_input_ids"][:, 0] == -1).all()
    assert torch.equal(batch["pos_tokenized_query"]["input_ids"], batch["neg_tokenized_query"]["input_ids"])


def test_dataset_rejects_inconsistent_files(tmp_path):
    c = CASES["L16_smtid"]
    bad = dict(c["files"])
    ex = [json.loads(l) for l in bad["examples"].splitlines()]
    for e in ex:
        e.pop("smtid_8_scores")
    bad["examples"] = "\n".join(json.dumps(e) for e in ex) + "\n"
    _write(tmp_path, bad)
    with pytest.raises(AssertionError):
        LngKnpMarginMSEforT5SeqAQDataset(str(tmp_path / "examples.jsonl"), None, str(tmp_path / "queries"), None, True)
    with pytest.raises(ValueError):
        LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), 8)([(1, 2, 3)])


@pytest.mark.parametrize("max_steps,ratio", [(250, 0.04), (7, 0.04), (100, 0.0), (33, 0.5)])
def test_learning_rate_schedule_is_hf_linear_with_warmup(max_steps, ratio):
    """HF Trainer: scheduler = get_linear_schedule_with_warmup(opt, ceil(max_steps * warmup_ratio), max_steps); the optimizer
    step i (0-based) runs at lambda(i) * lr, scheduler.step() follows it."""
    from transformers import get_linear_schedule_with_warmup
    lr = 1e-4
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=lr)
    warm = T.get_warmup_steps(max_steps, ratio)
    assert warm == math.ceil(max_steps * ratio)
    sched = get_linear_schedule_with_warmup(opt, warm, max_steps)
    for i in range(max_steps):
        assert abs(opt.param_groups[0]["lr"] - lr * T.linear_schedule_with_warmup(i, warm, max_steps)) <= 1e-12, i
        opt.step()
        sched.step()


class _Recorder:
    """Stands in for T5SeqAQEncoderForLngKnpMarginMSE: records what the loop feeds the device step."""

    class _Base:
        class _EM:
            class ctx:
                prec = "f16x2"
                seen = []

                @classmethod
                def set_precision(cls, p):
                    cls.prec = p
                    cls.seen.append(p)

                @classmethod
                def get_precision(cls):
                    return cls.prec

            def export_state_dict(self):
                return {"w": torch.ones(2)}

        def engine_model(self):
            return self._EM()

    def __init__(self):
        self.base_model, self.calls = self._Base(), []

    def training_step(self, lr, max_grad_norm=1.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **batch):
        self.calls.append(dict(lr=lr, clip=max_grad_norm, betas=betas, eps=eps, wd=weight_decay, n=batch["pos_doc_encoding"].shape[0],
                               first=batch["pos_doc_encoding"][0].tolist()))
        return {"rank": torch.tensor(2.0), "rank_4": torch.tensor(1.0)}


def test_loop_schedule_steps_and_logging(tmp_path):
    c = CASES["L8_smtid"]
    _write(tmp_path, c["files"])
    ds = LngKnpMarginMSEforT5SeqAQDataset(str(tmp_path / "examples.jsonl"), None, str(tmp_path / "queries"), None, True)
    coll = LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), 16)
    args = T.LngKnpTrainingArgs(output_dir=str(tmp_path / "out"), learning_rate=1e-4, warmup_ratio=0.25, per_device_train_batch_size=2,
                                num_train_epochs=4, logging_steps=5, save_steps=0, bf16=True)
    rec = _Recorder()
    logs = []
    tr = T.LngKnpTrainer(rec, ds, coll, args, log=logs.append)
    assert tr.steps_per_epoch == 3 and tr.max_steps == 12 and tr.warmup_steps == 3      # 5 examples, batches of 2, 4 epochs
    hist = tr.train()
    # bf16 operands for the duration of the run only: the ctx is shared with the search path of the process
    assert rec.base_model._EM.ctx.seen[-2:] == ["bf16", "f16x2"] and rec.base_model._EM.ctx.prec == "f16x2"
    assert len(rec.calls) == 12 and [c_["n"] for c_ in rec.calls[:3]] == [2, 2, 1]       # the ragged last batch is kept
    want = [1e-4 * T.linear_schedule_with_warmup(i, 3, 12) for i in range(12)]
    assert [c_["lr"] for c_ in rec.calls] == want and want[0] == 0.0 and abs(want[3] - 1e-4) < 1e-18 and want[-1] > 0
    assert all(c_["clip"] == 1.0 and c_["betas"] == (0.9, 0.999) and c_["eps"] == 1e-8 and c_["wd"] == 0.0 for c_ in rec.calls)
    # epochs are reshuffled (DistributedSampler(shuffle=True).set_epoch): the first example of an epoch changes
    assert len({tuple(rec.calls[3 * e]["first"]) for e in range(4)}) > 1
    assert [h["step"] for h in hist] == [5, 10, 12] and abs(hist[0]["loss"] - 3.0) < 1e-6 and len(logs) == 3
    assert json.loads(logs[0])["rank_4"] == 1.0
    with pytest.raises(NotImplementedError):
        T.LngKnpTrainer(rec, ds, coll, T.LngKnpTrainingArgs(output_dir="x", ln_to_weight={"rank": 0.5}))


def test_checkpoint_rotation(tmp_path):
    c = CASES["L8_smtid"]
    _write(tmp_path, c["files"])
    ds = LngKnpMarginMSEforT5SeqAQDataset(str(tmp_path / "examples.jsonl"), None, str(tmp_path / "queries"), None, True)
    args = T.LngKnpTrainingArgs(output_dir=str(tmp_path / "out"), per_device_train_batch_size=5, max_steps=7, save_steps=2,
                                save_total_limit=2, logging_steps=100, bf16=False)
    os.makedirs(args.output_dir)
    tr = T.LngKnpTrainer(_Recorder(), ds, LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), 16), args, log=lambda s: None)
    saved = []
    tr.save_checkpoint = lambda path, tokenizer=None: (os.makedirs(path, exist_ok=True), saved.append(os.path.basename(path)))
    tr.train()
    assert saved == ["checkpoint-2", "checkpoint-4", "checkpoint-6"]
    assert sorted(os.listdir(args.output_dir)) == ["checkpoint-4", "checkpoint-6"]


def test_training_restores_the_ctx_precision_when_a_step_raises(tmp_path):
    c = CASES["L8_smtid"]
    _write(tmp_path, c["files"])
    ds = LngKnpMarginMSEforT5SeqAQDataset(str(tmp_path / "examples.jsonl"), None, str(tmp_path / "queries"), None, True)
    rec = _Recorder()
    rec.base_model._EM.ctx.prec = "f32"

    def boom(**kw):
        raise RuntimeError("device step failed")

    rec.training_step = boom
    tr = T.LngKnpTrainer(rec, ds, LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), 16),
                         T.LngKnpTrainingArgs(output_dir=str(tmp_path / "o"), max_steps=3, bf16=True), log=lambda s: None)
    with pytest.raises(RuntimeError):
        tr.train()
    assert rec.base_model._EM.ctx.prec == "f32"
    assert T.LngKnpTrainingArgs(output_dir="x").bf16 is False     # off unless --use_fp16, like main.py and TrainingArguments


class _State:
    def __init__(self, n=6):
        self.total, self.step = n, 0
        self.exp_avg, self.exp_avg_sq = torch.zeros(n), torch.zeros(n)


def test_resume_continues_the_interrupted_run(tmp_path):
    """A run stopped after a checkpoint and resumed from it feeds the device step exactly what the uninterrupted run fed it
    from that step on (same batches: epoch shuffle + the negatives drawn from Python's RNG; same learning rates), with the
    AdamW moments and step count restored."""
    c = CASES["L8_smtid"]
    _write(tmp_path, c["files"])
    ds = LngKnpMarginMSEforT5SeqAQDataset(str(tmp_path / "examples.jsonl"), None, str(tmp_path / "queries"), None, True)
    coll = LngKnpMarginMSEforT5SeqAQCollator(WordTokenizer(), 16)

    def make(out, max_steps):
        rec = _Recorder()
        rec._st = _State()
        rec.train_state = lambda: rec._st
        rec.base_model.save_pretrained = lambda path: None
        step = rec.training_step

        def stepping(**kw):                      # the moments move with every step, like the device optimizer's
            rec._st.step += 1
            rec._st.exp_avg += 1.0
            rec._st.exp_avg_sq += 0.5
            d = step(**kw)
            rec.calls[-1]["neg"] = kw["neg_doc_encoding"].tolist()
            return d

        rec.training_step = stepping
        args = T.LngKnpTrainingArgs(output_dir=str(tmp_path / out), per_device_train_batch_size=2, max_steps=max_steps, save_steps=4,
                                    logging_steps=3, warmup_ratio=0.25)
        os.makedirs(args.output_dir, exist_ok=True)
        return rec, T.LngKnpTrainer(rec, ds, coll, args, log=lambda s: None)

    import ripor_amd.modeling.t5_generative_retriever as M
    real = M.expected_keys
    M.expected_keys = lambda cfg: ["w"]
    _Recorder._Base.config = None
    try:
        full, tr_full = make("full", 10)
        tr_full.train()
        part, tr_part = make("part", 10)
        tr_part.max_steps = 5                      # the run dies after step 5; its last checkpoint is checkpoint-4
        tr_part.train()
        ck = str(tmp_path / "part" / "checkpoint-4")
        assert sorted(os.listdir(ck)) == ["optimizer.pt", "rng_state.pth", "trainer_state.json"]
        res, tr_res = make("part", 10)
        tr_res.train(resume_from_checkpoint=ck)
    finally:
        M.expected_keys = real
    assert tr_res.global_step == 10 and len(res.calls) == 6
    assert res.calls == full.calls[4:]             # steps 4..9: same lr, same examples, same sampled negatives
    assert res._st.step == full._st.step == 10 and torch.equal(res._st.exp_avg, full._st.exp_avg)
    assert [h["step"] for h in tr_res.history] == [h["step"] for h in tr_full.history]
    with pytest.raises(ValueError):                # another batch size: the data order would not continue
        r2, t2 = make("part", 10)
        t2.steps_per_epoch += 1
        t2.train(resume_from_checkpoint=ck)
    # a distributed run keeps one RNG file per rank (HF Trainer's rng_state_<rank>.pth): every rank draws its own negatives
    r3, t3 = make("part", 10)
    t3.world, t3.rank = 2, 1
    assert os.path.basename(t3._rng_file(ck)) == "rng_state_1.pth"
    t3.save_rng_state(ck)
    assert "rng_state_1.pth" in os.listdir(ck) and "rng_state.pth" in os.listdir(ck)
