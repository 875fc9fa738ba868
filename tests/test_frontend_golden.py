"""not-gpu: the query front-end and ``truncate_run`` pinned to what the reference's OWN classes produced.
tests/golden/c8_frontend.npz was written by tests/golden/make_golden.py running, unmodified and in place,
/root/reference/t5_pretrainer/dataset/dataset.py:266-332 (CollectionDatasetWithDocIDPreLoad, id_style="row_id", add_prefix=True,
is_query=True — how evaluate.py:461-462 builds it), dataset/dataloader.py:62-79 (CollectionDataWithDocIDLoader, max_length 256,
DistributedSampler(shuffle=False) — evaluate.py:463-468) and utils/metrics.py:9-15 (truncate_run) on a query file with tabs
inside the text, CRLF line ends, unicode line separators, a query longer than 256 tokens, an empty text and an empty last line,
tokenised by a SentencePiece model trained offline (stored in the fixture as data).

Here: ripor_amd.evaluate.QueryCollection / query_batches over ripor_amd.dataset.sharding.shard_indices must yield the same
ids, texts, token ids, masks and batch boundaries for every (world, rank, batch size) recorded, and
ripor_amd.utils.metrics.truncate_run the same dictionaries in the same key order. tests/test_gpu_cli.py feeds the same
fixture through the CLI."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from ripor_amd import evaluate as EV
from ripor_amd.dataset.sharding import shard_indices
from ripor_amd.utils import metrics as MT


def frontend_fixture(root):
    """Materialise the fixture's checkpoint dir (tokenizer only) and query dir under root -> (ckpt, qdir, results, runs)."""
    z = np.load(os.path.join(GOLDEN_DIR, "c8_frontend.npz"), allow_pickle=False)
    ckpt, qdir = os.path.join(root, "checkpoint"), os.path.join(root, "msmarco_frontend", "dev_queries")
    os.makedirs(ckpt)
    os.makedirs(qdir)
    with open(os.path.join(ckpt, "spiece.model"), "wb") as f:
        f.write(z["spiece_model"].tobytes())
    with open(os.path.join(ckpt, "tokenizer_config.json"), "w") as f:
        f.write(str(z["tokenizer_config"]))
    with open(os.path.join(qdir, "raw.tsv"), "w", newline="") as f:
        f.write(str(z["raw_tsv"]))
    return ckpt, qdir, json.loads(str(z["results"])), json.loads(str(z["runs"]))


def test_query_collection_equals_the_reference_dataset(tmp_path):
    _, qdir, res, _ = frontend_fixture(str(tmp_path))
    coll = EV.QueryCollection(qdir)
    items = res["items"]                       # [id, "query: " + text, [-1]] per row, from the reference's __getitem__
    assert len(coll) == len(items) == 11
    assert coll.ids == [it[0] for it in items]
    assert coll.texts == [it[1] for it in items]
    assert all(it[2] == [-1] for it in items)
    # the cases the file was written for
    assert coll.texts[1].endswith("with a tab and another") and "\t" not in coll.texts[1]
    assert coll.texts[2].startswith("query:   ") and coll.texts[2].endswith("  ") and "\r" not in coll.texts[2]
    assert "\x0b" not in coll.texts[3] and " " not in coll.texts[3]
    assert coll.ids[5] == "300674" and coll.texts[10] == "query: "


def test_query_batches_equal_the_reference_loader(tmp_path):
    from transformers import AutoTokenizer
    ckpt, qdir, res, _ = frontend_fixture(str(tmp_path))
    tok = AutoTokenizer.from_pretrained(ckpt)
    coll = EV.QueryCollection(qdir)
    seen_truncation = False
    for key, want in res["loaders"].items():
        W, r, bs = (int(x.lstrip("wrbs")) for x in key.split("_"))
        got = list(EV.query_batches(coll, tok, shard_indices(len(coll), W, r), bs, 256))
        assert len(got) == len(want), key
        for g, w in zip(got, want):
            assert set(g) == set(w) == {"input_ids", "attention_mask", "decoder_input_ids", "id"}, key
            for k in w:
                assert g[k].tolist() == w[k], (key, k)
            assert g["id"].dtype == g["decoder_input_ids"].dtype and str(g["id"].dtype) == "torch.int64"
            seen_truncation |= len(w["input_ids"][0]) == 256
    assert seen_truncation, "the fixture's long query must hit max_length"


def test_truncate_run_equals_the_reference(tmp_path):
    _, _, res, runs = frontend_fixture(str(tmp_path))
    for k, want in res["truncate"].items():
        got = MT.truncate_run(runs, int(k))
        assert got == want
        # key order inside a query is part of the contract (a tie at the cut keeps run-file order)
        assert {q: [[d, s] for d, s in v.items()] for q, v in got.items()} == res["truncate_order"][k]
    assert list(MT.truncate_run(runs, 2)["q_tie_at_cut"]) == ["d2", "d1"]
    assert list(MT.truncate_run(runs, 2)["q_all_equal"]) == ["b", "a"]
