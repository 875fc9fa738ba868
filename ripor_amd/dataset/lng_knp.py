"""Data side of the prefix-oriented ranking fine-tune: the examples file written by the
training-data generation pass -> batches in the layout ``T5SeqAQEncoderForLngKnpMarginMSE`` consumes.

Same class names, constructor arguments, item tuples and batch keys as the reference
(``LngKnpMarginMSEforT5SeqAQDataset`` dataset/dataset.py:418-525, ``LngKnpMarginMSEforT5SeqAQCollator``
dataset/data_collator.py:11-88, ``CollectionDatasetPreLoad`` dataset/dataset.py:266-332), so ``main.py``'s
``t5seq_aq_encoder_lng_knp_margin_mse`` branch reads the same files:

* examples: one JSON object per line — ``qid``, ``smtids`` (or ``docids``), ``scores`` and, by smtid length 8 / 16 / 32,
  ``smtid_4_scores`` [, ``smtid_8_scores`` [, ``smtid_16_scores``]]; entry 0 is the positive, a negative is drawn with
  ``random.sample(range(1, n), k=1)`` per item like the reference (so Python's ``random`` seed fixes the epoch);
* doc encoding = smtid[1:], decoder_input_ids = smtid[:-1] with smtid[0] = -1 (dataset.py:488-500);
* the positive and the negative example of a row carry the same query text (``"query: " + text``).
"""
from __future__ import annotations

import json
import os
import random
from typing import Dict, List, Optional

import torch

_SUB = {8: [4], 16: [4, 8], 32: [4, 8, 16]}


class CollectionDatasetPreLoad:
    """``raw.tsv`` (``id\\ttext``) held in memory (reference dataset/dataset.py:266-332). ``id_style="content_id"``:
    ``ds[str(id)] -> (id, text)``; ``"row_id"``: ``ds[i] -> (id, text)``."""

    def __init__(self, data_dir: str, id_style: str = "content_id"):
        assert id_style in ("row_id", "content_id"), "provide valid id_style"
        self.id_style = id_style
        self.data_dict: Dict = {}
        self.line_dict: Dict = {}
        with open(os.path.join(data_dir, "raw.tsv")) as reader:
            for i, line in enumerate(reader):
                if len(line) > 1:
                    id_, *data = line.split("\t")
                    text = " ".join(" ".join(data).splitlines())
                    if id_style == "row_id":
                        self.data_dict[i] = text
                        self.line_dict[i] = id_.strip()
                    else:
                        self.data_dict[id_] = text.strip()
        self.nb_ex = len(self.data_dict)

    def __len__(self):
        return self.nb_ex

    def __getitem__(self, idx):
        if self.id_style == "row_id":
            return self.line_dict[idx], self.data_dict[idx]
        return str(idx), self.data_dict[str(idx)]


class LngKnpMarginMSEforT5SeqAQDataset(torch.utils.data.Dataset):
    def __init__(self, dataset_path, document_dir, query_dir, docid_to_smtid_path, smtid_as_docid=False):
        # the reference also preloads the document collection (never read by __getitem__): only if a directory is given
        self.document_dataset = CollectionDatasetPreLoad(document_dir, id_style="content_id") if document_dir else None
        self.query_dataset = CollectionDatasetPreLoad(query_dir, id_style="content_id")
        self.examples = []
        with open(dataset_path) as fin:
            for line in fin:
                if line.strip():
                    self.examples.append(json.loads(line))
        self.smtid_as_docid = smtid_as_docid
        if smtid_as_docid:
            assert docid_to_smtid_path is None
            self.docid_to_smtid = None
        elif docid_to_smtid_path is not None:
            with open(docid_to_smtid_path) as fin:
                self.docid_to_smtid = json.load(fin)
            first = next(iter(self.docid_to_smtid.values()))
            assert first[0] == -1, first
        else:
            self.docid_to_smtid = None
        ex = self.examples[0]
        smtid_len = len(ex["smtids"][0].split("_"))
        if smtid_len not in _SUB:
            raise ValueError("not valid smtid_len = {}".format(smtid_len))
        print("smtid_len is {}".format(smtid_len))
        for k in (4, 8, 16):   # exactly the sub-lengths below the smtid length are present (dataset.py:447-459)
            assert (f"smtid_{k}_scores" in ex) == (k in _SUB[smtid_len]), (k, smtid_len)
        self.smtid_len = smtid_len

    def __len__(self):
        return len(self.examples)

    def __getitem__(self, idx):
        ex = self.examples[idx]
        key = "smtids" if self.smtid_as_docid else "docids"
        positive = ex[key][0]
        neg_idx = random.sample(range(1, len(ex[key])), k=1)[0]
        negative = ex[key][neg_idx]
        q = self.query_dataset[str(ex["qid"])][1]
        if self.smtid_as_docid:
            pos = [-1] + [int(x) for x in positive.split("_")]
            neg = [-1] + [int(x) for x in negative.split("_")]
        else:
            pos, neg = self.docid_to_smtid[str(positive)], self.docid_to_smtid[str(negative)]
        text = "query: " + q.strip()
        item = [text, text, pos[1:], neg[1:], ex["scores"][0], ex["scores"][neg_idx], pos[:-1], neg[:-1]]
        for k in _SUB[self.smtid_len]:
            item += [ex[f"smtid_{k}_scores"][0], ex[f"smtid_{k}_scores"][neg_idx]]
        return tuple(item)


class LngKnpMarginMSEforT5SeqAQCollator:
    """Tuples of the dataset above -> the ``forward(**inputs)`` dict (reference data_collator.py:11-88). ``tokenizer_type``:
    a checkpoint directory / model name for ``AutoTokenizer.from_pretrained``, or a tokenizer object."""

    def __init__(self, tokenizer_type, max_length):
        self.max_length = max_length
        if isinstance(tokenizer_type, str):
            from transformers import AutoTokenizer
            self.tokenizer = AutoTokenizer.from_pretrained(tokenizer_type)
        else:
            self.tokenizer = tokenizer_type

    def __call__(self, batch):
        n = len(batch[0])
        if n not in (10, 12, 14):
            raise ValueError("element in batch don't have deisred len since its length is {}".format(n))
        cols = [list(x) for x in zip(*batch)]
        q_pos, q_neg, pos_enc, neg_enc, s_pos, s_neg, pos_dec, neg_dec = cols[:8]
        tok = dict(add_special_tokens=True, padding="longest", truncation="longest_first", max_length=self.max_length,
                   return_attention_mask=True, return_tensors="pt")
        q_pos, q_neg = self.tokenizer(q_pos, **tok), self.tokenizer(q_neg, **tok)
        q_pos["decoder_input_ids"] = torch.LongTensor(pos_dec)
        q_neg["decoder_input_ids"] = torch.LongTensor(neg_dec)
        out = {"pos_tokenized_query": q_pos, "neg_tokenized_query": q_neg,
               "pos_doc_encoding": torch.LongTensor(pos_enc), "neg_doc_encoding": torch.LongTensor(neg_enc),
               "teacher_pos_scores": torch.FloatTensor(s_pos), "teacher_neg_scores": torch.FloatTensor(s_neg)}
        for j, k in enumerate((4, 8, 16)[: (n - 8) // 2]):
            out[f"smtid_{k}_teacher_pos_scores"] = torch.FloatTensor(cols[8 + 2 * j])
            out[f"smtid_{k}_teacher_neg_scores"] = torch.FloatTensor(cols[9 + 2 * j])
        return out
