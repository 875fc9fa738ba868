"""Small helpers of the retrieve tasks (mirror of reference t5_pretrainer/utils/utils.py:13-59)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def is_first_worker() -> bool:
    return not dist.is_available() or not dist.is_initialized() or dist.get_rank() == 0


def makedir(dir_: str) -> None:
    os.makedirs(dir_, exist_ok=True)


_DATASET_RULES = (
    ("TREC_DL_2019", "TREC_DL_2019"), ("trec2020", "TREC_DL_2020"), ("TREC_DL_2020", "TREC_DL_2020"),
)


def get_dataset_name(path: str) -> str:
    """Dataset directory name derived from a query-collection / qrel path (same rules and the
    same precedence as the reference, utils/utils.py:13-35)."""
    for needle, name in _DATASET_RULES:
        if needle in path:
            return name
    if "msmarco" in path:
        return "MSMARCO_TRAIN" if "train_queries" in path else "MSMARCO"
    if "MSMarco-v2" in path:
        if "dev_1" in path:
            return "MSMARCO_v2_dev1"
        assert "dev_2" in path
        return "MSMARCO_v2_dev2"
    if "toy" in path:
        return "TOY"
    if "nq-320k" in path:
        return "NQ_320K"
    return "other_dataset"


def convert_ptsmtids_to_strsmtid(input_smtids: torch.Tensor, seq_length: int):
    """``[Q, B, L+1]`` int tensor (column 0 = start id) -> ``"c1_c2_.._cL"`` strings per beam
    (reference utils/utils.py:46-59)."""
    assert input_smtids.dim() == 3, input_smtids.dim()
    assert input_smtids.size(2) == seq_length + 1, (input_smtids.size(1), seq_length)
    rows = input_smtids.cpu().tolist()
    return [["_".join(map(str, beam[1:])) for beam in beams] for beams in rows]
