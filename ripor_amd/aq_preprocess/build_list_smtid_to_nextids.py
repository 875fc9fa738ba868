"""CLI-compatible replacement of reference aq_preprocess/build_list_smtid_to_nextids.py:13-41.

The reference pre-builds and pickles the per-level ``{prefix_string: [next ids]}`` dicts
(``list_smtid_to_nextids.pkl``). This implementation's trie is the sorted code matrix, so the cache
written next to ``docid_to_smtid.json`` is the binary ``list_smtid_to_nextids.rprtrie`` produced by
``rpr_trie_save`` (sorted codes + permutation)."""
from __future__ import annotations

import argparse
import json
import os

import numpy as np


def cache_path(docid_to_smtid_path: str) -> str:
    return os.path.join(os.path.dirname(docid_to_smtid_path), "list_smtid_to_nextids.rprtrie")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--docid_to_smtid_path", default=None, type=str)
    args = ap.parse_args(argv)
    out = cache_path(args.docid_to_smtid_path)
    from .. import engine as E
    from ..evaluate import fresh_trie_cache
    if fresh_trie_cache(args.docid_to_smtid_path) is not None:
        print(f"{out} exists")
        return
    try:
        docids, codes = E.read_docid_to_smtid(args.docid_to_smtid_path)   # streaming C++ reader
    except E.RiporHipError as e:
        print(f"rpr_d2s reader declined ({e}); falling back to json.load")
        with open(args.docid_to_smtid_path) as fin:
            docid_to_smtids = json.load(fin)
        docids = list(docid_to_smtids.keys())
        codes = np.asarray([v[1:] for v in docid_to_smtids.values()], dtype=np.int64)
    # the file records the smallest vocab that holds the codes; the model's decoder vocab size widens it at load time
    V = int(codes.max()) + 1
    for l in range(codes.shape[1]):
        print(f"{l}-th step has {len(np.unique(codes[:, :l + 1], axis=0)) if l < 3 else -1:,} effective smtid "
              f"(-1: not counted for deep levels)")
    print("save list_smtid_to_nextids")
    # host only: sorts on the CPU threads and writes sorted codes + permutation + docids; no GPU is touched
    E.build_trie_file(codes, V, out, docids=docids, source_path=args.docid_to_smtid_path)


if __name__ == "__main__":
    main()
