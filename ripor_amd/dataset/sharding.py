"""Query sharding across ranks, identical to the reference's
``DistributedSampler(dataset, shuffle=False)`` (t5_pretrainer/evaluate.py:468): the index list is
padded to a multiple of the world size by wrapping around to its start, and rank r takes
``indices[r::world]``. Duplicates introduced by the padding are harmless: results are merged by
qid (evaluate.py:508-515)."""
from __future__ import annotations

import math
from typing import List


def shard_indices(n: int, world_size: int, rank: int) -> List[int]:
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} out of range for world size {world_size}")
    if n == 0:
        return []
    total = math.ceil(n / world_size) * world_size
    idx = list(range(n))
    pad = total - n
    if pad > 0:
        if pad <= n:
            idx += idx[:pad]
        else:
            idx += (idx * math.ceil(pad / n))[:pad]
    return idx[rank:total:world_size]
