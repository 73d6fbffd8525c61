"""Multi-GPU plumbing (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" in
CPU tests).  The hot path shards with no data-path collective: reads are independent under isolated semantics
(SURVEY.md §8e), so each rank runs the kernels on its own contiguous block of reads.  The only exchange is the
gather of the per-read record tables to rank 0, which then chains and prints in input order.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


def shard_bounds(lens: Sequence[int], world: int) -> List[int]:
    """Contiguous blocks balanced by sum of lengths (cost ~ L * unit): bounds[r]..bounds[r+1] is rank r's slice."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    if world <= 1 or n == 0:
        return [0] + [n] * max(world, 1)
    csum = np.concatenate([[0], np.cumsum(lens)])
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return bounds
