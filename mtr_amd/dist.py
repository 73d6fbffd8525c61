"""Multi-GPU plumbing (one process per GPU, torch.distributed: "nccl" = RCCL over xGMI on the GPU box, "gloo" in
CPU tests).  The hot path shards with no data-path collective: reads are independent under isolated semantics
(SURVEY.md §8e), so each rank runs the kernels on its own contiguous block of reads.  The only exchange is the
gather of the per-read record tables to rank 0, which then chains and prints in input order.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 2560          # sizeof(mtr_record), include/mtr_hip.h


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


def gather_records(local_records: torch.Tensor, local_counts: torch.Tensor, dst: int = 0):
    """Gather variable-length record tables to `dst`.

    local_records: uint8 [n_local_records * RECORD_BYTES] on the rank's device (CUDA for nccl, CPU for gloo);
    local_counts:  int32 [n_local_reads] on the same device.
    Returns on dst: (list of per-rank uint8 record tensors, list of per-rank int32 count tensors); None elsewhere.
    Two small all_gathers carry the sizes, then one padded gather per payload (fixed-size collectives only).
    """
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = local_records.device
    sizes = torch.tensor([local_records.numel(), local_counts.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    max_rec = int(max(int(s[0]) for s in all_sizes))
    max_cnt = int(max(int(s[1]) for s in all_sizes))
    rec_pad = torch.zeros(max(max_rec, 1), dtype=torch.uint8, device=dev)
    rec_pad[: local_records.numel()] = local_records
    cnt_pad = torch.zeros(max(max_cnt, 1), dtype=torch.int32, device=dev)
    cnt_pad[: local_counts.numel()] = local_counts
    rec_list = [torch.zeros_like(rec_pad) for _ in range(world)] if rank == dst else None
    cnt_list = [torch.zeros_like(cnt_pad) for _ in range(world)] if rank == dst else None
    dist.gather(rec_pad, rec_list, dst=dst)
    dist.gather(cnt_pad, cnt_list, dst=dst)
    if rank != dst:
        return None
    recs = [rec_list[r][: int(all_sizes[r][0])] for r in range(world)]
    cnts = [cnt_list[r][: int(all_sizes[r][1])] for r in range(world)]
    return recs, cnts
