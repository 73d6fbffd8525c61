"""The N > 1 path on CPU: read sharding and the gather of per-read record tables, world_size 2 over gloo
(the same mtr_amd.dist code runs over RCCL with CUDA tensors on the GPU box)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mtr_amd.dist import RECORD_BYTES, gather_records, shard_bounds


def test_shard_bounds_balanced_and_contiguous():
    rng = np.random.RandomState(0)
    lens = rng.randint(500, 5000, size=1000)
    for world in (1, 2, 3, 8):
        b = shard_bounds(lens, world)
        assert b[0] == 0 and b[-1] == len(lens) and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        tot = [int(lens[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert max(tot) - min(tot) <= 2 * lens.max()
    assert shard_bounds([10], 4) == [0, 0, 0, 0, 1] or shard_bounds([10], 4)[-1] == 1
    assert shard_bounds([], 2) == [0, 0, 0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.RandomState(100 + rank)
    n_reads = 5 + 3 * rank                                   # ragged: different read and record counts per rank
    counts = rng.randint(0, 4, size=n_reads).astype(np.int32)
    if rank == 1:
        counts[:] = 0                                        # a rank without any record
    n_rec = int(counts.sum())
    recs = rng.randint(0, 256, size=n_rec * RECORD_BYTES).astype(np.uint8)
    out = gather_records(torch.from_numpy(recs), torch.from_numpy(counts), dst=0)
    if rank == 0:
        r, c = out
        q.put([(x.numpy().tobytes(), y.numpy().tolist()) for x, y in zip(r, c)])
    else:
        assert out is None
    q.put((rank, recs.tobytes(), counts.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_records_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=90) for _ in range(world + 1)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gathered = [x for x in items if isinstance(x, list)][0]
    sent = {x[0]: (x[1], x[2]) for x in items if isinstance(x, tuple)}
    for r in range(world):
        assert gathered[r][0] == sent[r][0], f"record bytes of rank {r} changed in the gather"
        assert gathered[r][1] == sent[r][1]
