"""The N > 1 path on CPU: the launcher's gather of variable-length result blobs (mtr_amd.run.gather_bytes: an
all_gather of the sizes + one padded gather) with world_size 2 over gloo — the same code runs over RCCL with CUDA tensors on the
GPU box — and the sharded generation of bench.py's strong-scaling read set."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mtr_amd import synth
from mtr_amd.run import gather_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _payload(rank, rnd):
    rng = np.random.RandomState(100 * rnd + rank)
    n = 0 if (rank == 1 and rnd == 1) else int(rng.randint(1, 5000))     # a rank with nothing to send in one round
    return rng.randint(0, 256, size=n).astype(np.uint8).tobytes()


def _worker(rank, world, store, q):
    dist.init_process_group("gloo", init_method="file://" + store, rank=rank, world_size=world)     # no port picked ahead of the ranks
    for rnd in range(3):
        blobs, sizes = gather_bytes(dist, torch, _payload(rank, rnd), rank, world, torch.device("cpu"))
        assert sizes == [len(_payload(r, rnd)) for r in range(world)]
        if rank == 0:
            q.put((rnd, blobs))
        else:
            assert blobs is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_bytes_world2_gloo(tmp_path):
    world, port = 2, str(tmp_path / "store")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    items = [q.get(timeout=90) for _ in range(3)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rnd, blobs in items:
        assert blobs == [_payload(r, rnd) for r in range(world)], f"round {rnd}: bytes changed in the gather"


def test_a_rank_generates_its_block_of_the_strong_scaling_set_from_a_checkpoint():
    """bench.py --strong c4: rank r takes reads [r n / N, (r + 1) n / N) of the seeded stream and enters the generator at the nearest
    committed checkpoint; the blocks must be exactly the reads of the whole stream (the oracle's known answer is over that stream)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "c4_rng_checkpoints.npz"))
    ck = (z["idx"], z["keys"], z["pos"])
    assert ck[0][0] == 0 and len(ck[0]) == 20 and ck[0][1] == 5000
    whole = synth.make_reads("c4", 5200, 4)
    for lo, hi in ((0, 7), (4990, 5010), (5000, 5200)):
        part = synth.make_reads_range("c4", lo, hi, 4, ck)
        assert [i for i, _ in part] == [str(i) for i in range(lo, hi)]
        assert all(np.array_equal(a[1], b[1]) for a, b in zip(whole[lo:hi], part))
    assert all(np.array_equal(a[1], b[1]) for a, b in zip(whole[3:9], synth.make_reads_range("c4", 3, 9, 4, None)))
