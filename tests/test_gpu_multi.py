"""The boundary's new edges and the multi-GPU product path on the GPU box (-m gpu):
  * the wire form of the record table (mtr_fetch_results_packed / mtr_unpack_records) carries exactly the records of
    mtr_fetch_results; the host's own packing (mtr_upload_batch_packed) gives the records of mtr_upload_batch;
  * python -m mtr_amd.run with as many ranks as GPUs are visible, and with 2 ranks sharing the GPU (gloo moves the bytes:
    RCCL refuses two ranks on one device), on BASELINE config 4 (one multi-read file) and config 5 (the 15 files of
    test_multiple_TRs, -p): stdout byte-identical to the reference's goldens;
  * the advisor's case: a pure tandem-repeat read with a unit of 250..256 bases as the longest read of its batch."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mtr_amd
from mtr_amd import synth
from tests import golden_util as gu
from tests.test_run_gloo import BUNDLED, golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    # PyTorch brings its own HIP runtime (same soname as /opt/rocm's, which libmtr_hip.so is linked against): whichever is loaded
    # first serves both, and torch does not find the GPU through the other one.  So, as bench.py and the launcher do: torch first.
    import torch
    torch.cuda.init()
    e = mtr_amd.Engine()
    yield e
    e.close()


def test_wire_form_carries_the_records(eng):
    reads = [c for _, c in synth.make_reads("c4", 400, 31)]
    want = eng.process(reads)
    assert eng.fetch_via_wire() == want
    data, counts = eng.fetch_packed()
    assert counts.tolist() == [len(r) for r in want]
    assert len(data) == sum(56 + ((r.rep_period + 3) & ~3) + 4 * r.rep_period for per in want for r in per)
    assert len(data) < 0.45 * 2560 * int(counts.sum())          # what the wire form is for
    # a prefix of the batch (what the host fetches after a device-side failure)
    part, pc = eng.fetch_packed(limit=57)
    assert pc.tolist() == counts[:57].tolist() and data.startswith(part)


def test_wire_form_on_device_memory_for_the_gather(eng):
    import torch
    reads = [c for _, c in synth.make_reads("c2", 200, 32)]
    want = eng.process(reads)
    data, counts = eng.fetch_packed()
    buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
    c2, total, nbytes = eng.export_packed_device(buf.data_ptr(), buf.numel())
    assert c2.tolist() == counts.tolist() and total == sum(len(r) for r in want) and nbytes == len(data)
    assert buf[:nbytes].cpu().numpy().tobytes() == data
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_OVERFLOW"):
        eng.export_packed_device(buf.data_ptr(), 16)


def test_packed_upload_equals_byte_upload(eng):
    reads = [c for _, c in synth.make_reads("c4", 300, 33)] + [np.array([0, 1, 2, 3] * 3 + [2], np.uint8), np.zeros(16, np.uint8), np.full(33, 3, np.uint8)]
    want = eng.process(reads)
    eng.upload_packed(reads)
    eng.run()
    assert eng.fetch() == want


def _run(args, world, backend=None):
    cmd = [sys.executable, "-m", "mtr_amd.run", "--stats"] + (["--gpus", str(world)] if world > 1 else []) + (["--backend", backend] if backend else []) + args
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "MTR_REPLAY_TABLE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return subprocess.run(cmd, capture_output=True, env=env, timeout=900, cwd=ROOT)


def _worlds():
    import torch
    n = torch.cuda.device_count()
    return [(n, None)] + ([(2, "gloo")] if n == 1 else [])       # on a one-GPU box also 2 ranks sharing the GPU


@pytest.mark.timeout(1800)
def test_launcher_config4_one_file_over_the_visible_gpus():
    for world, backend in _worlds():
        p = _run(["--chunk-bytes", "20000", gu.input_path("synth_c4")], world, backend)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == golden("synth_c4", "default"), (world, backend)
        assert f"ranks={world}" in p.stderr.decode()


@pytest.mark.timeout(1800)
def test_launcher_config5_bundled_files_pearson():
    files = [gu.input_path(n) for n in BUNDLED]
    for world, backend in _worlds():
        p = _run(["-p", *files], world, backend)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == b"".join(golden(n, "p") for n in BUNDLED), (world, backend)


@pytest.mark.timeout(1800)
def test_launcher_alignments_and_file_order():
    for world, backend in _worlds():
        p = _run(["-a", "--chunk-bytes", "30000", gu.input_path("synth_c2")], world, backend)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == golden("synth_c2", "a"), (world, backend)
    # -B: a rank replays the reads before its chunks (mtr_file_state_skip); the oracle's -B run is the reference here
    from tests.oracle_binding import ORACLE_DIR
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    fa = os.path.join(gu.GOLDEN, "file_order", "mixed_lengths.fa")
    want = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), "-B", fa], capture_output=True, check=True).stdout
    for world, backend in _worlds():
        p = _run(["-B", "--chunk-bytes", "15000", fa], world, backend)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == want, (world, backend)


@pytest.mark.parametrize("unit_len", [250, 253, 256])
def test_pure_repeat_with_unit_250_to_256_as_longest_read(unit_len):
    """the packed two-parameter pass (units up to 256 bases) writes ONE byte per cell and is checked against the
    traceback buffer as such; reads that are one long repeat of a 250..256-base unit, the longest reads of their batch, in
    both kernel modes.  (The directional index never opens a window over more than ~85 % of such a read - a scan of 250
    shapes with the oracle found none above - so the reference's own inputs stay well inside the buffer either way.)"""
    from tests.oracle_binding import Oracle
    rng = np.random.RandomState(unit_len)
    unit = rng.randint(0, 4, unit_len).astype(np.uint8)
    fl = [rng.randint(0, 4, 40).astype(np.uint8) for _ in range(4)]
    reads = [np.concatenate([fl[0], np.tile(unit, 20)[:4920], fl[1]]), np.concatenate([fl[2], np.tile(unit, 9)[:2020], fl[3]])] + [c for _, c in synth.make_reads("c2", 6, 5)]
    orc = Oracle()
    want = [orc.process(c) for c in reads]
    orc.close()
    for staged in ("0", "1"):
        os.environ["MTR_STAGED"] = staged
        try:
            e = mtr_amd.Engine()
            got = e.process(reads)
            e.close()
        finally:
            del os.environ["MTR_STAGED"]
        assert [[tuple(r) for r in g] for g in got] == want, staged
    assert any(r[3] == unit_len for r in want[0])


def _torchrun(world, args, env, timeout):
    """python -m torch.distributed.run as the driver starts bench.py (a named master port: torchrun's agent listens on it before any
    rank starts).  The port is taken below the kernel's ephemeral range, so no outgoing connection can sit on it; should a listener of
    something else hold it, the launch is repeated on the next port instead of failing the suite on plumbing."""
    p = None
    for attempt in range(6):
        port = 29500 + (os.getpid() * 7 + attempt * 131 + _torchrun.calls * 17) % 400
        _torchrun.calls += 1
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                            "--master-port", str(port), *args], capture_output=True, env=env, timeout=timeout, cwd=ROOT)
        if p.returncode == 0 or not any(w in p.stderr for w in (b"EADDRINUSE", b"Address already in use", b"address already in use")):
            return p
    return p


_torchrun.calls = 0


@pytest.mark.timeout(900)
def test_bench_strong_scaling_gathers_the_oracles_record_stream():
    """BASELINE config 4 as bench.py measures it with N ranks: 100 000 reads in contiguous blocks, every rank's record tables
    (wire form, exported on the device) gathered to rank 0; the sha256 of the gathered stream must be the CPU oracle's
    (tests/golden/c4_100k_wire.json).  With one GPU visible two ranks share it and the exchange goes through gloo
    (MTR_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device); the driver's runs use RCCL, one GPU per rank."""
    import json
    import torch
    n = torch.cuda.device_count()
    world = n if n > 1 else 2
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if n == 1:
        env["MTR_BENCH_BACKEND"] = "gloo"
    p = _torchrun(world, [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--strong", "c4", "--steps", "1", "--warmup", "1"], env, 800)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["scaling"] == "strong"
    st = line["strong"]
    assert st["ranks_seen"] == world and sum(st["reads_per_rank"]) == 100000 and st["matches_oracle"] is True, st


# ---- RCCL on the one-GPU box: one rank under torch.distributed.run, backend "nccl"  -------------------------------
def _torchrun_one_rank(args, extra_env=None, timeout=800):
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "MTR_REPLAY_TABLE", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MTR_BENCH_BACKEND")}
    env.update(extra_env or {})
    return _torchrun(1, args, env, timeout)


@pytest.mark.timeout(900)
def test_rccl_weak_step_of_the_bench_with_one_rank():
    """bench.py's N > 1 step - init_process_group("nccl"), the size all_gather, the gather of the wire-form tables, GPU_MAX_HW_QUEUES -
    with ONE rank: what an 8-GPU run executes per rank, on the box the suite has."""
    import json
    p = _torchrun_one_rank([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--reads", "3000", "--no-cli", "--no-latency",
                            "--cpu-sample", "0"], {"MTR_BENCH_FORCE_DIST": "1"})
    assert p.returncode == 0, p.stderr.decode()[-800:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["exchange"]["backend"] == "nccl" and line["exchange"]["forced_on_one_rank"] is True
    assert "pinned host memory" in line["value_definition"] and line["value"] > 0


@pytest.mark.timeout(900)
def test_rccl_strong_c4_10000_reads_match_the_oracles_hash():
    """bench.py --strong c4 --strong-reads 10000 through RCCL with one rank: the gathered record stream's sha256 is the CPU oracle's
    (tests/golden/c4_10000_wire.json, made by tests/golden/make_c4_wire_hash.py -n 10000)."""
    import json
    p = _torchrun_one_rank([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--strong", "c4", "--strong-reads", "10000", "--steps", "1", "--warmup", "1"],
                           {"MTR_BENCH_FORCE_DIST": "1"})
    assert p.returncode == 0, p.stderr.decode()[-800:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    st = line["strong"]
    assert line["exchange"]["backend"] == "nccl" and st["ranks_seen"] == 1 and st["reads_per_rank"] == [10000]
    assert st["matches_oracle"] is True and st["records"] == 27428, st


@pytest.mark.timeout(900)
def test_rccl_launcher_with_one_rank_force_dist():
    """python -m mtr_amd.run --force-dist under torch.distributed.run with one rank: process group over RCCL, the size all_gather,
    the padded gather and the per-round verdict broadcast run on the GPU box; stdout byte-identical to the reference's."""
    p = _torchrun_one_rank(["-m", "mtr_amd.run", "--force-dist", "--stats", "--chunk-bytes", "20000", gu.input_path("synth_c4")])
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == golden("synth_c4", "default")
    err = p.stderr.decode()
    assert "ranks=1" in err and "backend=nccl" in err, err[-400:]
    files = [gu.input_path(n) for n in BUNDLED[:6]]
    p = _torchrun_one_rank(["-m", "mtr_amd.run", "--force-dist", "-a", *files])            # several files: one round, -a blobs
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == b"".join(golden(n, "a") for n in BUNDLED[:6])
