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


# ---- several GPUs in ONE process: the C host (mTR -g N, mtr_amd/host/multi.c) and the RCCL gather of the C-ABI (mtr_gather_*) ------------
def _mtr_g(args, env_extra=None):
    exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
    subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "mTR"], check=True)
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "MTR_REPLAY_TABLE", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.update(env_extra or {})
    return subprocess.run([exe, *args], capture_output=True, env=env, timeout=900, cwd=ROOT)


def _gather_line(p):
    lines = [ln for ln in p.stderr.decode().splitlines() if "\tgather " in ln]
    assert len(lines) == 1, p.stderr.decode()[-800:]
    return lines[0]


def test_gather_api_moves_the_wire_form_through_rccl(eng):
    """mtr_gather_stage + mtr_gather_exchange (ncclCommInitAll, ncclSend / ncclRecv in one group, one copy to pinned host memory) against
    mtr_fetch_results_packed of the same batches.  One GPU here: with MTR_GATHER_SELF=1 the first GPU's own tables go through ncclSend / ncclRecv to
    itself, so RCCL's point-to-point path runs on the one-GPU box; without it they take the local copy.  Several batches per exchange, slots reused."""
    import ctypes as C
    lib = eng.lib
    n = C.c_int32()
    assert lib.mtr_device_count(C.byref(n)) == 0 and n.value >= 1
    lib.mtr_gather_last_error.restype = C.c_char_p
    batches = [[c for _, c in synth.make_reads("c4", 300, 70 + k)] for k in range(3)]
    want = []
    for b in batches:
        eng.process(b)
        want.append(eng.fetch_packed())
    for self_send in ("1", "0", "early"):
        os.environ["MTR_GATHER_SELF"] = "1" if self_send == "early" else self_send
        try:
            g = C.c_void_p()
            dev = (C.c_int32 * 1)(0)
            st = lib.mtr_gather_create(1, dev, C.byref(g))
            assert st == 0, lib.mtr_gather_last_error(g)
        finally:
            del os.environ["MTR_GATHER_SELF"]
        # RCCL comes up in the background; "early" does not wait: its first exchange most likely finds RCCL not up and copies straight to the host
        if self_send != "early":
            assert lib.mtr_gather_wait_ready(g) == 0, lib.mtr_gather_last_error(g)
        e2 = mtr_amd.Engine()
        for rounds in range(2):                                    # the second round reuses the staging slots and the buffers
            tickets = []
            for b, (data, counts) in zip(batches, want):
                e = eng if len(tickets) % 2 == 0 else e2
                e.upload(b); e.run()
                cnt = (C.c_int32 * len(b))()
                total, nbytes, ticket = C.c_int64(), C.c_int64(), C.c_int32()
                assert lib.mtr_gather_stage(g, 0, e.h, cnt, C.byref(total), C.byref(nbytes), C.byref(ticket)) == 0
                assert list(cnt) == counts.tolist() and nbytes.value == len(data) and total.value == int(counts.sum())
                tickets.append(ticket.value)
            assert len(set(tickets)) == len(tickets)
            ptrs = (C.c_void_p * len(tickets))()
            sizes = (C.c_int64 * len(tickets))()
            tk = (C.c_int32 * len(tickets))(*tickets)
            st = lib.mtr_gather_exchange(g, len(tickets), tk, ptrs, sizes)
            assert st == 0, lib.mtr_gather_last_error(g)
            for i, (data, _) in enumerate(want):
                assert sizes[i] == len(data) and C.string_at(ptrs[i], sizes[i]) == data, (self_send, rounds, i)
            assert lib.mtr_gather_exchange(g, len(tickets), tk, ptrs, sizes) != 0          # the tickets were released
        e2.close()
        st6 = (C.c_int64 * 6)()
        assert lib.mtr_gather_wait_ready(g) == 0
        assert lib.mtr_gather_get_stats(g, st6, 6) == 0 and st6[0] + st6[1] == 2 and st6[5] == 1 and st6[4] > 0
        if self_send == "1":
            assert st6[0] == 2 and st6[2] == 2 * sum(len(d) for d, _ in want)        # both exchanges over RCCL (to the GPU itself)
        if self_send == "0":
            assert st6[0] == 0 and st6[3] == 2 * sum(len(d) for d, _ in want)        # the first GPU's own tables take the local copy
        lib.mtr_gather_destroy(g)
    # a device given twice: RCCL cannot be had (the reason is kept), the exchange still delivers - straight to the host
    g = C.c_void_p()
    dev = (C.c_int32 * 2)(0, 0)
    assert lib.mtr_gather_create(2, dev, C.byref(g)) == 0
    assert lib.mtr_gather_wait_ready(g) != 0 and b"two ranks" in lib.mtr_gather_last_error(g)
    eng.upload(batches[0]); eng.run()
    cnt = (C.c_int32 * len(batches[0]))()
    total, nbytes, ticket = C.c_int64(), C.c_int64(), C.c_int32()
    assert lib.mtr_gather_stage(g, 1, eng.h, cnt, C.byref(total), C.byref(nbytes), C.byref(ticket)) == 0
    ptrs, sizes, tk = (C.c_void_p * 1)(), (C.c_int64 * 1)(), (C.c_int32 * 1)(ticket.value)
    assert lib.mtr_gather_exchange(g, 1, tk, ptrs, sizes) == 0 and C.string_at(ptrs[0], sizes[0]) == want[0][0]
    lib.mtr_gather_destroy(g)


@pytest.mark.timeout(1800)
def test_c_host_over_gpus_config4_and_config5():
    """mTR -g N, the C host of the multi-GPU path (one process, a run per GPU, RCCL called directly): BASELINE config 4 (one file) and config 5 (the 15
    bundled files, -p) with as many GPUs as are visible through RCCL (on a one-GPU box: -g 1 with the tables sent to the GPU itself through
    ncclSend / ncclRecv), and with 2 and 4 runs sharing the card (every run's tables fetched to the host) - stdout byte-identical to the goldens"""
    import torch
    n = torch.cuda.device_count()
    files = [gu.input_path(nm) for nm in BUNDLED]
    for g, extra, mode in [(n, {"MTR_GATHER": "rccl", "MTR_GATHER_SELF": "1"}, "rccl"), (2 * n, {}, "host"), (4 * n, {}, "host")]:
        p = _mtr_g(["-c", "-g", str(g), gu.input_path("synth_c4")], dict(extra, MTR_CHUNK_BYTES="20000"))
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == golden("synth_c4", "default"), (g, mode)
        line = _gather_line(p)
        assert line.startswith(f"{g} GPUs\tgather {mode}, "), line
        if mode == "rccl":
            assert ", 0 exchange(s) over RCCL" not in line and "+ 0 straight to the host" in line, line
        p = _mtr_g(["-p", "-c", "-g", str(g), *files], extra)
        assert p.returncode == 0, p.stderr.decode()[-800:]
        assert p.stdout == b"".join(golden(nm, "p") for nm in BUNDLED), (g, mode)
        assert f"gather {mode}" in _gather_line(p)


@pytest.mark.timeout(1800)
def test_c_host_over_gpus_alignments_file_order_and_a_failing_read(tmp_path):
    import torch
    n = torch.cuda.device_count()
    p = _mtr_g(["-a", "-g", str(2 * n), gu.input_path("synth_c2")], {"MTR_CHUNK_BYTES": "30000"})
    assert p.returncode == 0 and p.stdout == golden("synth_c2", "a"), p.stderr.decode()[-800:]
    from tests.oracle_binding import ORACLE_DIR
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    fa = os.path.join(gu.GOLDEN, "file_order", "mixed_lengths.fa")
    want = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), "-B", fa], capture_output=True, check=True).stdout
    for g, extra in ((n, {"MTR_GATHER": "rccl", "MTR_GATHER_SELF": "1"}), (3 * n, {})):
        p = _mtr_g(["-B", "-g", str(g), fa], dict(extra, MTR_CHUNK_BYTES="15000"))
        assert p.returncode == 0 and p.stdout == want, (g, p.stderr.decode()[-800:])
    # a read whose DP exceeds the (lowered) WrapDPsize, on a GPU other than the first: the reads before it are printed, then the reference's message
    rng = np.random.RandomState(64)
    reads = [synth.make_read(rng, 12, 14, 100, 100)[0] for _ in range(12)] + [c for _, c in synth.make_reads("headline2k", 3, 65)] + [synth.make_read(rng, 5, 30, 10, 10)[0]]
    fb = tmp_path / "fails.fa"
    synth.write_fasta(str(fb), [(f"read{i}", c) for i, c in enumerate(reads)])
    env = {"MTR_TEST_WRAP_DP_SIZE": "40000"}
    o = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), str(fb)], capture_output=True, env=dict(os.environ, **env))
    assert o.returncode != 0 and b"WrapDPsize" in o.stderr and o.stdout.count(b"\n") >= 5
    for g, extra in ((n, {"MTR_GATHER": "rccl", "MTR_GATHER_SELF": "1"}), (2 * n, {})):
        p = _mtr_g(["-g", str(g), str(fb)], dict(extra, MTR_CHUNK_BYTES="2500", **env))
        assert p.returncode != 0 and b"WrapDPsize" in p.stderr, (g, p.stderr.decode()[-500:])
        assert p.stdout == o.stdout, g


@pytest.mark.timeout(1800)
def test_c_host_rccl_config4_ten_thousand_reads_against_the_oracles_hash(tmp_path):
    """10 000 config-4 reads through mTR -g <GPUs> with the RCCL gather: the record stream is not visible on stdout, so the check is the report itself -
    identical to the single-GPU command line's (whose records are checked read by read elsewhere) - plus several rounds of real size (3 MB each)"""
    import torch
    n = torch.cuda.device_count()
    fa = tmp_path / "c4_10k.fa"
    synth.write_fasta(str(fa), [(str(i), c) for i, c in enumerate(c for _, c in synth.make_reads("c4", 10000, 44))])
    one = _mtr_g([str(fa)])
    assert one.returncode == 0 and one.stdout.count(b"\n") > 10000
    p = _mtr_g(["-c", "-g", str(n), str(fa)], {"MTR_GATHER": "rccl", "MTR_GATHER_SELF": "1", "MTR_CHUNK_BYTES": str(4 << 20)})
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == one.stdout
    line = _gather_line(p)
    assert "gather rccl" in line and int(line.split(" exchange")[0].split()[-1]) >= 4 // n and "+ 0 straight to the host" in line, line


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
