"""mTR -g N on CPU: the C host for several GPUs of one node in ONE process (mtr_amd/host/multi.c - one run per GPU, the
wire-form record tables staged per GPU and gathered per round, rank 0's chaining + printing in output order), driven through the
command line with the replay engine (tests/replay_engine.c) standing in for libmtr_hip.so: it answers every read with the
records the REFERENCE produced (golden G4) and implements the gather of include/mtr_hip.h in host memory with the same tickets
and lifetimes, so stdout must equal the reference's stdout byte for byte whatever N, the chunking and the gather mode are.
(On the GPU box the same command runs on libmtr_hip.so with RCCL: tests/test_gpu_multi.py.)"""
import os
import subprocess

import pytest

from tests import golden_util as gu
from tests import host_util as hu


@pytest.fixture(scope="module")
def cli():
    hu.build_replay()
    return hu.build_host()


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    d = tmp_path_factory.mktemp("replay_multi")
    return {m: hu.write_table(str(d / f"{m}.bin"), gu.cases(m)) for m in ("default", "p")}


def _run(cli, tables, args, mode="default", **env_extra):
    env = hu.replay_env(tables[mode])
    env.update({k: str(v) for k, v in env_extra.items()})
    return subprocess.run([cli, *args], capture_output=True, env=env, timeout=300)


def _golden(name, mode="default"):
    return open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()


def _gather_line(p):
    lines = [ln for ln in p.stderr.decode().splitlines() if "\tgather " in ln]
    assert len(lines) == 1, p.stderr.decode()[-500:]
    return lines[0]


@pytest.mark.parametrize("gpus", [1, 2, 3, 8])
@pytest.mark.parametrize("chunk", [3000, 20000, 1 << 20])
def test_config4_in_small_over_n_gpus(cli, tables, gpus, chunk):
    """BASELINE config 4 in small (mixed unit lengths, one file): chunk c -> GPU c % N, round c / N; several rounds, every round one exchange"""
    p = _run(cli, tables, ["-c", "-g", str(gpus), gu.input_path("synth_c4")], MTR_CHUNK_BYTES=chunk)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert p.stdout == _golden("synth_c4")
    line = _gather_line(p)
    if gpus == 1:
        assert line.startswith("1 GPUs\tgather host, ") and "nothing to gather" in line, line
    else:
        assert line.startswith(f"{gpus} GPUs\tgather rccl, ") and ", 0 exchange(s) over RCCL" not in line and "+ 0 straight to the host" in line, line


def test_config5_files_longest_first_with_pearson(cli, tables):
    """BASELINE config 5: the bundled files of test_multiple_TRs (one read each), -p; the files go longest first to the least loaded GPU, one round,
    output in command-line order"""
    names = ["3_5", "10_50", "2_5_10_20_set", "3_10", "5_10", "20_50", "worm_chrI", "3_20"]
    want = b"".join(_golden(n, "p") for n in names)
    for gpus in (2, 3, 8):
        p = _run(cli, tables, ["-p", "-c", "-g", str(gpus), *[gu.input_path(n) for n in names]], mode="p")
        assert p.returncode == 0, p.stderr.decode()[-500:]
        assert p.stdout == want, gpus
        assert "gather rccl, 1 exchange(s) over RCCL" in _gather_line(p)


def test_alignments_and_the_host_gather(cli, tables):
    """-a: the chains are made where the batch is resident, so every GPU's tables go to the host there (no exchange); MTR_GATHER=host and a gather
    that cannot be created (no RCCL, a device given twice) take the same path - same bytes on stdout"""
    p = _run(cli, tables, ["-a", "-c", "-g", "2", gu.input_path("synth_c2")], MTR_CHUNK_BYTES=5000)
    assert p.returncode == 0 and p.stdout == _golden("synth_c2", "a")
    assert "gather host, 0 exchange(s)" in _gather_line(p)
    for extra in ({"MTR_GATHER": "host"}, {"MTR_REPLAY_DEVICES": "1"}):
        p = _run(cli, tables, ["-c", "-g", "3", gu.input_path("synth_c4")], MTR_CHUNK_BYTES=3000, **extra)
        assert p.returncode == 0 and p.stdout == _golden("synth_c4"), extra
        assert "gather host, 0 exchange(s)" in _gather_line(p), extra
    # RCCL that never comes up (missing library, ...): the staged tables go straight to the host inside the exchange, round by round
    p = _run(cli, tables, ["-c", "-g", "3", gu.input_path("synth_c4")], MTR_CHUNK_BYTES=3000, MTR_REPLAY_GATHER_FAIL=1)
    assert p.returncode == 0 and p.stdout == _golden("synth_c4")
    line = _gather_line(p)
    assert "gather rccl, 0 exchange(s) over RCCL + " in line and "+ 0 straight" not in line and "RCCL not usable" in line, line
    # one GPU with RCCL asked for by name (the rehearsal of the RCCL path on a one-GPU box)
    p = _run(cli, tables, ["-c", "-g", "1", gu.input_path("synth_c4")], MTR_CHUNK_BYTES=3000, MTR_GATHER="rccl")
    assert p.returncode == 0 and p.stdout == _golden("synth_c4") and "gather rccl, " in _gather_line(p)
    # RCCL asked for by name where it cannot be had: an error, not a silent fall-back
    p = _run(cli, tables, ["-g", "3", gu.input_path("synth_c4")], MTR_GATHER="rccl", MTR_REPLAY_GATHER_FAIL="1")
    assert p.returncode != 0 and b"MTR_GATHER=rccl" in p.stderr and p.stdout == b""


def _good_prefix(tmp_path, n_good, tail):
    src = open(gu.input_path("synth_c2")).read().split(">")[1:]
    fa = tmp_path / f"mix{n_good}_{len(tail)}.fa"
    fa.write_text("".join(">" + r for r in src[:n_good]) + tail)
    want = _golden("synth_c2").split(b"\n")
    ids = {r.split("\n", 1)[0].encode() for r in src[:n_good]}
    return str(fa), b"".join(ln + b"\n" for ln in want if ln and ln.split(b"\t")[0] in ids)


@pytest.mark.parametrize("gpus", [2, 3])
def test_input_that_ends_on_another_gpu(cli, tables, tmp_path, gpus):
    """a bad character / an empty record in a chunk some OTHER GPU owns: everything before it is reported, nothing after it, the reference's message
    and exit status (handle_one_file.c:185, :283) - whichever GPU ran what"""
    fa, want = _good_prefix(tmp_path, 9, ">bad\nACGTNACGT\n>never\n" + "ACGT" * 50 + "\n")
    p = _run(cli, tables, ["-g", str(gpus), fa], MTR_CHUNK_BYTES=2500)
    assert p.returncode != 0 and b"Invalid character: N" in p.stderr
    assert p.stdout == want and len(want) > 0
    fa, want = _good_prefix(tmp_path, 7, ">empty\n>later\n" + "ACGTACGTAC" * 30 + "\n")
    p = _run(cli, tables, ["-g", str(gpus), fa], MTR_CHUNK_BYTES=2500)
    assert p.returncode == 0 and p.stdout == want


def test_a_device_side_failure_on_one_gpu(cli, tables):
    """a DP beyond WrapDPsize (replayed: MTR_REPLAY_FAIL_AT = the read of a batch that fails) makes the reference exit inside that read with the earlier reads printed
    (wrap_around_DP.c:96-99): the failing batch hands over the reads before the failure through the host path, later batches are dropped"""
    src = gu.input_path("synth_c2")
    ok = _run(cli, tables, ["-g", "2", src], MTR_CHUNK_BYTES=4000)
    assert ok.returncode == 0 and ok.stdout == _golden("synth_c2")
    p = _run(cli, tables, ["-g", "2", src], MTR_CHUNK_BYTES=4000, MTR_REPLAY_FAIL_AT=3)
    assert p.returncode != 0 and b"WrapDPsize" in p.stderr
    assert _golden("synth_c2").startswith(p.stdout) and len(p.stdout) < len(_golden("synth_c2"))


def test_file_order_mode_over_gpus(cli, tables):
    """-B (the reference's whole-file behaviour): a GPU replays the reads before its chunks through mtr_file_state_skip; with the replay engine the
    records do not depend on the state, so this checks the plumbing: every GPU parses what it must, the output is complete and in order"""
    p = _run(cli, tables, ["-B", "-g", "3", gu.input_path("synth_c4")], MTR_CHUNK_BYTES=3000)
    assert p.returncode == 0 and p.stdout == _golden("synth_c4")


@pytest.mark.parametrize("where,n", [("device", 2), ("device", 7), ("parser", 5), ("printer", 2)])
def test_an_allocation_failure_in_one_run_ends_the_job(cli, tables, where, n):
    """MTR_TEST_FAIL_ALLOC (alloc.c) refuses the n-th allocation of a device / parser / printer thread (counted over the threads of that kind, so it hits
    one of the three runs): what is printed stays printed, the message, status 1, nobody hangs"""
    p = _run(cli, tables, ["-g", "3", gu.input_path("synth_c4")], MTR_CHUNK_BYTES=3000, MTR_TEST_FAIL_ALLOC=f"{where}:{n}")
    assert p.returncode != 0 and b"cannot allocate" in p.stderr, (p.returncode, p.stderr.decode()[-300:])
    assert _golden("synth_c4").startswith(p.stdout) and len(p.stdout) < len(_golden("synth_c4"))


def test_option_errors(cli, tables):
    p = _run(cli, tables, ["-g", "0", gu.input_path("3_5")])
    assert p.returncode != 0 and b"-g takes a number of GPUs" in p.stderr
    p = _run(cli, tables, ["-g", "2", "/nonexistent.fa"])
    assert p.returncode != 0 and p.stdout == b""


@pytest.mark.parametrize("name,chunk,nth", [("synth_c4", 3000, 2), ("synth_c3", 45000, 2), ("synth_c3", 45000, 4), ("synth_c3", 45000, 6)])
def test_a_further_context_that_runs_out_of_memory_does_not_end_the_job(cli, tables, tmp_path, name, chunk, nth):
    """six batches in flight for long reads (pipeline.c; synth_c3: 42 kb reads, one per chunk here, the file six times over) are six contexts with their own
    device buffers: one that is created but cannot launch (MTR_ERR_OOM: a card shared with other runs) is given up, the batches in flight are finished, and its
    batch runs on the first context - same stdout, status 0.  The FIRST context failing that way is fatal as before."""
    fa = tmp_path / "six_times.fa"
    fa.write_bytes(open(gu.input_path(name), "rb").read() * 6)
    want = _golden(name) * 6
    p = _run(cli, tables, [str(fa)], MTR_CHUNK_BYTES=chunk, MTR_REPLAY_OOM_CTX=nth, MTR_HOST_TIMING=1)
    assert p.returncode == 0, p.stderr.decode()[-500:]
    assert p.stdout == want
    assert "ran out of memory: going on with fewer" in p.stderr.decode()
    p = _run(cli, tables, [str(fa)], MTR_CHUNK_BYTES=chunk, MTR_REPLAY_OOM_CTX=1)
    assert p.returncode == 1 and b"replayed failure" in p.stderr
    assert want.startswith(p.stdout)
