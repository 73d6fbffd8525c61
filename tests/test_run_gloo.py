"""The multi-GPU product path on CPU: python -m mtr_amd.run with 2 and 3 ranks over gloo and the replay engine
(tests/replay_engine.c answers every read with the reference's recorded records).  What runs is the real thing apart from
the kernels: every rank cuts + parses its chunks, batches go through the C-ABI, results are serialised, gathered to rank
0 by torch.distributed and chained + printed there in input order.  stdout must equal the reference's
(handle_one_file.c:281-287, handle_one_read.c:252, test_multiple_TRs/test.sh:8-31)."""
import os
import subprocess
import sys

import pytest

from tests import golden_util as gu
from tests import host_util as hu

BUNDLED = ["3_5", "3_10", "3_20", "3_50", "5_10", "5_20", "5_50", "10_20", "10_50", "20_50", "2_5_10_20_set",
           "2_5_10_20_50_100_200_set", "worm_chrI", "worm_chrII_1", "worm_chrII_2"]      # test_multiple_TRs/test.sh order


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    hu.build_host()
    lib = hu.build_replay()
    d = tmp_path_factory.mktemp("replay")
    return lib, {m: hu.write_table(str(d / f"{m}.bin"), gu.cases(m)) for m in ("default", "p")}


def run(lib, table, world, args, extra_env=None):
    env = dict(os.environ, MTR_REPLAY_TABLE=table, **(extra_env or {}))
    return subprocess.run([sys.executable, "-m", "mtr_amd.run", "--gpus", str(world), "--backend", "gloo", "--engine-lib", lib, "--stats", *args],
                          capture_output=True, env=env, timeout=300)


def golden(name, mode):
    return open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,chunk", [(2, "20000"), (3, "7000"), (2, "0")])
def test_one_file_sharded_over_ranks_prints_the_reference_stdout(setup, world, chunk):
    """BASELINE config 4 in small: ONE multi-read file, chunks round-robin over the ranks, several gather rounds"""
    lib, tables = setup
    p = run(lib, tables["default"], world, ["--chunk-bytes", chunk, gu.input_path("synth_c4")])
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == golden("synth_c4", "default")
    stats = [l for l in p.stderr.decode().splitlines() if l.startswith("[mtr_amd.run]")][0]
    assert f"ranks={world}" in stats
    if chunk != "0":
        assert f"ranks_with_chunks={world}" in stats and "rounds=1 " not in stats     # every rank contributed, more than one round


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode,flags", [("p", ["-p"]), ("a", ["-a"]), ("default", [])])
def test_bundled_files_lpt_over_ranks(setup, mode, flags):
    """BASELINE config 5: the 15 files of test_multiple_TRs (one read each, 2.6-140 kb), -p; longest-first over the ranks,
    output in command-line order"""
    lib, tables = setup
    files = [gu.input_path(n) for n in BUNDLED]
    p = run(lib, tables["p" if mode == "p" else "default"], 2, [*flags, *files])
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == b"".join(golden(n, mode) for n in BUNDLED)
    assert "ranks_with_chunks=2" in p.stderr.decode()


@pytest.mark.timeout(600)
def test_alignments_travel_through_the_gather(setup):
    lib, tables = setup
    p = run(lib, tables["default"], 2, ["-a", "--chunk-bytes", "30000", gu.input_path("synth_c2")])
    assert p.returncode == 0, p.stderr.decode()[-800:]
    assert p.stdout == golden("synth_c2", "a")


@pytest.mark.timeout(600)
def test_a_bad_record_on_another_rank_stops_the_output_there(setup, tmp_path):
    """reads before the bad record (whichever rank ran them) are printed, nothing after it, exit status 1"""
    lib, tables = setup
    src = open(gu.input_path("synth_c4")).read().split(">")[1:]
    fa = tmp_path / "bad.fa"
    fa.write_text("".join(">" + r for r in src[:14]) + ">bad\nACGTXACGT\n" + "".join(">" + r for r in src[14:]))
    ids = {r.split("\n", 1)[0].encode() for r in src[:14]}
    want = b"".join(l + b"\n" for l in golden("synth_c4", "default").split(b"\n") if l and l.split(b"\t")[0] in ids)
    p = run(lib, tables["default"], 2, ["--chunk-bytes", "9000", str(fa)])
    assert p.returncode == 1 and b"Invalid character: X" in p.stderr
    assert p.stdout == want


def test_plan_is_the_same_on_every_rank_and_lpt_spreads_the_long_reads(setup):
    """15 reads of 2.6-140 kb over 8 ranks, no read shared: the four reads of 90-140 kb land on four different ranks"""
    import ctypes as C
    from mtr_amd import run as R
    lib = R.load_host()
    files = [gu.input_path(n) for n in BUNDLED]
    owners = []
    for rank in (0, 5):
        o = R.Opts(print_alignment=0, manhattan=1, file_order=0, device=0, min_match_ratio=0.6, rank=rank, world=8, lpt=1, chunk_bytes=0,
                   parse_threads=1, print_threads=1, engine_lib=setup[0].encode())
        paths = (C.c_char_p * len(files))(*[f.encode() for f in files])
        os.environ["MTR_REPLAY_TABLE"] = setup[1]["default"]
        h = lib.mtrh_run_start(C.byref(o), paths, len(files))
        assert h
        owners.append([lib.mtrh_run_owner(h, c) for c in range(lib.mtrh_run_n_chunks(h))])
        lib.mtrh_run_stop(h)
    assert owners[0] == owners[1] and len(owners[0]) == 15
    big = [owners[0][BUNDLED.index(n)] for n in ("2_5_10_20_50_100_200_set", "worm_chrI", "worm_chrII_1", "worm_chrII_2")]
    assert len(set(big)) == 4 and set(owners[0]) == set(range(8))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("where", ["parser:3", "device:2"])
def test_a_failed_allocation_on_a_rank_ends_every_rank(setup, where):
    """mtr_amd/host/alloc.c: a worker thread that cannot allocate reports it through the result path (no exit() inside the thread):
    the message reaches rank 0 with the round, the job ends with status 1 and no rank is left waiting in a collective"""
    lib, tables = setup
    p = run(lib, tables["default"], 2, ["--chunk-bytes", "9000", gu.input_path("synth_c4")], extra_env={"MTR_TEST_FAIL_ALLOC": where})
    assert p.returncode == 1, (p.returncode, p.stderr.decode()[-500:])
    assert b"cannot allocate" in p.stderr
    assert golden("synth_c4", "default").startswith(p.stdout)


_CHURN = r"""
import socket, sys, time
# take and drop ports of the ephemeral range as fast as the kernel hands them out, and keep a few hundred of them listening:
# what a busy box does to a port somebody picked by bind-and-close and has not bound again yet
held = []
t_end = time.time() + float(sys.argv[1])
while time.time() < t_end:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    s.listen(1)
    held.append(s)
    if len(held) > 400:
        for x in held[:200]:
            x.close()
        del held[:200]
"""


@pytest.mark.timeout(600)
def test_twenty_launches_with_a_late_rank_zero_and_port_churn(setup):
    """VERDICT r3 (the EADDRINUSE that cost the round its parity tests): 20 launches of 4 gloo ranks, rank 0 joining the process group
    3 s after the others (MTR_TEST_RDZV_DELAY), while another process churns through the ephemeral ports and other launches run beside
    this one (4 lanes of 5 consecutive launches).  The launcher rendezvous through a FileStore: there is no port to lose.  20 / 20."""
    import concurrent.futures as cf
    lib, tables = setup
    churn = subprocess.Popen([sys.executable, "-c", _CHURN, "200"])
    want = golden("synth_c4", "default")

    def lane(_):
        bad = []
        for _ in range(5):
            p = run(lib, tables["default"], 4, ["--chunk-bytes", "15000", gu.input_path("synth_c4")], extra_env={"MTR_TEST_RDZV_DELAY": "3"})
            if p.returncode != 0 or p.stdout != want:
                bad.append((p.returncode, p.stderr.decode()[-600:]))
        return bad
    try:
        with cf.ThreadPoolExecutor(4) as ex:
            bad = [b for lane_bad in ex.map(lane, range(4)) for b in lane_bad]
    finally:
        churn.kill()
        churn.wait()
    assert not bad, (len(bad), bad[:2])
