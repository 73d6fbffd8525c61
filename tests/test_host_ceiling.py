"""The host pipeline alone (CPU): what `mTR -g 8` sustains when the GPUs cost nothing - tests/null_engine.c answers every batch at once with two records
per read -, and the report line's "%f" without printf.  VERDICT r5 item 5: eight GPUs at the headline rate need 2.2 M reads/s from ONE process that cuts,
parses, packs (handle_one_file.c:201-293), unpacks, chains, formats and writes (chaining.cpp:243-363) for all of them."""
import ctypes as C
import os
import random

import numpy as np
import pytest

from mtr_amd import synth
from tests import host_util


@pytest.fixture(scope="module")
def fasta_200k(tmp_path_factory):
    reads = [c for _, c in synth.make_reads("headline2k", 2000, 2)]
    d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else str(tmp_path_factory.mktemp("ceil"))
    one = os.path.join(d, f"mtr_ceiling_{os.getpid()}_2k.fa")
    synth.write_fasta(one, [(str(i), reads[i]) for i in range(len(reads))])
    big = os.path.join(d, f"mtr_ceiling_{os.getpid()}_200k.fa")
    blob = open(one, "rb").read()
    with open(big, "wb") as f:
        for _ in range(100):
            f.write(blob)
    os.unlink(one)
    yield big, 200000
    os.unlink(big)


def test_host_pipeline_rate_behind_a_null_engine(fasta_200k):
    path, n = fasta_200k
    got = host_util.host_ceiling(path, n, n_gpus=8, records=2, repeats=3)
    assert got["lines"] == 2 * n                                  # every read's two records chained and printed
    # [measured, 8 vCPU container] 1.1-1.5 M reads/s at 4.3 us of CPU per read (round 5: 238 k reads/s through the replay engine); the floors
    # leave a factor of three for a busy machine
    assert got["reads_per_s_per_core_used"] >= 60e3, got
    assert got["reads_per_s"] >= min(os.cpu_count() or 1, 8) * 40e3, got
    parse_only = host_util.host_ceiling(path, n, n_gpus=8, records=0, repeats=2)
    assert parse_only["lines"] == 0
    assert parse_only["reads_per_s"] >= got["reads_per_s"] * 0.8, (parse_only, got)


def test_one_gpu_and_eight_give_the_same_report(fasta_200k, tmp_path):
    path, _ = fasta_200k
    env = dict(os.environ, MTR_LIB=host_util.build_null())
    exe = host_util.build_host()
    import subprocess
    outs = []
    for g in ("1", "3", "8"):
        outs.append(subprocess.run([exe, "-g", g, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, check=True).stdout)
    plain = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, check=True).stdout
    assert outs[0] == outs[1] == outs[2] == plain
    first = plain.split(b"\n", 1)[0].split(b"\t")
    assert first[0] == b"0" and len(first) == 13 and first[8] == b"0.900000"


def test_ratio_is_printed_as_printf_prints_it():
    lib = C.CDLL(os.path.join(host_util.HOST, "libmtr_host.so"))
    host_util.build_host()
    buf = C.create_string_buffer(64)
    libc = C.CDLL(None)
    libc.snprintf.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_double]
    ref = C.create_string_buffer(64)

    def check(m, l):
        k = lib.mtrh_format_ratio(m, l, buf)
        with np.errstate(all="ignore"):
            f = np.float32(m) / np.float32(l)              # the reference's float division (chaining.cpp:136)
        libc.snprintf(ref, 64, b"%f", float(f))
        assert buf.value == ref.value and k == len(buf.value), (m, l, buf.value, ref.value)

    for l in range(1, 300):
        for m in range(0, l + 1):
            check(m, l)
    rng = random.Random(6)
    for _ in range(100000):
        l = rng.randint(1, 1000000)
        check(rng.randint(0, l), l)
    for _ in range(20000):
        l = rng.randint(1, 100000)
        check(rng.randint(0, 40 * l), l)
    for m, l in ((5, 0), (0, 0), (-5, 3), (16777216, 1), (2 ** 31 - 1, 1), (1, 2 ** 31 - 1), (3, 7), (1, 3), (2, 3), (1, 1)):
        check(m, l)
