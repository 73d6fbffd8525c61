"""Host side of the boundary on CPU: mtr_amd/host (FASTA cutting + parsing, batching, wire form, chaining, printers,
plain C) driven end to end through the command line, with the replay engine (tests/replay_engine.c) standing in for
libmtr_hip.so: every read is answered with the records the REFERENCE produced for it (golden G4), so stdout must equal
the reference's stdout byte for byte."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import golden_util as gu
from tests import host_util as hu


@pytest.fixture(scope="module")
def cli():
    hu.build_replay()
    return hu.build_host()


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
    d = tmp_path_factory.mktemp("replay")
    return {m: hu.write_table(str(d / f"{m}.bin"), gu.cases(m)) for m in ("default", "p")}


def _cases():
    out = []
    for name, mode in gu.cases():
        out.append((name, mode, []))
        if mode == "default" and os.path.exists(os.path.join(gu.GOLDEN, f"{name}.a.stdout")):
            out.append((name, "a", ["-a"]))
    return out


@pytest.mark.parametrize("name,mode,flags", _cases())
def test_cli_with_replayed_records_matches_reference_stdout(cli, tables, name, mode, flags):
    env = hu.replay_env(tables["p" if mode == "p" else "default"])
    p = subprocess.run([cli, *(["-p"] if mode == "p" else []), *flags, gu.input_path(name)], capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[:500]
    assert p.stdout == open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()


def test_small_chunks_and_batches_change_nothing(cli, tables, tmp_path):
    """one multi-read file cut into many chunks (several parser threads, both contexts in turn)"""
    import sys
    env = dict(hu.replay_env(tables["default"]))
    want = open(os.path.join(gu.GOLDEN, "synth_c4.default.stdout"), "rb").read()
    for cb in ("3000", "20000", "1000000"):
        p = subprocess.run([sys.executable, "-m", "mtr_amd.run", "--engine-lib", env["MTR_LIB"], "--chunk-bytes", cb, gu.input_path("synth_c4")],
                           capture_output=True, env=env)
        assert p.returncode == 0, p.stderr.decode()[:500]
        assert p.stdout == want, cb


def _good_prefix(tmp_path, n_good, tail):
    """a FASTA with the first n_good reads of synth_c2 followed by `tail`; returns (path, the reference's stdout for the prefix)"""
    src = open(gu.input_path("synth_c2")).read().split(">")[1:]
    fa = tmp_path / f"mix{n_good}_{len(tail)}.fa"
    fa.write_text("".join(">" + r for r in src[:n_good]) + tail)
    want = open(os.path.join(gu.GOLDEN, "synth_c2.default.stdout"), "rb").read().split(b"\n")
    ids = {r.split("\n", 1)[0].encode() for r in src[:n_good]}
    keep = b"".join(l + b"\n" for l in want if l and l.split(b"\t")[0] in ids)
    return str(fa), keep


def test_reads_before_a_bad_record_are_reported_first(cli, tables, tmp_path):
    """the reference reports read after read and dies AT the bad character (handle_one_file.c:185): everything before it is on
    stdout, then the message on stderr and a failing exit status"""
    fa, want = _good_prefix(tmp_path, 5, ">bad\nACGTNACGT\n>never\nACGT\n")
    p = subprocess.run([cli, fa], capture_output=True, env=hu.replay_env(tables["default"]))
    assert p.returncode != 0 and b"Invalid character: N" in p.stderr
    assert p.stdout == want and len(want) > 0


def test_an_empty_record_ends_the_input(cli, tables, tmp_path):
    """handle_one_file.c:283: the loop stops at the first record without bases; exit status 0"""
    fa, want = _good_prefix(tmp_path, 4, ">empty\n>later\nACGTACGTACGT\n")
    p = subprocess.run([cli, fa], capture_output=True, env=hu.replay_env(tables["default"]))
    assert p.returncode == 0, p.stderr
    assert p.stdout == want


def test_reads_before_a_device_failure_are_reported_first(cli, tables, tmp_path):
    """a DP beyond WrapDPsize makes the reference exit inside that read (wrap_around_DP.c:96-99) with the earlier reads printed"""
    fa, want = _good_prefix(tmp_path, 6, "")
    fa3, want3 = _good_prefix(tmp_path, 3, "")
    env = dict(hu.replay_env(tables["default"]), MTR_REPLAY_FAIL_AT="3")
    p = subprocess.run([cli, fa], capture_output=True, env=env)
    assert p.returncode != 0 and b"WrapDPsize" in p.stderr
    assert p.stdout == want3


def test_cli_errors(cli, tables, tmp_path):
    env = hu.replay_env(tables["default"])
    r = subprocess.run([cli, "-m", "1.5", gu.input_path("3_5")], capture_output=True, env=env)
    assert r.returncode != 0 and b"must range from 0 to 1" in r.stderr
    r = subprocess.run([cli, str(tmp_path / "missing.fa")], capture_output=True, env=env)
    assert r.returncode != 0 and b"cannot open" in r.stderr
    r = subprocess.run([cli], capture_output=True, env=env)
    assert r.returncode != 0 and b"input file name is expected" in r.stderr
    r = subprocess.run([cli, "-c", gu.input_path("3_5")], capture_output=True, env=env)
    assert r.returncode == 0 and r.stderr.decode().startswith("Computation time\n") and "Count of queries" in r.stderr.decode()


def test_cli_fails_loudly_without_gpu(cli):
    """no CPU fallback: with the product library the driver must refuse to run when there is no HIP device"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from mtr_amd import build as b
    b.build()
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "MTR_REPLAY_TABLE")}
    p = subprocess.run([cli, gu.input_path("3_5")], capture_output=True, env=env)
    assert p.returncode != 0 and b"no usable HIP device" in p.stderr and p.stdout == b""


# ---- the FASTA reader against a restatement of the reference's loop -----------------------------------------------------
class Batch(C.Structure):
    pass


Batch._fields_ = [("n", C.c_int32), ("lens", C.POINTER(C.c_int32)), ("offs", C.POINTER(C.c_int64)), ("codes", C.POINTER(C.c_uint8)),
                  ("woff", C.POINTER(C.c_int64)), ("packed", C.POINTER(C.c_uint32)), ("n_words", C.c_int64),
                  ("ids", C.POINTER(C.c_void_p)), ("id_lens", C.POINTER(C.c_int32)), ("id_store", C.c_void_p),
                  ("end", C.c_int), ("bad_char", C.c_char), ("end_id", C.c_void_p), ("end_id_len", C.c_int32), ("next", C.POINTER(Batch))]


class File(C.Structure):
    _fields_ = [("path", C.c_char_p), ("map", C.c_void_p), ("size", C.c_size_t), ("fd", C.c_int)]


def reference_reader(data: bytes):
    """handle_one_file.c:201-269 + :281-287 restated: fgets windows of 4095 characters; returns ([(id, codes)], end)"""
    lut = {ord(c): v for c, v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3])}
    reads, cur, cur_id, have_header, pos = [], [], b"", False, 0
    while pos < len(data):
        w = data[pos:pos + 4095]
        nl = w.find(b"\n")
        if nl >= 0:
            w = w[:nl + 1]
        pos += len(w)
        if w[:1] == b">":
            ident = w[1:]
            for stop in (b"\0", b"\n", b"\r"):
                k = ident.find(stop)
                if k >= 0:
                    ident = ident[:k]
            if not have_header:
                have_header, cur_id = True, ident
                continue
            if not cur:
                return reads, "empty"
            reads.append((cur_id, cur))
            cur, cur_id = [], ident
            continue
        for ch in w:
            if ch in (0, 10, 13):
                break
            if ch not in lut:
                return reads, "bad:" + chr(ch)
            cur.append(lut[ch])
            if len(cur) >= 1000000:
                return reads, "toolong"
    if cur:
        reads.append((cur_id, cur))
        return reads, "eof"
    return reads, "empty"


def parse_with_host(path, n_target):
    lib = C.CDLL(os.path.join(hu.HOST, "libmtr_host.so"))
    lib.mtrh_plan_chunks.restype = C.POINTER(C.c_size_t)
    lib.mtrh_plan_chunks.argtypes = [C.POINTER(File), C.c_int, C.POINTER(C.c_int)]
    lib.mtrh_parse_chunk.restype = C.POINTER(Batch)
    lib.mtrh_parse_chunk.argtypes = [C.POINTER(File), C.c_size_t, C.c_size_t, C.c_int, C.c_int64]
    lib.mtrh_batch_free.argtypes = [C.POINTER(Batch)]
    f = File()
    assert lib.mtrh_file_open(C.byref(f), path.encode()) == 0
    nc = C.c_int()
    off = lib.mtrh_plan_chunks(C.byref(f), n_target, C.byref(nc))
    reads, end = [], "eof"
    for c in range(nc.value):
        head = lib.mtrh_parse_chunk(C.byref(f), off[c], off[c + 1], 3, 1 << 40)      # tiny batches: 3 reads each
        b = head
        while b:
            bb = b.contents
            for i in range(bb.n):
                codes = np.ctypeslib.as_array(bb.codes, shape=(bb.offs[i] + bb.lens[i],))[bb.offs[i]:].tolist()
                words = np.ctypeslib.as_array(bb.packed, shape=(bb.n_words,))[bb.woff[i]: bb.woff[i] + bb.lens[i] // 16 + 4]
                unpacked = [(int(words[q >> 4]) >> (30 - 2 * (q & 15))) & 3 for q in range(bb.lens[i])]
                assert unpacked == codes and all(int(w) == 0 for w in words[bb.lens[i] // 16 + 1:])
                reads.append((C.string_at(bb.ids[i], bb.id_lens[i]), codes))
            if bb.end != 0:
                end = {1: "empty", 2: "bad:" + bb.bad_char.decode("latin1"), 3: "toolong"}[bb.end]
            b = bb.next
        lib.mtrh_batch_free(head)
        if end != "eof":
            break
    lib.mtrh_file_close(C.byref(f))
    return reads, end


FASTA_CASES = {
    "plain": b">r1 desc\nACGT\nacgt\n>r2\nGGGG\n",
    "no_trailing_newline": b">r1\nACGT\n>r2\nGG",
    "crlf": b">r1 x\r\nACGT\r\nTT\r\n>r2\r\nCC\r\n",
    "bases_before_header": b"ACGT\n>r1\nGG\n>r2\nTT\n",
    "no_header": b"ACGTACGT\nGG\n",
    "empty_file": b"",
    "empty_record_middle": b">r1\nAC\n>r2\n>r3\nGG\n",
    "header_only_at_end": b">r1\nAC\n>r2\n",
    "two_headers_first": b">a\n>b\nACGT\n",
    "bad_char": b">r1\nACGT\n>r2\nACNGT\n>r3\nAA\n",
    "cr_hides_rest_of_window": b">r1\nAC\rNNNN\nGT\n",
    "nul_hides_rest": b">r1\nAC\0XX\nGT\n>r2\0hidden\nTT\n",
    "long_line": b">r1\n" + b"ACGT" * 3000 + b"\n>r2\nAC\n",
    "gt_at_window_start": b">r1\n" + b"A" * 4095 + b">x\nCC\n>r3\nGG\n",
    "long_header": b">" + b"h" * 5000 + b"\nACGT\n",
    "long_header_acgt": b">" + b"A" * 4094 + b"CCCC\nGG\n>r2\nTT\n",
    "blank_lines": b">r1\n\nAC\n\n>r2\nGG\n\n",
}


@pytest.mark.parametrize("case", sorted(FASTA_CASES))
def test_fasta_reader_equals_the_reference_loop(cli, tmp_path, case):
    data = FASTA_CASES[case]
    fa = tmp_path / "x.fa"
    fa.write_bytes(data)
    want = reference_reader(data)
    for n_target in (1, 2, 5):
        got = parse_with_host(str(fa), n_target)
        assert got == want, (case, n_target)


def test_fasta_reader_on_many_chunks_of_a_real_file(cli, tmp_path):
    data = open(gu.input_path("synth_c4"), "rb").read()
    want = reference_reader(data)
    for n_target in (1, 3, 11):
        assert parse_with_host(gu.input_path("synth_c4"), n_target) == want
    big = tmp_path / "toolong.fa"
    big.write_bytes(b">ok\nACGT\n>huge\n" + b"ACGTACGTAC" * 100001 + b"\n")
    assert parse_with_host(str(big), 1) == reference_reader(big.read_bytes())


def test_an_overlong_record_at_the_start_of_a_chunk_names_the_record_before_it(cli, tmp_path):
    """the reference's message for a record that reaches MAX_INPUT_LENGTH carries the ID of the record BEFORE it
    (handle_one_file.c:243); when the over-long record opens a chunk that record lives in the previous chunk."""
    big = tmp_path / "toolong2.fa"
    big.write_bytes(b">first one\nACGT\n>second\nGGCC\n>huge\n" + b"ACGTACGTAC" * 100001 + b"\n")
    lib = C.CDLL(os.path.join(hu.HOST, "libmtr_host.so"))
    lib.mtrh_file_open.argtypes = [C.POINTER(File), C.c_char_p]
    lib.mtrh_parse_chunk.restype = C.POINTER(Batch)
    lib.mtrh_parse_chunk.argtypes = [C.POINTER(File), C.c_size_t, C.c_size_t, C.c_int, C.c_int64]
    lib.mtrh_batch_free.argtypes = [C.POINTER(Batch)]
    f = File()
    assert lib.mtrh_file_open(C.byref(f), str(big).encode()) == 0
    data = big.read_bytes()
    begin = data.index(b">huge")
    head = lib.mtrh_parse_chunk(C.byref(f), begin, len(data), 100, 1 << 40)       # a chunk that starts at the over-long record
    bb = head.contents
    assert bb.end == 3 and bb.n == 0
    assert C.string_at(bb.end_id, bb.end_id_len) == b"second"
    lib.mtrh_batch_free(head)
    lib.mtrh_file_close(C.byref(f))


@pytest.mark.parametrize("where,n", [("parser", 3), ("parser", 6), ("device", 2), ("device", 4), ("printer", 1), ("printer", 2)])
def test_a_failed_allocation_in_a_worker_thread_ends_the_run_like_a_device_error(cli, tables, tmp_path, where, n):
    """alloc.c: the n-th allocation of a parser / device / printer thread fails (MTR_TEST_FAIL_ALLOC).  No exit() inside the thread and
    no hang: what was printed before is a prefix of the reference's stdout, the message is on stderr, the status is 1."""
    want = open(os.path.join(gu.GOLDEN, "synth_c4.default.stdout"), "rb").read()
    env = dict(hu.replay_env(tables["default"]), MTR_TEST_FAIL_ALLOC=f"{where}:{n}")
    p = subprocess.run([cli, gu.input_path("synth_c4")], capture_output=True, env=env, timeout=60)
    assert p.returncode == 1, (p.returncode, p.stderr.decode()[-300:])
    assert b"cannot allocate" in p.stderr
    assert want.startswith(p.stdout) and len(p.stdout) < len(want)


def test_parsing_straight_into_the_image_equals_parsing_through_codes(tmp_path):
    """mtrh_parse_chunk_packed (16 characters at a time into the 2-bit image, no code array) against mtrh_parse_chunk on the hostile files, on wrapped lines of
    every width around the block size, on long lines that cross the 4 095-character windows, and on reads whose bases arrive behind 1 .. 15 pending ones"""
    lib = C.CDLL(os.path.join(hu.HOST, "libmtr_host.so"))
    lib.mtrh_plan_chunks.restype = C.POINTER(C.c_size_t)
    lib.mtrh_plan_chunks.argtypes = [C.POINTER(File), C.c_int, C.POINTER(C.c_int)]
    for fn in (lib.mtrh_parse_chunk, lib.mtrh_parse_chunk_packed):
        fn.restype = C.POINTER(Batch)
        fn.argtypes = [C.POINTER(File), C.c_size_t, C.c_size_t, C.c_int, C.c_int64]
    lib.mtrh_batch_free.argtypes = [C.POINTER(Batch)]
    rng = np.random.default_rng(16)
    cases = dict(FASTA_CASES)
    for width in (1, 7, 15, 16, 17, 31, 32, 33, 60, 70, 80, 1023, 1024, 1025):
        recs = []
        for r in range(6):
            n = int(rng.integers(1, 5000))
            seq = "".join("ACGTacgt"[int(x)] for x in rng.integers(0, 8, n))
            recs.append(f">w{width}_{r} x\n" + "\n".join(seq[i:i + width] for i in range(0, n, width)) + "\n")
        cases[f"wrapped_{width}"] = "".join(recs).encode()
    cases["long_lines"] = b"".join(b">L%d\n" % k + bytes(rng.choice(list(b"ACGT"), int(n)).astype(np.uint8)) + b"\n" for k, n in enumerate((4094, 4095, 4096, 4097, 8190, 8191, 12300, 33)))
    cases["nul_laden"] = b"".join(b">Z%d\n" % k + bytes(rng.choice(list(b"ACGT"), int(n))) + b"\n" for k, n in enumerate((4094, 4095, 4096)))     # (int64 items: seven NULs behind every base)
    cases["every_byte"] = b">e\n" + b"".join(b"ACGTACGTACGTACG" + bytes([v]) + b"ACGT\n" for v in range(256) if v not in (10, 13, 0)) + b">f\nAC\n"
    cases["space_and_nul"] = b">s\n" + b"ACGT" * 4 + b" " + b"ACGT" * 4 + b"\n"
    for v in range(256):                                     # every byte value inside a block of sixteen, at every position of the block
        cases[f"byte_{v}"] = b">b\n" + b"".join(b"ACGTACGTACGTACGTACGT"[:q] + bytes([v]) + b"ACGTACGTACGTACGTACGTAC"[:21 - q] + b"\n" for q in (0, 5, 15, 16)) + b">c\nAC\n"
    cases["bad_in_block"] = b">a\n" + b"ACGT" * 10 + b"N" + b"ACGT" * 10 + b"\n>b\nAC\n"
    cases["cr_in_block"] = b">a\r\n" + b"ACGT" * 9 + b"\r\n" + b"GG" * 40 + b"\r\n>b\r\nAC\r\n"
    cases["nul_in_block"] = b">a\n" + b"ACGT" * 5 + b"\0" + b"TTTT" * 8 + b"\n" + b"CC" * 20 + b"\n"
    n_reads = 0
    for name, data in cases.items():
        path = str(tmp_path / f"{name}.fa")
        with open(path, "wb") as fh:
            fh.write(data)
        for n_target in ((1,) if name.startswith("byte_") else (1, 2, 5)):
            f = File()
            assert lib.mtrh_file_open(C.byref(f), path.encode()) == 0
            nc = C.c_int()
            off = lib.mtrh_plan_chunks(C.byref(f), n_target, C.byref(nc))
            for c in range(nc.value):
                for max_reads in (3, 1000):
                    ha = lib.mtrh_parse_chunk(C.byref(f), off[c], off[c + 1], max_reads, 1 << 40)
                    hb = lib.mtrh_parse_chunk_packed(C.byref(f), off[c], off[c + 1], max_reads, 1 << 40)
                    a, b = ha, hb
                    while a or b:
                        assert bool(a) == bool(b), (name, n_target, c)
                        x, y = a.contents, b.contents
                        assert (x.n, x.n_words, x.end, x.bad_char, x.end_id_len) == (y.n, y.n_words, y.end, y.bad_char, y.end_id_len), (name, n_target, c)
                        assert not y.codes
                        if x.n:
                            assert list(x.lens[:x.n]) == list(y.lens[:x.n]) and list(x.woff[:x.n]) == list(y.woff[:x.n]) and list(x.offs[:x.n]) == list(y.offs[:x.n])
                            assert list(x.id_lens[:x.n]) == list(y.id_lens[:x.n])
                            assert [C.string_at(x.ids[i], x.id_lens[i]) for i in range(x.n)] == [C.string_at(y.ids[i], y.id_lens[i]) for i in range(x.n)]
                            wa = np.ctypeslib.as_array(x.packed, shape=(x.n_words,)); wb = np.ctypeslib.as_array(y.packed, shape=(y.n_words,))
                            assert np.array_equal(wa, wb), (name, n_target, c)
                            n_reads += x.n
                        a, b = x.next, y.next
                    lib.mtrh_batch_free(ha); lib.mtrh_batch_free(hb)
            lib.mtrh_file_close(C.byref(f))
    assert n_reads > 500
