"""Test helpers for the host side (mtr_amd/host): builds the host binaries and the replay engine (tests/replay_engine.c,
a stand-in for libmtr_hip.so that answers with the reference's recorded records), and writes replay tables from the
golden G4 captures.  Test infrastructure only."""
from __future__ import annotations

import os
import struct
import subprocess

import numpy as np

from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mtr_amd", "host")
REPLAY_SRC = os.path.join(ROOT, "tests", "replay_engine.c")
REPLAY_LIB = os.path.join(ROOT, "tests", "libmtr_replay.so")
P = np.uint64(1099511628211)


def build_host():
    subprocess.run(["make", "-s", "-C", HOST, "mTR", "libmtr_host.so"], check=True)
    return os.path.join(HOST, "mTR")


def build_replay():
    hdr = os.path.join(ROOT, "include", "mtr_hip.h")
    if not os.path.exists(REPLAY_LIB) or max(os.path.getmtime(REPLAY_SRC), os.path.getmtime(hdr)) > os.path.getmtime(REPLAY_LIB):
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                        "-o", REPLAY_LIB, REPLAY_SRC], check=True)
    return REPLAY_LIB


NULL_SRC = os.path.join(ROOT, "tests", "null_engine.c")
NULL_LIB = os.path.join(ROOT, "tests", "libmtr_null.so")


def build_null():
    """tests/null_engine.c: an engine that costs the host nothing (every read answered with the same records at once) - the host pipeline's ceiling"""
    hdr = os.path.join(ROOT, "include", "mtr_hip.h")
    if not os.path.exists(NULL_LIB) or max(os.path.getmtime(NULL_SRC), os.path.getmtime(hdr)) > os.path.getmtime(NULL_LIB):
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-Wall", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                        "-o", NULL_LIB, NULL_SRC, "-lpthread"], check=True)
    return NULL_LIB


def host_ceiling(fasta, n_reads, n_gpus=8, records=2, repeats=2, extra_env=None):
    """`mTR -g n_gpus <fasta>` behind the null engine: wall clock, CPU seconds (user + sys of the child) and lines printed; best of `repeats`.
    The FASTA is the caller's (page cache warm after the first run)."""
    import resource
    import time
    env = dict(os.environ, MTR_LIB=build_null(), MTR_NULL_RECORDS=str(records))
    env.pop("MTR_REPLAY_TABLE", None)
    if extra_env:
        env.update(extra_env)
    exe = build_host()
    best = None
    for _ in range(repeats):
        r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as null:
            p = subprocess.run([exe, "-g", str(n_gpus), fasta], stdout=subprocess.PIPE if best is None else null, stderr=subprocess.PIPE, env=env)
        dt = time.perf_counter() - t0
        r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
        assert p.returncode == 0, p.stderr.decode()[-400:]
        cur = {"seconds": dt, "user_s": r1.ru_utime - r0.ru_utime, "sys_s": r1.ru_stime - r0.ru_stime}
        if best is None:
            cur["lines"] = p.stdout.count(b"\n")
            lines = cur["lines"]
            best = cur                                   # (the first run also reads its report through a pipe: slower, kept only if nothing beats it)
        elif dt < best["seconds"]:
            cur["lines"] = lines
            best = cur
    best.update(reads=n_reads, gpus=n_gpus, records_per_read=records, reads_per_s=n_reads / best["seconds"],
                cores_used=(best["user_s"] + best["sys_s"]) / best["seconds"], host_cores=os.cpu_count(),
                reads_per_s_per_core_used=n_reads / (best["user_s"] + best["sys_s"]))
    return best


def hash_codes(codes: np.ndarray) -> int:
    """sum of (code + 1) * P^i mod 2^64 (tests/replay_engine.c: hash_codes)"""
    n = len(codes)
    if n == 0:
        return 0
    with np.errstate(over="ignore"):
        pw = np.concatenate([np.ones(1, np.uint64), np.cumprod(np.full(n - 1, P, np.uint64))]) if n > 1 else np.ones(1, np.uint64)
        return int(np.sum((codes.astype(np.uint64) + np.uint64(1)) * pw, dtype=np.uint64))


def wire_record(t) -> bytes:
    """one record (tests.golden_util.g4_tuple order) in the wire form of include/mtr_hip.h"""
    unit, score = t[13], t[14]
    per = t[3]
    assert len(unit) == per and len(score) >= per, (per, len(unit), len(score))
    head = struct.pack("<14i", *t[:13], 0)
    return head + unit.encode() + b"\0" * ((-per) % 4) + struct.pack(f"<{per}i", *score[:per])


def write_table(path, cases):
    """cases: [(golden name, mode)] with mode 'default' or 'p' -> replay table of every read of those inputs"""
    out = []
    for name, mode in cases:
        reads = gu.read_fasta(gu.input_path(name))
        cap = gu.capture_by_read(name, mode)
        assert len(reads) == len(cap), (name, len(reads), len(cap))
        for (_, codes), per_read in zip(reads, cap):
            wire = b"".join(wire_record(gu.g4_tuple(ev)) for ev in per_read["G4"])
            out.append(struct.pack("<iiQq", len(codes), len(per_read["G4"]), hash_codes(codes), len(wire)) + wire)
    with open(path, "wb") as fh:
        fh.write(b"MTRREPLY" + struct.pack("<q", len(out)) + b"".join(out))
    return path


def replay_env(table):
    return dict(os.environ, MTR_LIB=build_replay(), MTR_REPLAY_TABLE=table)
