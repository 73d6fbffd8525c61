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
