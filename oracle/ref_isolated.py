#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY — run the unmodified reference under *isolated semantics*.

Isolated semantics = the reference run once per read, one process per
read, outputs concatenated in input order.  This script splits a FASTA, runs
oracle/_ref/ref_capture (the reference + interposed capture hooks) on each single-read file in
parallel, and concatenates stdout (G5) and the capture JSONL (G1..G4).  With --check it also runs
oracle/mtr_oracle_cli on the whole FASTA and compares both streams byte for byte.

Only usable where oracle/_ref has been built (i.e. where /root/reference exists).
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CAPTURE = os.path.join(HERE, "_ref", "ref_capture")
ORACLE_CLI = os.path.join(HERE, "mtr_oracle_cli")


def split_fasta(path):
    """Yield (header_line_without_gt, sequence) keeping the reference's notion of a record."""
    recs, hdr, seq = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if hdr is not None:
                    recs.append((hdr, "".join(seq)))
                hdr, seq = line[1:], []
            else:
                seq.append(line)
    if hdr is not None:
        recs.append((hdr, "".join(seq)))
    return recs


def _run_one(args):
    idx, hdr, seq, flags, level, tmpdir = args
    fa = os.path.join(tmpdir, f"r{idx}.fa")
    cap = os.path.join(tmpdir, f"r{idx}.jsonl")
    with open(fa, "w") as fh:
        fh.write(f">{hdr}\n{seq}\n")
    p = subprocess.run([REF_CAPTURE, *flags, "-l", str(level), fa, cap], capture_output=True)
    if p.returncode != 0:
        raise RuntimeError(f"reference failed on read {idx}: {p.stderr.decode()[:500]}")
    with open(cap, "rb") as fh:
        c = fh.read()
    os.unlink(fa)
    os.unlink(cap)
    return p.stdout, c


def run_reference_isolated(fasta, flags=(), level=1, workers=8):
    recs = split_fasta(fasta)
    with tempfile.TemporaryDirectory(prefix="mtr_iso_") as tmpdir:
        jobs = [(i, h, s, list(flags), level, tmpdir) for i, (h, s) in enumerate(recs)]
        with ThreadPoolExecutor(max_workers=workers) as ex:
            outs = list(ex.map(_run_one, jobs))
    return b"".join(o for o, _ in outs), b"".join(c for _, c in outs)


def run_reference_whole_file(fasta, flags=(), level=1):
    """The reference as its users run it: one process for the whole file.  Its results depend on the order of the
    reads: leak A (stale tail of inputString_w_rand / orgInputString) shows in the capture stream,
    leak B (std::set in heap-pointer order) only in which of two equal-score chains is printed."""
    with tempfile.TemporaryDirectory(prefix="mtr_file_") as tmpdir:
        cap = os.path.join(tmpdir, "cap.jsonl")
        p = subprocess.run([REF_CAPTURE, *flags, "-l", str(level), fasta, cap], capture_output=True)
        if p.returncode != 0:
            raise RuntimeError(f"reference failed: {p.stderr.decode()[:500]}")
        with open(cap, "rb") as fh:
            return p.stdout, fh.read()


def run_oracle(fasta, flags=(), level=1):
    with tempfile.NamedTemporaryFile(suffix=".jsonl", delete=False) as tf:
        cap = tf.name
    try:
        p = subprocess.run([ORACLE_CLI, *flags, "-l", str(level), "-C", cap, fasta], capture_output=True)
        if p.returncode != 0:
            raise RuntimeError(f"oracle failed: {p.stderr.decode()[:500]}")
        with open(cap, "rb") as fh:
            c = fh.read()
    finally:
        os.unlink(cap)
    return p.stdout, c


def first_diff(a: bytes, b: bytes):
    la, lb = a.split(b"\n"), b.split(b"\n")
    for i, (x, y) in enumerate(zip(la, lb)):
        if x != y:
            return i, x[:300], y[:300]
    if len(la) != len(lb):
        return min(len(la), len(lb)), b"<eof>", b"<eof>"
    return None


def main():
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("fasta")
    ap.add_argument("--flags", default="", help="mTR flags, e.g. '-p' or '-a'")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--out", help="write reference stdout here")
    ap.add_argument("--cap", help="write reference capture JSONL here")
    ap.add_argument("--check", action="store_true", help="compare with oracle/mtr_oracle_cli")
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--file-order", action="store_true", help="the reference in ONE process for the whole file against the oracle's -B mode")
    a = ap.parse_args()
    flags = a.flags.split()
    if a.file_order:
        r_out, r_cap = run_reference_whole_file(a.fasta, flags, a.level)
    else:
        r_out, r_cap = run_reference_isolated(a.fasta, flags, a.level, a.workers)
    if a.out:
        open(a.out, "wb").write(r_out)
    if a.cap:
        open(a.cap, "wb").write(r_cap)
    rc = 0
    if a.check:
        o_out, o_cap = run_oracle(a.fasta, flags + (["-B"] if a.file_order else []), a.level)
        for name, x, y in (("stdout", r_out, o_out), ("capture", r_cap, o_cap)):
            d = first_diff(x, y)
            n = x.count(b"\n")
            if d is None:
                print(f"{name}: identical ({n} lines)")
            else:
                rc = 1
                print(f"{name}: DIFFERS at line {d[0]} of {n}\n  ref:    {d[1]!r}\n  oracle: {d[2]!r}")
    return rc


if __name__ == "__main__":
    sys.exit(main())
