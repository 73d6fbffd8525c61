#!/usr/bin/env python3
"""Accuracy table of unit prediction on synthetic reads with a known unit.

The table the reference's test_single_TR/test.sh prints: for unit lengths 2..200 x 10 copies, flanks of unit x copies
random bases, Nanopore error profile (substitution 1.6 %, insertion 9.0 %, deletion 3.8 %), N reads each:
  * reads for which a reported unit is a rotation of the true unit;
  * report lines whose unit aligns to the true unit with match ratio >= 1, 0.99, 0.98, 0.96, 0.94 (wrap-around
    alignment, tools/unit_score.c).
Reads come from this repository's seeded generator (mtr_amd/synth.py; the reference's generator is seeded from
random_device), so the table is reproducible.  Runs the C driver mtr_amd/host/mTR (GPU); with --reference the
reference binary oracle/_ref/mTR_ref (CPU, slow) is scored on the same files next to it.

  python tools/accuracy.py [-n 1000] [--units 2 5 10 20 50 100 200] [--copies 10] [--reference] [--seed 1]
"""
from __future__ import annotations

import argparse
import collections
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
THRESHOLDS = (1.0, 0.99, 0.98, 0.96, 0.94)


def load_scorer():
    so = os.path.join(HERE, "libunit_score.so")
    src = os.path.join(HERE, "unit_score.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src], check=True)
    lib = C.CDLL(so)
    lib.us_is_rotation.argtypes = [C.c_char_p, C.c_char_p]
    lib.us_is_rotation.restype = C.c_int
    lib.us_match_ratio.argtypes = [C.c_char_p, C.c_char_p]
    lib.us_match_ratio.restype = C.c_double
    return lib


def score(lib, report: str, truth: list[str]):
    """report = stdout of mTR; truth[i] = unit of read i (IDs are the decimal read indices)."""
    exact = set()
    ratios = []
    for line in report.splitlines():
        f = line.split("\t")
        if len(f) < 13:
            continue
        rid, unit = int(f[0]), f[12].strip()
        if lib.us_is_rotation(unit.encode(), truth[rid].encode()):
            exact.add(rid)
        ratios.append(lib.us_match_ratio(truth[rid].encode(), unit.encode()))
    r = np.array(ratios) if ratios else np.zeros(0)
    return {"exact_rotation_reads": len(exact), "report_lines": len(ratios),
            **{f"ratio>={t:g}": int((r >= t).sum()) for t in THRESHOLDS}}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("-n", type=int, default=1000)
    ap.add_argument("--units", type=int, nargs="+", default=[2, 5, 10, 20, 50, 100, 200])
    ap.add_argument("--copies", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--reference", action="store_true", help="also score oracle/_ref/mTR_ref on the same reads (CPU)")
    ap.add_argument("--file-order", action="store_true", help="run the GPU driver with -B (the reference's whole-file behaviour), so that "
                    "with --reference the two reports can only differ where two chains tie")
    a = ap.parse_args()
    from mtr_amd import synth

    lib = load_scorer()
    mtr = os.path.join(ROOT, "mtr_amd", "host", "mTR")
    ref = os.path.join(ROOT, "oracle", "_ref", "mTR_ref")
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for u in a.units:
            rng = np.random.RandomState(a.seed * 1000 + u)
            flank = u * a.copies
            reads, truth = [], []
            for i in range(a.n):
                codes, unit = synth.make_read(rng, u, a.copies, flank, flank)
                reads.append((str(i), codes))
                truth.append("".join("ACGT"[int(x)] for x in unit))
            fa = os.path.join(td, f"u{u}.fa")
            synth.write_fasta(fa, reads)
            row = {"unit_len": u, "copies": a.copies, "reads": a.n}
            outs = {}
            for label, exe in (("gpu", mtr),) + ((("ref", ref),) if a.reference else ()):
                t0 = time.perf_counter()
                p = subprocess.run([exe] + (["-B"] if label == "gpu" and a.file_order else []) + [fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                if p.returncode != 0:
                    sys.exit(f"{exe} failed: {p.stderr[-400:]}")
                row[label] = score(lib, p.stdout, truth)
                row[label]["seconds"] = round(time.perf_counter() - t0, 2)
                outs[label] = collections.Counter(p.stdout.split("\n"))
            if len(outs) == 2:
                row["report_lines_only_in_one"] = sum(((outs["gpu"] - outs["ref"]) + (outs["ref"] - outs["gpu"])).values())
            rows.append(row)
            print(row, flush=True)
    print()
    hdr = ["unit", "who", "reads", "exact rotation", "lines"] + [f">={t:g}" for t in THRESHOLDS] + ["s"]
    print(" | ".join(hdr))
    for row in rows:
        for label in ("gpu", "ref"):
            if label in row:
                s = row[label]
                print(" | ".join(str(x) for x in [row["unit_len"], label, row["reads"], s["exact_rotation_reads"], s["report_lines"]]
                                 + [s[f"ratio>={t:g}"] for t in THRESHOLDS] + [s["seconds"]]))


if __name__ == "__main__":
    main()
