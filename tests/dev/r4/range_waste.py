#!/usr/bin/env python3
"""Which candidate ranges does the reference's loop never reach, and what would searching them cost?  

handle_one_read.c:227-246 runs the ranges by ascending start; an accepted repeat removes the later ranges inside it (:178-188).  The
staged chain searches every candidate range.  This script replays the CPU oracle's capture stream (level 1: G1 ranges, G3 DP calls,
G3r revision rounds, G4 records) on N headline reads and reports, by the window width w of a range: how many ranges, how many of
them the loop never reaches, the DP cells and revision rounds of the executed ones, which widths PRODUCE the pruning records - and
what a split of the chain into "wide windows first, then whatever their records leave" would search.

  python tests/dev/r4/range_waste.py [-n 150] [--config headline2k]        (CPU only; ~40 s)
"""
import argparse
import collections
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_isolated  # noqa: E402
from mtr_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=150)
    ap.add_argument("--config", default="headline2k")
    a = ap.parse_args()
    reads = synth.make_reads(a.config, a.n, 2)
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "x.fa")
        synth.write_fasta(fa, reads)
        _, cap = ref_isolated.run_oracle(fa, [], level=1)
    per = []
    for line in cap.decode().splitlines():
        e = json.loads(line)
        if e["t"] == "G1":
            rg = json.loads(e["ranges"].replace("'", '"')) if isinstance(e["ranges"], str) else e["ranges"]
            rd = {"ranges": rg, "set": {(r[0], r[1]) for r in rg}, "recs": [], "last": None, "cells": collections.Counter(), "rev": collections.Counter()}
            per.append(rd)
        elif e["t"] == "G3":
            if (e["qs"], e["qe"]) in rd["set"]:
                rd["last"] = (e["qs"], e["qe"])               # (the other G3 calls are re-alignments of revisions of the same range)
            rd["cells"][rd["last"]] += (e["qe"] - e["qs"] + 1) * len(e["unit"])
        elif e["t"] == "G3r":
            rd["rev"][rd["last"]] += 1
        elif e["t"] == "G4":
            rd["recs"].append((rd["last"], e))

    def pruned_by(R, recs):
        P = set()
        for _, ev in recs:
            rs, re = ev["rep_start"], ev["rep_end"]
            for j, (s, e, _) in enumerate(R):
                if rs <= s < re and e < re:                  # remove_redundant_ranges_from_directional_index
                    P.add(j)
        return P

    by_w = collections.defaultdict(lambda: collections.Counter())
    out = {"config": a.config, "reads": a.n, "split": []}
    for T in (0, 40, 160, 320, 640):
        tc = te = ts = 0
        for rd in per:
            R = sorted((r[0], r[1], r[2]) for r in rd["ranges"])
            P = pruned_by(R, rd["recs"])
            wide = {(s, e) for (s, e, w) in R if w >= T}
            P1 = pruned_by(R, [(p, ev) for p, ev in rd["recs"] if p in wide])
            tc += len(R); te += len(R) - len(P)
            ts += sum(1 for j, (s, e, w) in enumerate(R) if w >= T or j not in P1)
            if T == 0:
                for prod, ev in rd["recs"]:
                    w = [x[2] for x in R if (x[0], x[1]) == prod]
                    by_w[w[0] if w else -1]["ranges_pruned_by_its_records"] += len(pruned_by(R, [(prod, ev)]))
                for j, (s, e, w) in enumerate(R):
                    by_w[w]["ranges"] += 1
                    by_w[w]["never_reached"] += j in P
                    by_w[w]["dp_cells_executed"] += rd["cells"][(s, e)]
                    by_w[w]["revision_rounds_executed"] += rd["rev"][(s, e)]
        out["split"].append({"wide_first_threshold_w": T, "candidates": tc, "executed_by_the_reference": te, "searched": ts, "searched_over_executed": round(ts / te, 4)})
    out["by_window_width"] = {str(w): dict(c) for w, c in sorted(by_w.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
