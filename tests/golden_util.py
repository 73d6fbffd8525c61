"""Helpers to read tests/golden (vectors recorded from the unmodified reference, see make_golden.py)."""
from __future__ import annotations

import gzip
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
_LUT = np.full(256, 255, np.uint8)
for _ch, _v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
    _LUT[ord(_ch)] = _v


def input_path(name):
    for ext in (".fa", ".fasta"):
        p = os.path.join(GOLDEN, "inputs", name + ext)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(name)


def read_fasta(path):
    """[(id, codes uint8)] with the reference's record rules (header line = id, sequence lines concatenated)."""
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
    return [(h, _LUT[np.frombuffer(s.encode(), np.uint8)]) for h, s in recs if len(s) > 0]


def cases(mode=None):
    out = []
    for f in sorted(os.listdir(GOLDEN)):
        if f.endswith(".cap.jsonl.gz"):
            name, m = f[: -len(".cap.jsonl.gz")].rsplit(".", 1)
            if mode is None or m == mode:
                out.append((name, m))
    return out


def capture_by_read(name, mode):
    """list (one entry per read, input order) of dicts: {'G1': {...}, 'G3': [...], 'G4': [...], ...}"""
    reads = []
    with gzip.open(os.path.join(GOLDEN, f"{name}.{mode}.cap.jsonl.gz"), "rt") as fh:
        for line in fh:
            ev = json.loads(line)
            if ev["t"] == "G1":
                reads.append({"G1": ev, "G3": [], "G3p": [], "G3r": [], "G4": []})
            else:
                reads[-1][ev["t"]].append(ev)
    return reads


def g4_tuple(ev):
    return (ev["rep_start"], ev["rep_end"], ev["repeat_len"], ev["period"], ev["copies"], ev["mat"], ev["mis"], ev["ins"],
            ev["del"], ev["k"], ev["G"], ev["MM"], ev["D"], ev["unit"], tuple(ev["score"]))


def g1_usable(ev):
    L = ev["L"]
    return [(s, e, w, int(bits, 16)) for s, e, w, bits in ev["ranges"] if -1 < e < L and bits != "bff0000000000000"]
