#!/usr/bin/env python3
"""Known answers for the command line on whole BASELINE configs: sha256 (and line count) of the stdout the CPU ORACLE's command line (oracle/mtr_oracle_cli,
pinned to the reference: same report lines as reference mTR run one read per process) prints for the reads of a mtr_amd.synth config written by
synth.write_fasta (IDs = the decimal read index).  bench.py's secondary.c2 / secondary.c4 compare `mTR -g N <fasta>`'s stdout with them: the multi-GPU
product path must print what the reference prints, whatever N.

  python tests/golden/make_stdout_hash.py c4 100000 [-j 7]     -> tests/golden/c4_100000_stdout.json   (~6 min on 7 cores)
  python tests/golden/make_stdout_hash.py c2 1000              -> tests/golden/c2_1000_stdout.json
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("n", type=int)
    ap.add_argument("-j", type=int, default=7)
    ap.add_argument("-p", action="store_true", help="Pearson distance (-p)")
    a = ap.parse_args()
    from mtr_amd import synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    cli = os.path.join(ROOT, "oracle", "mtr_oracle_cli")
    seed = synth.CONFIGS[a.config][4]
    reads = synth.make_reads(a.config, a.n, seed)
    h = hashlib.sha256()
    lines = total = 0
    with tempfile.TemporaryDirectory() as td:
        per = max(1, (a.n + 8 * a.j - 1) // (8 * a.j))
        shards = []
        for lo in range(0, a.n, per):
            f = os.path.join(td, f"s{lo}.fa")
            synth.write_fasta(f, reads[lo:lo + per])
            shards.append(f)
        # isolated semantics: a read's lines do not depend on its neighbours, so the shards' outputs in order are the whole file's output
        running = []
        nxt = 0
        outs = {}
        while nxt < len(shards) or running:
            while nxt < len(shards) and len(running) < a.j:
                o = open(shards[nxt] + ".out", "wb")
                running.append((nxt, subprocess.Popen([cli] + (["-p"] if a.p else []) + [shards[nxt]], stdout=o, stderr=subprocess.DEVNULL), o))
                nxt += 1
            i, p, o = running.pop(0)
            assert p.wait() == 0
            o.close()
            outs[i] = shards[i] + ".out"
        for i in range(len(shards)):
            b = open(outs[i], "rb").read()
            h.update(b)
            lines += b.count(b"\n")
            total += len(b)
    out = {"config": a.config, "seed": seed, "n_reads": a.n, "pearson": bool(a.p), "stdout_lines": lines, "stdout_bytes": total, "sha256": h.hexdigest(),
           "made_by": "tests/golden/make_stdout_hash.py (oracle/mtr_oracle_cli on the FASTA of synth.write_fasta)"}
    path = os.path.join(ROOT, "tests", "golden", f"{a.config}_{a.n}{'_p' if a.p else ''}_stdout.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, out)


if __name__ == "__main__":
    main()
