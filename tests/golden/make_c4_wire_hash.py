#!/usr/bin/env python3
"""Known answer for BASELINE config 4 at full size: sha256 of the record tables (wire form of include/mtr_hip.h, read
after read in input order) that the CPU ORACLE (oracle/mtr_oracle.c, pinned to the reference) produces for the 100 000
mixed-unit reads of mtr_amd.synth config "c4" (seed 4).  bench.py --strong c4 compares the stream gathered from the N
ranks with it: the multi-GPU result must be bit-identical to the reference's, whatever N.

  python tests/golden/make_c4_wire_hash.py [-j 6]      -> tests/golden/c4_100k_wire.json   (~3 min on 6 cores)
  python tests/golden/make_c4_wire_hash.py --config c3 -n 100   -> tests/golden/c3_100_wire.json (config 3's shape: 100 reads of 42 kb)
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def work(args):
    lo, hi = args
    from mtr_amd import synth
    from tests.host_util import wire_record
    from tests.oracle_binding import Oracle
    reads = READS[lo:hi]
    orc = Oracle()
    out, nrec = [], 0
    for _, codes in reads:
        recs = orc.process(codes)
        nrec += len(recs)
        out.append(b"".join(wire_record(r) for r in recs))
    orc.close()
    return lo, b"".join(out), nrec


def main():
    global READS
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=6)
    ap.add_argument("-n", type=int, default=100000)
    ap.add_argument("--config", default="c4", help="c4 (seed 4); c3 = BASELINE config 3's shape, seed 3 (bench.py's secondary.c3: -n 100)")
    a = ap.parse_args()
    from mtr_amd import synth
    seed = synth.CONFIGS[a.config][4]          # (c4: 4, c3: 3, c2: 1)
    READS = synth.make_reads(a.config, a.n, seed)
    step = 2 if a.config == "c3" else 100 if a.config == "c2" else 500
    jobs = [(lo, min(lo + step, a.n)) for lo in range(0, a.n, step)]
    h = hashlib.sha256()
    total_bytes = total_rec = 0
    with mp.get_context("fork").Pool(a.j) as pool:
        for lo, blob, nrec in pool.imap(work, jobs):          # imap keeps input order
            h.update(blob)
            total_bytes += len(blob)
            total_rec += nrec
    out = {"config": a.config, "seed": seed, "n_reads": a.n, "records": total_rec, "wire_bytes": total_bytes, "sha256": h.hexdigest(),
           "sum_len": int(sum(len(c) for _, c in READS)), "made_by": "tests/golden/make_c4_wire_hash.py (CPU oracle)"}
    path = os.path.join(ROOT, "tests", "golden", "c4_100k_wire.json" if (a.n == 100000 and a.config == "c4") else f"{a.config}_{a.n}_wire.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
