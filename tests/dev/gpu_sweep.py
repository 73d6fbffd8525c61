#!/usr/bin/env python3
"""Broad parity sweep (development aid): thousands of seeded reads of every synthetic config, every kernel mode, Manhattan
and Pearson, against the CPU oracle (run in a process pool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from concurrent.futures import ProcessPoolExecutor
import numpy as np
from mtr_amd import synth

def oracle_chunk(args):
    manhattan, reads = args
    from tests.oracle_binding import Oracle
    o = Oracle(manhattan=manhattan)
    out = [o.process(r) for r in reads]
    o.close()
    return out

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    import mtr_amd
    bad_total = 0
    with ProcessPoolExecutor(max_workers=12) as pool:
        plan = (("headline2k", 101, True), ("c4", 102, True), ("c2", 103, True), ("headline2k", 104, False), ("c4", 105, False))
        if len(sys.argv) > 2:                               # one config at full size, e.g. "100000 c4" = BASELINE config 4
            plan = ((sys.argv[2], 4, True),)
        for cfg, seed, manhattan in plan:
            reads = [c for _, c in synth.make_reads(cfg, n if manhattan else n // 3, seed)]
            chunks = [reads[i:i + 100] for i in range(0, len(reads), 100)]
            t0 = time.time()
            want = [w for ch in pool.map(oracle_chunk, [(manhattan, c) for c in chunks]) for w in ch]
            t_or = time.time() - t0
            modes = {"per-read": dict(MTR_STAGED="0", MTR_QUAD_MIN="0", MTR_TWO_PASS="0"), "staged": dict(MTR_STAGED="1", MTR_QUAD_MIN="0", MTR_TWO_PASS="0"),
                     "staged+quads": dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="0"),
                     "staged+quads, two passes": dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="1")}
            for split in (("per-read", "staged+quads", "staged+quads, two passes") if len(reads) > 20000 else tuple(modes)):
                os.environ.update(modes[split])
                eng = mtr_amd.Engine(manhattan=manhattan)
                t0 = time.time(); got = eng.process(reads); t_gpu = time.time() - t0
                bad = [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]
                bad_total += len(bad)
                print(f"{cfg:11s} seed {seed} {'manhattan' if manhattan else 'pearson  '} {split}: {len(reads)} reads, {len(bad)} differ"
                      f" (gpu {t_gpu:.2f} s, oracle pool {t_or:.1f} s){' first: ' + str(bad[:5]) if bad else ''}", flush=True)
                eng.close()
    print("TOTAL MISMATCHES:", bad_total)
    sys.exit(1 if bad_total else 0)

if __name__ == "__main__":
    main()
