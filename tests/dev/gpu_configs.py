#!/usr/bin/env python3
"""Other BASELINE configs on the GPU box (development aid): config-3 shape (42 kb reads), a 140 kb read, mixed c4."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import mtr_amd
from mtr_amd import synth
from tests.oracle_binding import Oracle
eng = mtr_amd.Engine(); orc = Oracle()
def run(label, reads, check):
    t0 = time.time(); eng.upload(reads); t1 = time.time(); eng.run(); t2 = time.time(); res = eng.fetch(); t3 = time.time()
    kt = eng.kernel_times_ms(); c = eng.counters()
    print(f"{label}: {len(reads)} reads, upload {t1-t0:.2f}s run {t2-t1:.2f}s fetch {t3-t2:.2f}s  K1 {kt['k1_ranges']:.1f} ms K2 {kt['k2_units']:.1f} ms  records {c['records']} dp_cells {c['dp_cells']:.3g} global_tables {c['global_tables']}", flush=True)
    bad = 0
    for i in check:
        t = time.time(); want = orc.process(reads[i]); dt = time.time() - t
        ok = [tuple(r) for r in res[i]] == want
        bad += not ok
        print(f"   read {i} (L={len(reads[i])}): oracle {dt:.2f}s, {len(want)} records, {'identical' if ok else 'DIFFERENT'}", flush=True)
    return bad
bad = 0
bad += run("config-3 shape (unit 200 x 200)", [c for _, c in synth.make_reads("c3", 32, 3)], [0, 5])
rng = np.random.RandomState(8)
parts = []
for u, cpy in ((7, 300), (53, 120), (180, 90), (2, 400), (499, 30), (30, 500)):
    r, _ = synth.make_read(rng, u, cpy, 3000, 3000); parts.append(r)
long_read = np.concatenate(parts)
bad += run(f"one long read with six repeats", [long_read], [0])
bad += run("c4 mixed unit lengths", [c for _, c in synth.make_reads("c4", 20000, 4)], [3, 777, 12345])
print("MISMATCHES:", bad)
