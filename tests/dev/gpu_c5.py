#!/usr/bin/env python3
"""Development aid: BASELINE config 5 (the 15 single-read files of test_multiple_TRs, -p) — host-call time per file on one GPU
and what the launcher's longest-first assignment makes of it on 2/4/8 ranks (predicted from those times)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mtr_amd
from tests import golden_util as gu
from tests.test_run_gloo import BUNDLED
e = mtr_amd.Engine(manhattan=False)
times, sizes = [], []
for name in BUNDLED:
    reads = [c for _, c in gu.read_fasta(gu.input_path(name))]
    e.upload(reads)
    ts = []
    for _ in range(3):
        t = time.perf_counter(); e.run(); ts.append((time.perf_counter() - t) * 1e3)
    times.append(min(ts)); sizes.append(len(reads[0]))
    print(f"{name:28s} L {len(reads[0]):7d}  {min(ts):8.2f} ms", flush=True)
tot = sum(times)
print(f"one GPU, file after file: {tot:.1f} ms")
for world in (2, 4, 8):
    order = sorted(range(len(sizes)), key=lambda i: -sizes[i])
    load, tl = [0.0] * world, [0.0] * world
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        load[r] += sizes[i] ** 1.5; tl[r] += times[i]
    print(f"{world} ranks (longest first by L^1.5): slowest rank {max(tl):.1f} ms, mean {tot / world:.1f} ms -> speed-up {tot / max(tl):.2f}x of {world}")
