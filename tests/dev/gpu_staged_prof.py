#!/usr/bin/env python3
"""Development aid: a few staged-mode launches on the headline batch (run it under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MTR_STAGED", "1")
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
reads = [c for _, c in synth.make_reads(cfg, n, 2)]
e = mtr_amd.Engine()
e.upload(reads)
for _ in range(4):
    t = time.perf_counter(); e.run(); print(f"{(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
