#!/usr/bin/env python3
"""Latency of small batches: one wavefront per read (MTR_SPLIT=0) against the range-parallel mode (MTR_SPLIT=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import mtr_amd
from mtr_amd import synth
reads2k = [c for _, c in synth.make_reads("headline2k", 4096, 2)]
c3 = [c for _, c in synth.make_reads("c3", 100, 3)]
for split in ("0", "1"):
    os.environ["MTR_SPLIT"] = split
    eng = mtr_amd.Engine()
    eng.process(reads2k[:4])
    for label, batch in (("1 read of 2 kb", reads2k[:1]), ("16 reads of 2 kb", reads2k[:16]), ("256 reads of 2 kb", reads2k[:256]),
                         ("1024 reads of 2 kb", reads2k[:1024]), ("4096 reads of 2 kb", reads2k), ("1 read of 42 kb", c3[:1]), ("100 reads of 42 kb (config 3)", c3)):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); eng.process(batch); ts.append(time.perf_counter() - t0)
        print(f"MTR_SPLIT={split}  {label:32s} {min(ts)*1e3:9.2f} ms  (kernels {eng.kernel_times_ms()['k2_units']:.2f} ms)", flush=True)
    eng.close()
