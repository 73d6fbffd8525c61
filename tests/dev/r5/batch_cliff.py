#!/usr/bin/env python3
"""Development aid (GPU box): one resident batch of n headline reads - host call time of two consecutive runs (is there a cliff between 23 000 and 47 000 reads?)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import mtr_amd
from mtr_amd import synth
base = [c for _, c in synth.make_reads("headline2k", 10000, 2)]
for n in (23000, 35000, 47000, 70000):
    reads = [base[i % len(base)] for i in range(n)]
    e = mtr_amd.Engine(); e.upload(reads)
    ts = []
    for _ in range(2):
        t0 = time.time(); e.run(); ts.append((time.time() - t0) * 1e3)
    c = e.counters()
    print(n, "reads:", [round(t, 1) for t in ts], "ms;", e.last_mode(), "records", c["records"], "sent back", c["reads_sent_back"], flush=True)
    e.close()
