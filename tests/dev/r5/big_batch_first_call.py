#!/usr/bin/env python3
"""Development aid (GPU box): mtr_process_batch on ONE batch of n reads in a fresh process - what does the first call cost, and where (library debug timeline)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
base = [c for _, c in synth.make_reads("c4", 10000, 4)]
reads = [base[i % len(base)] for i in range(n)]
e = mtr_amd.Engine()
for it in range(2):
    t0 = time.time(); e.upload(reads); t1 = time.time(); e.run(); t2 = time.time(); d, c = e.fetch_packed(); t3 = time.time()
    print(f"call {it}: upload {1e3 * (t1 - t0):.0f} ms, run {1e3 * (t2 - t1):.0f} ms, fetch {1e3 * (t3 - t2):.0f} ms ({len(d) / 1e6:.0f} MB wire)", flush=True)
