#!/usr/bin/env python3
"""Development aid (GPU box): host call on one resident batch, the chain in one pass / two passes (MTR_TWO_PASS=0/1/2), several batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
import mtr_amd
from mtr_amd import synth
cases = [("headline2k", 10000), ("c4", 20000), ("c3", 100), ("c3", 600), ("c2", 1000), ("headline2k", 1)]
for cfg, n in cases:
    reads = [c for _, c in synth.make_reads(cfg, n, 2)]
    line = f"{cfg:11s} {n:6d} reads:"
    ref = None
    for mode in ("0", "1", "2"):
        os.environ["MTR_TWO_PASS"] = mode
        e = mtr_amd.Engine(); e.upload(reads)
        ts = []
        for _ in range(4):
            t = time.perf_counter(); e.run(); ts.append((time.perf_counter() - t) * 1e3)
        c = e.counters()
        got = e.fetch_packed()[0]
        if ref is None: ref = got
        line += f"  pass-mode {mode}: {min(ts[1:]):8.2f} ms (searched {c['ranges_searched']}, executed {c['ranges_executed']}, tables {c['kmer_tables']}, dp cells {c['dp_cells']/1e9:.2f} G, same records {got == ref})"
        e.close()
    print(line, flush=True)
