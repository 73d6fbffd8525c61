#!/usr/bin/env python3
"""Development aid: which of the pathological reads of gpu_fuzz.py cost the most — kernel time per block of 60 reads
(the categories are contiguous), one wave per read and range-parallel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "dev", "gpu_fuzz.py")); fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import mtr_amd
reads = fz.make(3, 6)
for split in ("0", "1"):
    os.environ["MTR_SPLIT"] = split
    eng = mtr_amd.Engine()
    eng.process(reads[:50])
    rows = []
    for lo in range(0, len(reads), 60):
        blk = reads[lo:lo + 60]
        eng.upload(blk); eng.run()
        ms = eng.kernel_times_ms()["k2_units"]
        c = eng.counters()
        rows.append((ms, lo, max(map(len, blk)), sum(map(len, blk)), c["records"], c["dp_cells"] + c["revise_dp_cells"], c["ranges_executed"]))
    rows.sort(reverse=True)
    print(f"MTR_SPLIT={split}: total {sum(r[0] for r in rows):.0f} ms; slowest blocks (ms, first read, max L, bases, records, cells, ranges):")
    for r in rows[:8]:
        print("   ", r, "head", "".join("ACGT"[b] for b in reads[r[1]][:30]))
    eng.close()
# the slowest block read by read
os.environ["MTR_SPLIT"] = "0"
eng = mtr_amd.Engine()
lo = rows[0][1]
per = []
for i in range(lo, min(lo + 60, len(reads))):
    eng.upload([reads[i]]); eng.run()
    c = eng.counters()
    per.append((eng.kernel_times_ms()["k2_units"], i, len(reads[i]), c["records"], c["dp_cells"] + c["revise_dp_cells"], c["ranges_executed"], c["kmer_tables"], c["walk_steps"], c["traceback_steps"]))
per.sort(reverse=True)
print("reads of that block alone, one wave each (ms, read, L, records, cells, ranges, tables, walk steps, traceback steps):")
for p in per[:6]:
    print("   ", p, "".join("ACGT"[b] for b in reads[p[1]][:40]))
