#!/usr/bin/env python3
"""Development aid (GPU box): device time of every BASELINE.json config with the final build (inputs resident, one run after a warm-up)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mtr_amd
from mtr_amd import synth
eng = mtr_amd.Engine()
for label, cfg, n, seed in (("C2: 1000 reads of 1.25 kb", "c2", 1000, 1), ("headline: 10 000 reads of 2 kb", "headline2k", 10000, 2),
                            ("C3 shape: 100 reads of 42 kb, unit 200 x 200", "c3", 100, 3), ("C4: 100 000 mixed reads of ~2 kb", "c4", 100000, 4)):
    reads = [c for _, c in synth.make_reads(cfg, n, seed)]
    eng.upload(reads); eng.run()
    t0 = time.time(); eng.run(); dt = time.time() - t0
    c = eng.counters()
    print(f"{label}: {dt * 1e3:.1f} ms wall for the run call, kernels {eng.kernel_times_ms()['k2_units']:.1f} ms, {n / dt:.0f} reads/s, records {c['records']}", flush=True)
# the command line end to end on the same shapes (wall clock incl. process start): config 3 with -a, config 5 (15 files, -p) through the launcher
import subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
with tempfile.TemporaryDirectory() as td:
    for label, cfg, n, seed, flags in (("C2 (1000 reads)", "c2", 1000, 1, []), ("C3 shape (100 x 42 kb) -a", "c3", 100, 3, ["-a"]), ("C3 shape (100 x 42 kb)", "c3", 100, 3, [])):
        fa = os.path.join(td, cfg + ".fa")
        synth.write_fasta(fa, synth.make_reads(cfg, n, seed))
        best = 1e9
        for _ in range(2):
            t0 = time.time(); subprocess.run([exe] + flags + [fa], stdout=subprocess.DEVNULL, check=True); best = min(best, time.time() - t0)
        print(f"mTR {' '.join(flags)} {label}: {best:.3f} s wall", flush=True)
