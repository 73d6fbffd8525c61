#!/usr/bin/env python3
"""End-to-end time of the C driver (FASTA in -> report lines out) on a synthetic file (development aid)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
cfg = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "in.fa")
    t0 = time.perf_counter()
    synth.write_fasta(fa, [(str(i), c) for i, (_, c) in enumerate(synth.make_reads(cfg, n, 4))])
    print(f"generated {n} reads of {cfg} in {time.perf_counter()-t0:.1f} s, {os.path.getsize(fa)/1e6:.0f} MB", flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "mtr_amd", "host", "mTR"), "-c", fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, MTR_HOST_TIMING="1"))
        dt = time.perf_counter() - t0
        print(f"run {rep}: rc {p.returncode}, {dt:.2f} s wall -> {n/dt:.0f} reads/s end to end, {len(p.stdout.splitlines())} report lines")
        print(p.stderr.decode()[-600:])
