#!/usr/bin/env python3
"""Development aid: where the command line's wall clock goes (MTR_HOST_TIMING stamps) for 1 / 1 000 / 100 000 reads."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mtr_amd import synth
exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
with tempfile.TemporaryDirectory() as td:
    base = synth.make_reads("headline2k", 10000, 2)
    for n in (1, 1000, 10000, 100000):
        fa = os.path.join(td, f"r{n}.fa")
        synth.write_fasta(fa, [(str(i), base[i % len(base)][1]) for i in range(n)])
        for rep in range(2):
            t = time.perf_counter()
            p = subprocess.run([exe, fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MTR_HOST_TIMING="1", **({"MTR_DEBUG": "1"} if False else {})))
            dt = time.perf_counter() - t
            print(f"== mTR {n} reads, run {rep}: {dt:.3f} s wall", flush=True)
            if rep == 1: print(p.stderr.decode(), flush=True)
