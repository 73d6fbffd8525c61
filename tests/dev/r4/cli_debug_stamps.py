#!/usr/bin/env python3
"""Development aid (GPU box): the command line on 10 000 / 100 000 reads with the host's stamps (MTR_HOST_TIMING) AND the library's own
(MTR_DEBUG: runtime start-up, allocations, chain enqueued), plus what an empty HIP program takes on the same box (tests/dev/hip_init_probe.hip)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from mtr_amd import synth
exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
probe = "/tmp/hip_init_probe"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", probe, os.path.join(ROOT, "tests", "dev", "hip_init_probe.hip")], check=True)
for rep in range(2):
    print(f"== empty HIP program, run {rep}\n" + subprocess.run([probe], capture_output=True, text=True).stdout, flush=True)
with tempfile.TemporaryDirectory() as td:
    base = synth.make_reads("headline2k", 10000, 2)
    for n in (10000, 100000):
        fa = os.path.join(td, f"r{n}.fa")
        synth.write_fasta(fa, [(str(i), base[i % len(base)][1]) for i in range(n)])
        for rep in range(3):
            env = dict(os.environ, MTR_HOST_TIMING="1")
            if rep == 2:
                env["MTR_DEBUG"] = "1"
            t = time.perf_counter()
            p = subprocess.run([exe, fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.perf_counter() - t
            print(f"== mTR {n} reads, run {rep}: {dt:.3f} s wall", flush=True)
            if rep >= 1:
                print("\n".join(l for l in p.stderr.decode().splitlines() if "chain:" not in l), flush=True)
