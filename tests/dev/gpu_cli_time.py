#!/usr/bin/env python3
"""Development aid: wall clock of the command line and of the launcher on synthetic FASTA files (page cache warm)."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mtr_amd import synth
exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "mTR", "libmtr_host.so"], check=True)
with tempfile.TemporaryDirectory() as td:
    base = synth.make_reads("headline2k", 10000, 2)
    for n in (1, 1000, 10000, 100000):
        fa = os.path.join(td, f"r{n}.fa")
        synth.write_fasta(fa, [(str(i), base[i % len(base)][1]) for i in range(n)])
        best = None
        for _ in range(3):
            t = time.perf_counter()
            p = subprocess.run([exe, fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MTR_HOST_TIMING="1"))
            dt = time.perf_counter() - t
            if best is None or dt < best[0]:
                best = (dt, p.stderr.decode().strip().splitlines()[-1] if p.stderr else "")
        print(f"mTR {n} reads: {best[0]:.3f} s = {n / best[0]:.0f} reads/s   {best[1]}", flush=True)
    fa = os.path.join(td, "r100000.fa")
    for world, backend in ((1, None), (2, "gloo")):
        cmd = [sys.executable, "-m", "mtr_amd.run", "--stats"] + (["--gpus", str(world), "--backend", backend] if world > 1 else []) + [fa]
        t = time.perf_counter()
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, cwd=ROOT)
        print(f"launcher {world} rank(s): {time.perf_counter() - t:.3f} s  rc={p.returncode}  {p.stderr.decode().strip().splitlines()[-1][:200] if p.stderr else ''}", flush=True)
