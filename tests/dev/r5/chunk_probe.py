#!/usr/bin/env python3
"""Development aid (GPU box): the command line on 100 000 / 300 000 reads of 2 kb for several chunk sizes (MTR_CHUNK_BYTES): python tests/dev/r5/chunk_probe.py"""
import os, subprocess, sys, time, tempfile
sys.path.insert(0, os.getcwd())
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("headline2k", 10000, 2)]
td = tempfile.mkdtemp()
for n in (100000, 300000):
    fa = os.path.join(td, f"r{n}.fa")
    synth.write_fasta(fa, [(str(i), reads[i % len(reads)]) for i in range(n)])
    for mb in (12, 24, 48, 96):
        best = None
        for _ in range(3):
            e = dict(os.environ, MTR_HOST_TIMING="1", MTR_CHUNK_BYTES=str(mb << 20))
            t0 = time.perf_counter(); p = subprocess.run(["mtr_amd/host/mTR", fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e); dt = time.perf_counter() - t0
            st = {ln.split("] ", 1)[1]: float(ln.split("+")[1].split(" s")[0]) for ln in p.stderr.decode().splitlines() if ln.startswith("[host +")}
            own = st.get("last batch fetched", 0) - st.get("first device context created", 0)
            if best is None or dt < best[0]: best = (dt, own)
        print(f"{n} reads, chunks of {mb} MiB: {best[0]:.3f} s wall = {n / best[0] / 1e3:.0f} k reads/s; first context -> last fetch {best[1]:.3f} s = {n / best[1] / 1e3:.0f} k reads/s", flush=True)
