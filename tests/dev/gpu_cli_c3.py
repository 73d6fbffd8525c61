#!/usr/bin/env python3
"""Config 3 through the C driver with -a (alignment blocks printed by the host): wall time next to the reference."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
with tempfile.TemporaryDirectory() as td:
    fa = os.path.join(td, "c3.fa")
    synth.write_fasta(fa, synth.make_reads("c3", n, 3))
    outs = {}
    for label, exe, reads in (("gpu", os.path.join(ROOT, "mtr_amd", "host", "mTR"), n), ("ref", os.path.join(ROOT, "oracle", "_ref", "mTR_ref"), min(n, 8))):
        path = fa
        if reads != n:
            path = os.path.join(td, "c3_small.fa"); synth.write_fasta(path, synth.make_reads("c3", reads, 3))
        t0 = time.perf_counter()
        p = subprocess.run([exe, "-a", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        dt = time.perf_counter() - t0
        outs[label] = p.stdout
        print(f"{label}: {reads} reads, -a, rc {p.returncode}, {dt:.2f} s wall, {len(p.stdout)/1e6:.1f} MB of output", flush=True)
    # the first 8 reads' output must agree byte for byte up to the reference's batch-order state leaks
    a = outs["gpu"].split(b"\n"); b = outs["ref"].split(b"\n")
    same = sum(1 for x, y in zip(a, b) if x == y)
    print(f"first {len(b)} lines: {same} identical")
