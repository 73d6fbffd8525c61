#!/usr/bin/env python3
"""Development aid (GPU box): the file-order mode at a larger size — every read of a widely mixed file against the oracle's
-B mode (sequential, one core), and the wall time of the C driver with and without -B."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import mtr_amd
from mtr_amd import synth
from tests.oracle_binding import Oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
max_len = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
reads = [c for _, c in synth.make_mixed_file(n, 77, max_len=max_len)]
print(f"{n} reads, {sum(len(c) for c in reads) / 1e6:.1f} Mb, lengths {min(map(len, reads))}..{max(map(len, reads))}", flush=True)
for manhattan in (True, False):
    t0 = time.time()
    o = Oracle(manhattan); o.set_file_order(True)
    want = [o.process(c) for c in reads]
    o.close()
    t_or = time.time() - t0
    e = mtr_amd.Engine(manhattan=manhattan); fs = mtr_amd.FileState()
    t0 = time.time()
    got = []
    for lo in range(0, n, 1000):
        got += e.process_in_file(reads[lo:lo + 1000], fs)
    t_gpu = time.time() - t0
    bad = [i for i in range(n) if [tuple(r) for r in got[i]] != want[i]]
    iso = e.process(reads)
    dif = sum(1 for i in range(n) if [tuple(r) for r in iso[i]] != want[i])
    print(f"manhattan={manhattan}: {len(bad)} reads differ from the oracle's file-order records {bad[:10]}; oracle {t_or:.1f} s, GPU {t_gpu:.2f} s; "
          f"{dif} reads have other records under isolated semantics", flush=True)
    e.close(); fs.close()
if len(sys.argv) > 3:
    sys.exit(0)
fa = "/tmp/c4_20k.fa"
synth.write_fasta(fa, synth.make_reads("c4", 20000, 5))
cli = os.path.join(ROOT, "mtr_amd", "host", "mTR")
for flags in ([], ["-B"]):
    t0 = time.time()
    p = subprocess.run([cli, *flags, fa], capture_output=True)
    print(f"mTR {' '.join(flags)} on 20 000 config-4 reads: {time.time() - t0:.2f} s, rc {p.returncode}, {p.stdout.count(10)} lines", flush=True)
