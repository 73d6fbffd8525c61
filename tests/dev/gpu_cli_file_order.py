#!/usr/bin/env python3
"""Development aid (GPU box): `mTR -B` on a file of more reads than one driver batch (16 384) against the oracle's -B mode —
the file state has to carry over from batch to batch and between the driver's two contexts."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
reads = synth.make_mixed_file(n, 5, max_len=3000)
fa = "/tmp/fo_cli.fa"
synth.write_fasta(fa, reads)
t0 = time.time(); a = subprocess.run([os.path.join(ROOT, "mtr_amd", "host", "mTR"), "-B", fa], capture_output=True); t1 = time.time()
b = subprocess.run([os.path.join(ROOT, "oracle", "mtr_oracle_cli"), "-B", fa], capture_output=True); t2 = time.time()
c = subprocess.run([os.path.join(ROOT, "mtr_amd", "host", "mTR"), fa], capture_output=True)
NL = b"\n"
ndiff = sum(1 for x, y in zip(a.stdout.split(NL), c.stdout.split(NL)) if x != y)
print(f"{n} reads: mTR -B {t1 - t0:.2f} s, oracle -B {t2 - t1:.1f} s; stdout identical: {a.stdout == b.stdout} ({a.stdout.count(10)} lines); "
      f"isolated run differs from it in {ndiff} lines")
sys.exit(0 if a.stdout == b.stdout and a.returncode == 0 else 1)
