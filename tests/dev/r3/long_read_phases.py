"""Development aid: the chain's phases on ONE long read (bundled files of config 5).  python tests/dev/r3/long_read_phases.py [-p] names..."""
import sys
import time

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
import mtr_amd
from tests import golden_util as gu

manhattan = "-p" not in sys.argv
for name in [a for a in sys.argv[1:] if a != "-p"]:
    if name.startswith("c3:"):
        from mtr_amd import synth
        reads = [c for _, c in synth.make_reads("c3", int(name[3:]))]
    else:
        reads = [c for _, c in gu.read_fasta(gu.input_path(name))]
    e = mtr_amd.Engine(manhattan=manhattan)
    for _ in range(2):
        e.upload(reads)
        t0 = time.perf_counter(); e.run(); dt = time.perf_counter() - t0
    print(name, sum(len(c) for c in reads), "bases", round(dt * 1e3, 2), "ms", {k: round(v, 2) for k, v in e.kernel_times_ms().items()}, flush=True)
    c = e.counters() if hasattr(e, "counters") else {}
    print("   counters", {k: v for k, v in c.items() if v and not k.startswith("cyc_")})
    e.close()
