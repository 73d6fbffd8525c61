#!/usr/bin/env python3
"""Development aid: where the walk kernels' wave-cycles go (the build with the phase timers, staged chain, headline batch).
A library built with -DMTR_PROFILE -DMTR_PROFILE_WALK_WIDTH (MTR_LIB=...) puts the cycles of the unit searches of windows of
< 32 / < 64 / < 128 / < 256 / >= 256 positions into spare43..46 and spare55 (round 4: 7.4 / 4.8 / 4.6 / 4.8 / ~55 G of 76 G)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MTR_LIB", os.path.join(ROOT, "mtr_amd", "libmtr_hip_prof.so"))
os.environ.setdefault("MTR_STAGED", "1")
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads = [c for _, c in synth.make_reads("headline2k", n, 2)]
eng = mtr_amd.Engine()
eng.upload(reads); eng.run(); eng.run()
c = eng.counters(); kt = eng.kernel_times_ms()
print(eng.last_mode(), {k: round(v, 2) for k, v in kt.items()})
tot = c["cyc_total"]
for k in sorted(c):
    if k.startswith("cyc_"):
        print(f"   {k:18s} {c[k]/1e6:10.1f} M  {100.0*c[k]/max(tot,1):5.1f} %")
print({k: c[k] for k in ("tb_refills", "traceback_steps", "dp_calls", "revise_dp_calls", "kmer_tables", "kmer_lookups", "tables_skipped", "walk_steps", "walk_slow_steps", "walk_calls", "walk_closed", "ranges_searched", "ranges_executed")})
