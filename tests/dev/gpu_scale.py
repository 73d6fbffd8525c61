#!/usr/bin/env python3
"""K2/K1 time vs batch size (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MTR_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mtr_amd", "libmtr_hip_prof.so"))   # the build with the phase timers
import mtr_amd
from mtr_amd import synth
sizes = [int(x) for x in sys.argv[1:]] or [64, 256, 1536, 3072, 10000, 20000]
reads = [c for _, c in synth.make_reads("headline2k", max(sizes), 2)]
eng = mtr_amd.Engine()
for n in sizes:
    eng.upload(reads[:n]); eng.run(); eng.run()
    c = eng.counters(); kt = eng.kernel_times_ms()
    print(f"n={n:6d}  K2 {kt['k2_units']:8.1f} ms  K1 {kt['k1_ranges']:7.1f} ms  reads/s {n/(kt['k2_units']+kt['k1_ranges'])*1e3:9.0f}  wave-Mcycles/read {c['cyc_total']/n/1e6:7.1f}  shader clock {c['cyc_total']/max(c['reserved'],1)*100/1e3:5.2f} GHz  wave-ms/read {c['reserved']/n/1e5:6.1f}", flush=True)
