import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
reads = [c for _, c in synth.make_reads("headline2k", n, 2)]
eng = mtr_amd.Engine(); eng.upload(reads); eng.run()
c = eng.counters()
tot = c["cyc_total"]
print("fused kernel ms", eng.kernel_times_ms())
for k, v in c.items():
    if k.startswith("cyc"):
        print(f"  {k:18s} {v/n/1e6:9.3f} Mcyc/read  {100*v/tot:6.2f} % of wave time")
