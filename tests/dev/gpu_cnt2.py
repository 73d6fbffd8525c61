import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mtr_amd
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("headline2k", 2000, 2)]
eng = mtr_amd.Engine(); eng.upload(reads); eng.run()
c = eng.counters()
print({k: v for k, v in c.items() if not k.startswith("cyc")})
