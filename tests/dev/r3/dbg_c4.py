import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["MTR_DEBUG"] = "1"
import mtr_amd
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("c4", 100000)]
e = mtr_amd.Engine()
for _ in range(2):
    e.upload(reads)
    t = time.perf_counter(); e.run(); print("run", round((time.perf_counter() - t) * 1e3, 1), "ms", e.last_mode(), flush=True)
