import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]); cfg = sys.argv[2]
reads = [c for _, c in synth.make_reads(cfg, n, 4)]
e = mtr_amd.Engine()
e.upload(reads)
for i in range(2):
    t = time.time(); e.run(); print("run", i, round(time.time() - t, 3), e.last_mode(), {k: v for k, v in e.counters().items() if k in ("reads_sent_back", "ranges_searched", "ranges_executed")}, flush=True)
