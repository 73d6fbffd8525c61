#!/usr/bin/env python3
"""Development aid: per-read cost of the per-read kernel (trace type 7, shader-clock cycles) next to the read's candidate
ranges, to judge cost predictors for ordering the work queue."""
import os, sys, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTR_STAGED"] = "0"; os.environ["MTR_SPLIT"] = "0"; os.environ["MTR_TRACE_MASK"] = str(1 << 7)
os.environ["MTR_LIB"] = os.path.join(ROOT, "mtr_amd", "libmtr_hip_prof.so")
import numpy as np
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cfg = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
reads = [c for _, c in synth.make_reads(cfg, n, 2)]
e = mtr_amd.Engine()
e.set_trace(n + 16)
e.upload(reads)
e.run()
ev = e.get_trace()
ev = ev[ev[:, 0] == 7]
cost = np.zeros(n); cost[ev[:, 1]] = ev[:, 2]
rng = e.test_ranges()
pickle.dump({"cost": cost, "ranges": [[(s, en, w) for s, en, w, _ in r] for r in rng], "lens": [len(r) for r in reads]}, open(os.path.join(ROOT, "gpurun_out", f"cost_{cfg}.pkl"), "wb"))
print("reads", n, "mean cost", cost.mean(), "max/mean", cost.max() / cost.mean(), "p99/mean", np.percentile(cost, 99) / cost.mean())
