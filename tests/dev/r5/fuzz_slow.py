#!/usr/bin/env python3
"""Development aid (GPU box): the fuzz set of seed 3 in the chain's two-pass mode with quads - where do 5 s go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "dev"))
import numpy as np
import gpu_fuzz
import mtr_amd
reads = gpu_fuzz.make(3, 3)
print(len(reads), "reads", sum(map(len, reads)), "bases, longest", max(map(len, reads)))
for name, env in (("staged+quads", dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="0")), ("staged+quads two passes", dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="1")),
                  ("staged two passes", dict(MTR_STAGED="1", MTR_QUAD_MIN="0", MTR_TWO_PASS="1")), ("staged+quads mode 2", dict(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS="2"))):
    os.environ.update(env)
    e = mtr_amd.Engine()
    e.upload(reads)
    for it in range(2):
        t0 = time.time(); e.run(); dt = time.time() - t0
        k = e.kernel_times_ms(); c = e.counters()
        print(f"{name} run {it}: {dt * 1e3:.1f} ms host; mode {e.last_mode()}; kernels {({kk: round(v, 1) for kk, v in k.items()})}; sent back {c['reads_sent_back']}, searched {c['ranges_searched']} executed {c['ranges_executed']}, dp_calls {c['dp_calls']} rev {c['revise_dp_calls']}", flush=True)
    e.close()
