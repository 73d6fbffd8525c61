#!/usr/bin/env python3
"""Development aid (GPU box): the fuzz set of tests/dev/gpu_fuzz.py in the two-pass chain - time, reads sent back, per-kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "dev"))
import gpu_fuzz
import mtr_amd
seed, manhattan = int(sys.argv[1]), sys.argv[2] == "m"
reads = gpu_fuzz.make(seed, 3)
for tp in ("0", "1"):
    os.environ.update(MTR_STAGED="1", MTR_QUAD_MIN="1", MTR_TWO_PASS=tp)
    e = mtr_amd.Engine(manhattan=manhattan)
    e.upload(reads)
    for _ in range(2):
        t = time.perf_counter(); e.run(); dt = time.perf_counter() - t
    c = e.counters()
    print(f"two_pass={tp}: {dt * 1e3:.1f} ms, mode {e.last_mode()}, sent back {c['reads_sent_back']}, searched {c['ranges_searched']}, executed {c['ranges_executed']}, kernels {({k: round(v, 1) for k, v in e.kernel_times_ms().items()})}", flush=True)
    e.close()
