#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle comparison with readable diagnostics (development aid; run on the GPU box)."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import mtr_amd
from mtr_amd import synth
from tests.oracle_binding import Oracle


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    manh = not (len(sys.argv) > 3 and sys.argv[3] == "p")
    reads = [c for _, c in synth.make_reads(cfg, n, 7)]
    eng = mtr_amd.Engine(manhattan=manh)
    orc = Oracle(manhattan=manh)
    eng.upload(reads)
    # ---- K1
    t0 = time.time()
    g = eng.test_ranges()
    print(f"K1 ran in {time.time()-t0:.3f}s  kernel {eng.kernel_times_ms()}")
    bad = 0
    for i, codes in enumerate(reads):
        o = orc.ranges(codes)
        if o != g[i]:
            bad += 1
            if bad <= 3:
                print(f"read {i}: ranges differ: oracle {len(o)} gpu {len(g[i])}")
                for k, (a, b) in enumerate(zip(o, g[i])):
                    if a != b:
                        print("   first diff at", k, "oracle", a, "gpu", b)
                        break
                print("   oracle head", o[:4], "\n   gpu head   ", g[i][:4])
    print(f"K1 ranges: {len(reads)-bad}/{len(reads)} reads identical")
    # ---- DP
    rng = np.random.RandomState(3)
    tasks = []
    for i, codes in enumerate(reads):
        L = len(codes)
        for U in (1, 2, 3, 5, 17, 64, 65, 100, 130, 257):
            if U * 6 >= L:
                continue
            qs = int(rng.randint(0, L // 3)); qe = int(rng.randint(qs + 5 * U, L - 1))
            st = int(rng.randint(qs, qe - U))
            unit = codes[st:st + U].copy()
            for (G, MM, D) in ((1, 1, 3), (1, 3, 1), (5, 1, 1)):
                tasks.append((i, qs, qe, unit, G, MM, D))
    t0 = time.time()
    out = eng.test_wrap_dp(tasks)
    print(f"DP test: {len(tasks)} tasks in {time.time()-t0:.3f}s kernel {eng.kernel_times_ms()} counters {eng.counters()['dp_cells']}")
    bad = 0
    for t, task in enumerate(tasks):
        o = orc.wrap_dp(reads[task[0]], task[1], task[2], task[3], task[4], task[5], task[6])
        if tuple(int(x) for x in out[t]) != o:
            bad += 1
            if bad <= 5:
                print(f"  DP task {t} U={len(task[3])} rows={task[2]-task[1]+1} params={task[4:]}: oracle {o} gpu {tuple(int(x) for x in out[t])}")
    print(f"DP: {len(tasks)-bad}/{len(tasks)} identical")
    # ---- full
    eng.set_trace(200000)
    t0 = time.time()
    eng.run()
    res = eng.fetch()
    print(f"full run in {time.time()-t0:.3f}s kernel {eng.kernel_times_ms()}")
    print("counters", eng.counters())
    bad = 0
    for i, codes in enumerate(reads):
        o = orc.process(codes)
        gg = [tuple(r) for r in res[i]]
        if o != gg:
            bad += 1
            if bad <= 3:
                print(f"read {i} (L={len(codes)}): records differ: oracle {len(o)} gpu {len(gg)}")
                for a, b in zip(o, gg):
                    if a != b:
                        print("   oracle", a[:13], a[13][:40]); print("   gpu   ", b[:13], b[13][:40]); break
    print(f"records: {len(reads)-bad}/{len(reads)} reads identical")
    print("oracle stats", orc.stats())


if __name__ == "__main__":
    main()
