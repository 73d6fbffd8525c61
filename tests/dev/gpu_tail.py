#!/usr/bin/env python3
"""Per-read cost distribution of K2 (development aid): which reads are the stragglers and why."""
import sys, os
os.environ["MTR_TRACE_MASK"] = str(1 << 7)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ.setdefault("MTR_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mtr_amd", "libmtr_hip_prof.so"))   # the build with the phase timers
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
reads = [c for _, c in synth.make_reads("headline2k", n, 2)]
eng = mtr_amd.Engine()
eng.set_trace(n + 16)
eng.upload(reads); eng.run()
ev = eng.get_trace()
ev = ev[ev[:, 0] == 7]
tot = ev[:, 2].astype(np.float64) * 1024 / 1e6
print(f"{len(ev)} reads: Mcycles/read mean {tot.mean():.1f} median {np.median(tot):.1f} p90 {np.percentile(tot,90):.1f} p99 {np.percentile(tot,99):.1f} max {tot.max():.1f}")
names = ["total", "dp_fwd", "dp_tb", "rev_fwd", "rev_tb", "tab", "walk", "dp_calls", "dp_cells_k", "lookups_k", "tables", "ranges", "tb_steps", "L"]
order = np.argsort(-tot)
print("slowest reads:")
for i in order[:8]:
    print("  read", ev[i, 1], {nm: int(ev[i, 2 + k]) for k, nm in enumerate(names)})
print("median-ish reads:")
for i in order[len(order)//2: len(order)//2 + 3]:
    print("  read", ev[i, 1], {nm: int(ev[i, 2 + k]) for k, nm in enumerate(names)})
for k, nm in enumerate(names[:7]):
    print(f"  share {nm:8s} all {ev[:,2+k].sum()/ev[:,2].sum():.3f}   top1% {ev[order[:max(1,len(order)//100)],2+k].sum()/ev[order[:max(1,len(order)//100)],2].sum():.3f}")
print("correlation of Mcycles/read with:")
for k, nm in enumerate(names):
    x = ev[:, 2 + k].astype(np.float64)
    if x.std() > 0:
        print(f"  {nm:10s} r = {np.corrcoef(x, tot)[0, 1]:.3f}")
rng = eng.test_ranges()
rid = ev[:, 1].astype(int)
nr = np.array([len(rng[i]) for i in rid], float)
sl = np.array([sum(e - s + 1 for s, e, w, d in rng[i]) for i in rid], float)
sw = np.array([sum(w for s, e, w, d in rng[i]) for i in rid], float)
mx = np.array([max([e - s + 1 for s, e, w, d in rng[i]] or [0]) for i in rid], float)
for nm, x in (("n_ranges_K1", nr), ("sum_range_len", sl), ("sum_w", sw), ("max_range_len", mx)):
    print(f"  {nm:14s} r = {np.corrcoef(x, tot)[0, 1]:.3f}")
A = np.stack([nr, sl, sw, mx, np.ones_like(nr)], 1)
coef, *_ = np.linalg.lstsq(A, tot, rcond=None)
pred = A @ coef
print("  linear fit r =", np.corrcoef(pred, tot)[0, 1], "coef", coef)
