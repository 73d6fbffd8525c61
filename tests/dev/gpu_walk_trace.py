#!/usr/bin/env python3
"""Development aid: the walks of the most expensive (window, k) searches of a read (trace type 8, timed build)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["MTR_LIB"] = os.path.join(ROOT, "mtr_amd", "libmtr_hip_prof.so")
os.environ.setdefault("MTR_STAGED", "1")
os.environ["MTR_TRACE_MASK"] = str(1 << 8)
import mtr_amd
from mtr_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
reads = [c for _, c in synth.make_reads(cfg, 1, 2)]
e = mtr_amd.Engine(); e.upload(reads); e.set_trace(400000); e.run()
ev = e.get_trace()
ev = ev[ev[:, 0] == 8]
print(len(ev), "walks")
by = collections.defaultdict(list)
for r in ev:
    by[(int(r[2]), int(r[3]), int(r[4]), int(r[5]))].append((int(r[7]), int(r[8]), int(r[9]), int(r[10])))     # (qs, qe, k, backward) -> (steps, period, kcycles, lookups)
tot = sorted(((sum(w[2] for w in ws), key, ws) for key, ws in by.items()), reverse=True)
for kc, key, ws in tot[:6]:
    print(f"window {key[0]}..{key[1]} k {key[2]} {'backward' if key[3] else 'forward '}: {len(ws)} walks, {kc / 1024:.1f} M cycles, steps {sum(w[0] for w in ws)}, look-ups {sum(w[3] for w in ws)}, closed {sum(1 for w in ws if w[1] > 0)}")
    print("    walks (steps, unit, kcycles):", [(w[0], w[1], w[2]) for w in ws[:24]], "..." if len(ws) > 24 else "")
