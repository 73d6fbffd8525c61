#!/usr/bin/env python3
"""Development aid: every walk of a read (trace type 8: window, k, direction, seed, steps, unit length) under two libraries, and where they differ.
usage: walk_diff.py <golden input name | synth config> <lib A> <lib B>"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("MTR_STAGED", "1")
    os.environ["MTR_TRACE_MASK"] = str(1 << 8)
    import mtr_amd
    from mtr_amd import synth
    from tests import golden_util as gu
    name = sys.argv[2]
    try:
        reads = [c for _, c in gu.read_fasta(gu.input_path(name))][:1]
    except Exception:
        reads = [c for _, c in synth.make_reads(name, 1, 2)]
    e = mtr_amd.Engine(); e.upload(reads); e.set_trace(400000); e.run()
    ev = e.get_trace()
    ev = ev[ev[:, 0] == 8]
    print(json.dumps(sorted([int(x) for x in (r[2], r[3], r[4], r[5], r[6], r[7], r[8])] for r in ev)))
    sys.exit(0)
name, la, lb = sys.argv[1], sys.argv[2], sys.argv[3]
out = {}
for lib in (la, lb):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", name], env=dict(os.environ, MTR_LIB=os.path.join(ROOT, "mtr_amd", lib)), capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        print(p.stderr[-3000:]); sys.exit(1)
    out[lib] = [tuple(x) for x in json.loads(p.stdout.strip().splitlines()[-1])]
a, b = out[la], out[lb]
print(len(a), "walks under", la, "/", len(b), "under", lb)
sa, sb = set(a), set(b)
print("only under", la, ":", len(sa - sb)); 
for x in sorted(sa - sb)[:30]: print("   ", x)
print("only under", lb, ":", len(sb - sa));
for x in sorted(sb - sa)[:30]: print("   ", x)
