#!/usr/bin/env python3
"""Development aid: the work counters of one launch in the per-read mode and in the staged mode (what the staged mode searches
and aligns beyond the sequential loop's pruning)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import mtr_amd
    from mtr_amd import synth
    reads = [c for _, c in synth.make_reads(sys.argv[2], int(sys.argv[3]), 2)]
    e = mtr_amd.Engine(); e.upload(reads); e.run()
    c = e.counters()
    print(json.dumps({k: int(v) for k, v in c.items() if not k.startswith("cyc_") and not k.startswith("spare")}))
    sys.exit(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline2k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
res = {}
for name, env in (("per-read", dict(MTR_STAGED="0", MTR_SPLIT="0")), ("staged", dict(MTR_STAGED="1"))):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", cfg, str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
print(f"{'counter':24s} {'per-read':>16s} {'staged':>16s}  ratio")
for k in res["per-read"]:
    a, b = res["per-read"][k], res["staged"].get(k, 0)
    print(f"{k:24s} {a:16d} {b:16d}  {b / a if a else 0:.2f}")
