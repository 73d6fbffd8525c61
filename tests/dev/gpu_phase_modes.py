#!/usr/bin/env python3
"""Development aid: phase timers (wave-cycles, libmtr_hip_prof.so) of one 10 000-read launch in the per-read and the staged mode."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["MTR_LIB"] = os.path.join(ROOT, "mtr_amd", "libmtr_hip_prof.so")
    import mtr_amd
    from mtr_amd import synth
    reads = [c for _, c in synth.make_reads(sys.argv[2], int(sys.argv[3]), 2)]
    e = mtr_amd.Engine(); e.upload(reads); e.run(); e.run()
    print(json.dumps({k: int(v) for k, v in e.counters().items()}))
    sys.exit(0)
cfg = sys.argv[1] if len(sys.argv) > 1 else "headline2k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
res = {}
for name, env in (("per-read", dict(MTR_STAGED="0")), ("staged", dict(MTR_STAGED="1"))):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", cfg, str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
print(f"{'timer (G wave-cycles)':24s} {'per-read':>10s} {'staged':>10s}")
for k in res["per-read"]:
    if k.startswith("cyc_"):
        print(f"{k:24s} {res['per-read'][k] / 1e9:10.2f} {res['staged'].get(k, 0) / 1e9:10.2f}")
for k in ("tb_refills", "traceback_steps", "dp_rows", "dp_calls", "revise_dp_calls", "revise_dp_cells", "dp_cells"):
    print(f"{k:24s} {res['per-read'][k]:>14d} {res['staged'].get(k, 0):>14d}")
