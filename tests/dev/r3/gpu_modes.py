#!/usr/bin/env python3
"""Development aid: the kernel modes against each other — records must be identical — and their time for a host call:
fused = one wavefront per read (mtr_k_reads); split = range-parallel (round 1); staged<U> = the staged mode with units up
to U bases aligned one DP per lane and the others one wavefront per DP."""
import os, sys, time, subprocess, pickle
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
MODES = [("fused", dict(MTR_STAGED="0", MTR_SPLIT="0")), ("staged_noquad", dict(MTR_STAGED="1", MTR_QUAD_MIN="0")), ("staged_auto", dict(MTR_STAGED="1")), ("staged_quads", dict(MTR_STAGED="1", MTR_QUAD_MIN="1"))]

def child(cfg, n, out):
    import mtr_amd
    from mtr_amd import synth
    from tests import golden_util as gu
    if cfg.startswith("golden:"):
        reads = [c for _, c in gu.read_fasta(gu.input_path(cfg[7:]))]
    else:
        reads = [c for _, c in synth.make_reads(cfg, n, 7)]
    e = mtr_amd.Engine()
    e.upload(reads)
    ts = []
    for _ in range(4):
        t = time.perf_counter(); e.run(); ts.append((time.perf_counter() - t) * 1e3)
    recs = e.fetch()
    pickle.dump(([[tuple(r) for r in g] for g in recs], min(ts)), open(out, "wb"))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child(sys.argv[2], int(sys.argv[3]), sys.argv[4]); sys.exit(0)

cases = [("golden:3_5", 0), ("golden:synth_c4", 0), ("golden:edge", 0), ("golden:10_50", 0), ("golden:worm_chrI", 0), ("golden:worm_chrII_1", 0), ("headline2k", 1), ("c3", 1), ("headline2k", 64), ("headline2k", 256), ("c3", 16), ("c3", 100),
         ("c4", 2000), ("headline2k", 1000), ("headline2k", 2000), ("headline2k", 4000), ("c2", 1000), ("headline2k", 10000), ("c4", 20000)]
if len(sys.argv) > 2:
    cases = [(sys.argv[1], int(sys.argv[2]))]
bad = 0
for cfg, n in cases:
    res = {}
    for name, envs in MODES:
        out = f"/tmp/st_{name}.pkl"
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", cfg, str(n), out], env=dict(os.environ, **envs), capture_output=True, text=True, timeout=900)
        if p.returncode != 0:
            print(cfg, n, name, "FAILED", p.stderr[-500:], flush=True); bad += 1; continue
        res[name] = pickle.load(open(out, "rb"))
    if "fused" not in res:
        continue
    ref = res["fused"][0]
    line = f"{cfg} n={len(ref)}:"
    for name, _ in MODES:
        if name not in res:
            continue
        diff = sum(1 for i in range(len(ref)) if ref[i] != res[name][0][i])
        bad += diff > 0
        line += f"  {name} {res[name][1]:.2f} ms" + (f" [{diff} reads DIFFER]" if diff else "")
    print(line, flush=True)
sys.exit(1 if bad else 0)
