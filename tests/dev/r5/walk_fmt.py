#!/usr/bin/env python3
"""Development aid: the fast walk steps of mtr_k_walks_k by table layout.  Needs libraries built with -DMTR_PROFILE -DMTR_PROFILE_WALK_FMT=1
(cycles inside walk_fast) and =2 (steps): mtr_amd/libmtr_hip_wfmt1.so / _wfmt2.so (hipcc line: README, development tools).  One process per library."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("MTR_STAGED", "1")
    import mtr_amd
    from mtr_amd import synth
    n = int(sys.argv[2])
    reads = [c for _, c in synth.make_reads(sys.argv[3], n, 2)]
    eng = mtr_amd.Engine()
    eng.upload(reads); eng.run(); eng.run()
    c = eng.counters()
    print(json.dumps({k: c[k] for k in ("prof43", "prof44", "prof45", "prof46", "prof47", "prof55", "cyc_walk_fast", "cyc_walk", "cyc_walk_slow", "cyc_tab_build", "walk_steps", "walk_slow_steps", "walk_calls", "kmer_tables")}))
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "10000"
wl = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
names = ("direct (k <= 6)", "packed, <= 1024 slots", "packed, 2048 slots", "split, counts in LDS", "split, counts in global memory", "table in global memory")
res = {}
for which in (1, 2):
    env = dict(os.environ, MTR_LIB=os.path.join(ROOT, "mtr_amd", f"libmtr_hip_wfmt{which}.so"))
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", n, wl], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    res[which] = json.loads(out.stdout.strip().splitlines()[-1])
print(f"{wl}, {n} reads, mtr_k_walks_k only; all kernels: walk {res[1]['cyc_walk']/1e9:.1f} G cycles (fast {res[1]['cyc_walk_fast']/1e9:.1f}, general steps {res[1]['cyc_walk_slow']/1e9:.1f}), tables {res[1]['cyc_tab_build']/1e9:.1f} G; {res[1]['walk_steps']} steps, {res[1]['walk_calls']} walks, {res[1]['kmer_tables']} tables")
for i, k in enumerate(("prof43", "prof44", "prof45", "prof46", "prof47", "prof55")):
    cyc, st = res[1][k], res[2][k]
    print(f"  {names[i]:34s} {cyc/1e9:8.2f} G cycles  {st:10d} fast steps  {cyc/max(st,1):8.0f} cycles a step")
