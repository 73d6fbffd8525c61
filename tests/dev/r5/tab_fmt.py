#!/usr/bin/env python3
"""Development aid: the k-mer table builds of a launch by table layout (all kernels).  Needs libraries built with -DMTR_PROFILE -DMTR_PROFILE_WALK_FMT=4
(cycles inside tab_build) and =5 (tables): mtr_amd/libmtr_hip_wfmt4.so / _wfmt5.so.  One process per library."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
n = sys.argv[1] if len(sys.argv) > 1 else "10000"
wl = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
names = ("direct, k <= 4", "direct, k = 5, 6", "packed, <= 1024 slots", "packed, 2048 slots", "split, keys in LDS", "table in global memory")
res = {}
for which in (4, 5):
    env = dict(os.environ, MTR_LIB=os.path.join(ROOT, "mtr_amd", f"libmtr_hip_wfmt{which}.so"))
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "walk_fmt.py"), "child", n, wl], env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    res[which] = json.loads(out.stdout.strip().splitlines()[-1])
print(f"{wl}, {n} reads: tables {res[4]['cyc_tab_build']/1e9:.1f} G cycles, {res[4]['kmer_tables']} tables")
for i, k in enumerate(("prof43", "prof44", "prof45", "prof46", "prof47", "prof55")):
    cyc, nt = res[4][k], res[5][k]
    print(f"  {names[i]:28s} {cyc/1e9:8.2f} G cycles  {nt:10d} tables  {cyc/max(nt,1):8.0f} cycles a table")
