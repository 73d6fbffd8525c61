#!/usr/bin/env python3
"""Development aid: instruction mix per innermost loop of one kernel in a `hipcc -S -gline-tables-only` listing.
usage: isa_loops.py file.s kernel_label [min_instr]"""
import re, sys
path, kern = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
loops = {}
cur_hdr, loc = None, (0, 0)
for l in lines[start:end + 1]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", s)
    if m:
        name, rest = m.group(1), m.group(2)
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", rest)
        if mm:
            cur_hdr = (mm.group(1), int(mm.group(2)))
        elif "Loop Header" in rest or "Parent Loop" in rest or "=>This" in rest:
            cur_hdr = None  # decided by following comment lines
            pending = name[2:]
        else:
            cur_hdr = None
        continue
    mm = re.match(r"^;\s*=>This (Inner )?Loop Header: Depth=(\d+)", s)
    if mm:
        cur_hdr = (pending, int(mm.group(2)))
        continue
    if s.startswith(".loc"):
        p = s.split(); loc = (int(p[1]), int(p[2])); continue
    if not s or s[0] in ";." or cur_hdr is None:
        continue
    op = s.split()[0]
    d = loops.setdefault(cur_hdr, {"n": 0, "valu": 0, "salu": 0, "lds": 0, "vmem": 0, "dpp": 0, "nop": 0, "locs": {}})
    d["n"] += 1
    k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    d[k] += 1
    if "dpp" in s or "row_" in s or "wave_sh" in s: d["dpp"] += 1
    if op == "s_nop": d["nop"] += 1
    d["locs"][loc] = d["locs"].get(loc, 0) + 1
for (h, depth), d in sorted(loops.items(), key=lambda x: -x[1]["n"]):
    if d["n"] < mn: continue
    byfile = {}
    for (f, ln), c in d["locs"].items():
        byfile.setdefault(f, []).append(ln)
    desc = " ".join(f"f{f}:{min(v)}-{max(v)}" for f, v in sorted(byfile.items()) if f in (1, 8, 9))
    print(f"{h:12s} depth {depth} n={d['n']:4d} valu={d['valu']:4d} salu={d['salu']:4d} lds={d['lds']:3d} vmem={d['vmem']:3d} dpp={d['dpp']:3d} nop={d['nop']:3d}  {desc}")
