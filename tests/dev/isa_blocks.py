#!/usr/bin/env python3
"""Development aid: per basic block of one kernel in a `hipcc -S -gline-tables-only` listing, the instruction
mix and the source lines (file:line) it was generated from.  usage: isa_blocks.py file.s kernel_substr file_no lo hi"""
import re, sys
path, kern, fno, lo, hi = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur, loc = [], None, None
for l in lines[start:end + 1]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        cur = {"name": m.group(1), "ins": [], "locs": []}
        blocks.append(cur); continue
    if s.startswith(".loc"):
        p = s.split(); loc = (int(p[1]), int(p[2])); continue
    if not s or s.startswith(";") or s.startswith(".") or cur is None:
        continue
    cur["ins"].append(s.split()[0]); cur["locs"].append(loc)
for b in blocks:
    hit = [x for x in b["locs"] if x and x[0] == fno and lo <= x[1] <= hi]
    if len(hit) * 2 < len(b["ins"]) or not b["ins"]:
        continue
    mix = {}
    for i in b["ins"]:
        k = "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("global_", "scratch_", "buffer_", "flat_")) else i
        mix[k] = mix.get(k, 0) + 1
    ls = sorted(set(x[1] for x in hit))
    print(b["name"], len(b["ins"]), mix, f"lines {ls[0]}-{ls[-1]}", "readlane", sum(1 for i in b["ins"] if "readlane" in i or "readfirstlane" in i), "nop", sum(1 for i in b["ins"] if i == "s_nop"), "scratch", sum(1 for i in b["ins"] if i.startswith("scratch")))
