#!/bin/bash
# Development aid, ON THE GPU BOX: every kernel launch of ONE long read's chain, in launch order (rocprofv3 --kernel-trace).
#   bash tests/dev/r3/prof_long_read.sh TAG [-p] name...
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_long_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d "$OUT/trace" -o trace --output-format csv -- python3 $ROOT/tests/dev/r3/long_read_phases.py "$@" > "$OUT/trace.out" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
# the second run of every read (the first one allocates): print the launches of the LAST chain per read
t0 = rows[0][0] if rows else 0
for t, n, ms in rows:
    if ms >= 0.05: print(f"{(t - t0) / 1e6:10.2f} ms  {n[:40]:40s} {ms:9.3f} ms")
print(open(os.path.join(out, "trace.out")).read()[:600])
PY
