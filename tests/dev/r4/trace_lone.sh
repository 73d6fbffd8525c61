#!/bin/bash
# Development aid, ON THE GPU BOX: per-kernel durations of lone staged launches (rocprofv3 --kernel-trace --stats).  bash tests/dev/r4/trace_lone.sh [n] [config] [tag]
set -o pipefail
N=${1:-10000}; CFG=${2:-headline2k}; TAG=${3:-lone}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4_trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MTR_STAGED=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- python3 $ROOT/tests/dev/gpu_staged_prof.py $N $CFG > "$OUT/trace.out" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
rows = []
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
rows.sort(key=lambda r: -r[3])
with open(os.path.join(out, "kernels.txt"), "w") as fh:
    for r in rows:
        line = f"{r[0][:40]:40s} calls {r[1]:4d}  avg {r[2]:9.3f} ms  total {r[3]:9.2f} ms"
        print(line); fh.write(line + "\n")
PY
