#!/bin/bash
# Development aid, ON THE GPU BOX: per-kernel time of staged-mode launches (rocprofv3 --kernel-trace --stats).
#   bash tests/dev/r3/prof_staged_trace.sh [n_reads] [config] [tag]
set -o pipefail
N=${1:-10000}; CFG=${2:-headline2k}; TAG=${3:-t}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_sttrace_${TAG}_${N}_$CFG
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
        rows.append((r["Name"].split("(")[0].replace("void ", ""), int(r["Calls"]), float(r["AverageNs"]) / 1e6))
rows.sort(key=lambda x: -x[1] * x[2])
tot = 0
for n, c, ms in rows:
    if c >= 4: tot += ms * c / 4
    print(f"{n[:44]:44s} {c:5d} {ms:9.3f} ms")
print(f"sum of kernels per launch ~ {tot:.2f} ms")
print(open(os.path.join(out, "trace.out")).read())
PY
