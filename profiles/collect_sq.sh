#!/bin/bash
# Run ON THE GPU BOX from the repo root: bash profiles/collect_sq.sh <tag>
# SQ issue/wait counters of the per-read kernel (8 SQ slots per pass -> two passes), no tracing domains.
set -o pipefail
TAG=${1:-latest}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$OUT/p1" -o a --output-format csv -- $BENCH > "$OUT/bench1.json" 2> "$OUT/p1.err" || exit 1
echo "pass 1 done"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d "$OUT/p2" -o b --output-format csv -- $BENCH > "$OUT/bench2.json" 2> "$OUT/p2.err" || exit 1
echo "pass 2 done"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for p in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        if "mtr_k_reads" not in r.get("Kernel_Name", ""):
            continue
        a = acc.setdefault(r["Counter_Name"], {})
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
s = {k: sum(v.values()) / len(v) for k, v in acc.items()}
s["_launches"] = {k: len(v) for k, v in acc.items()}
json.dump(s, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
print(json.dumps(s))
PY
