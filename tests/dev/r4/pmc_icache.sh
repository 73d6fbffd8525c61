#!/bin/bash
# Development aid, ON THE GPU BOX: instruction-cache counters of the bench command, per kernel.  bash tests/dev/r4/pmc_icache.sh [tag]
set -o pipefail
TAG=${1:-icache}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4_pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=4
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency --no-cli --no-secondary"
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -d "$OUT/pmc" -o pmc --output-format csv -- $BENCH > "$OUT/bench.json" 2> "$OUT/pmc.err" || { tail -5 "$OUT/pmc.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for p in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
with open(os.path.join(out, "icache.txt"), "w") as fh:
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        req, miss = v.get("SQC_ICACHE_REQ", 0), v.get("SQC_ICACHE_MISSES", 0)
        line = f"{k[:36]:36s} icache req {req:14.0f} miss {miss:14.0f} ({100*miss/max(req,1):5.1f} %)  wave cycles {v.get('SQ_WAVE_CYCLES',0):.3e} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.3f} valu {v.get('SQ_INSTS_VALU',0):.3e}"
        print(line); fh.write(line + "\n")
PY
