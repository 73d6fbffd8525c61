set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_icache
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d "$OUT/p1" -o a --output-format csv -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency > "$OUT/bench1.json" 2> "$OUT/p1.err" || { tail -5 $OUT/p1.err; exit 1; }
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
print(json.dumps(s))
PY
