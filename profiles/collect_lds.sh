#!/bin/bash
# Run ON THE GPU BOX from the repo root: bash profiles/collect_lds.sh <tag>
# LDS counters of the per-read kernel (north_star: "LDS-bank counters against gfx950 peak"); one --pmc pass, no tracing.
#   SQ_LDS_BANK_CONFLICT = extra LDS-array cycles lost to bank conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles
#   (MI355X_MICROARCH.md, LDS), SQ_LDS_ADDR_CONFLICT / SQ_LDS_UNALIGNED_STALL = the other two stall sources.
set -o pipefail
TAG=${1:-latest}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lds_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency"
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d "$OUT/p1" -o a --output-format csv -- $BENCH > "$OUT/bench1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
echo "lds pass done"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
acc = {}
for p in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        k = r.get("Kernel_Name", "")
        if not k.startswith("void mtr_k") and not k.startswith("mtr_k"):
            continue
        kn = k.split("(")[0]
        a = acc.setdefault(kn, {}).setdefault(r["Counter_Name"], {})
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
s = {}
for kn, cs in acc.items():
    d = {c: sum(v.values()) / len(v) for c, v in cs.items()}
    d["_launches"] = max(len(v) for v in cs.values())
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["bank_conflict_frac_of_lds_cycles"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("SQ_WAVE_CYCLES"):
        # SQ_WAVE_CYCLES is reported in units of 4 cycles (profiles/README.md); LDS array busy share of wave time
        d["lds_idx_active_per_wave_cycle"] = d.get("SQ_LDS_IDX_ACTIVE", 0.0) / (4.0 * d["SQ_WAVE_CYCLES"])
    s[kn] = d
json.dump(s, open(os.path.join(out, "lds_summary.json"), "w"), indent=1)
print(json.dumps(s))
PY
