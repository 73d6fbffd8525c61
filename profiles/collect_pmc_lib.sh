#!/bin/bash
# Run ON THE GPU BOX: bash profiles/collect_pmc_lib.sh <tag> <lib.so> [waves_per_cu] — WRITE_SIZE / FETCH_SIZE of bench.py with
# the given library build (A/B experiments: e.g. a build with 256 VGPRs per wavefront = no register spills).
set -o pipefail
TAG=$1; LIB=$2; WPC=${3:-16}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MTR_LIB=$ROOT/$LIB MTR_K2_WAVES_PER_CU=$WPC
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $BENCH > "$OUT/bench_w.json" 2> "$OUT/w.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $BENCH > "$OUT/bench_f.json" 2> "$OUT/f.err" || exit 1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
b = json.loads(open(out + "/bench_w.json").read().strip().splitlines()[-1])
res = {"tag": tag, "reads_per_s_under_rocprof": b["value"], "kernels_ms_alone": b.get("kernels_ms_alone")}
for name, sub in (("WRITE_SIZE", "write"), ("FETCH_SIZE", "fetch")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "mtr_k" in r["Kernel_Name"]:
                a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024
    res[name] = {k: {"launches": v[0], "GB_per_launch": v[1] / v[0] / 1e9} for k, v in acc.items()}
json.dump(res, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
