#!/bin/bash
# Run ON THE GPU BOX: bash profiles/collect_split_pmc.sh — HBM bytes of the range phase and of the unit phase SEPARATELY:
# the range-parallel mode (MTR_SPLIT=1) runs them as different kernels (mtr_k_ranges / mtr_k_range_units / mtr_k_replay).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_split
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MTR_SPLIT=1 MTR_SPLIT_MAX_READS=1000000
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-latency"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $BENCH > "$OUT/bench_w.json" 2> "$OUT/w.err" || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $BENCH > "$OUT/bench_f.json" 2> "$OUT/f.err" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"bench": json.loads(open(out + "/bench_w.json").read().strip().splitlines()[-1])["work_per_launch"]}
for tag, sub in (("WRITE_SIZE", "write"), ("FETCH_SIZE", "fetch")):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == tag:
                a = acc[r["Kernel_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]) * 1024
    res[tag] = {k: {"launches": v[0], "bytes_per_launch": v[1] / v[0]} for k, v in acc.items()}
json.dump(res, open(out + "/split_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
