#!/bin/bash
# Run ON THE GPU BOX: bash profiles/collect_calib.sh — WRITE_SIZE / FETCH_SIZE of tests/dev/pmc_calib (known byte counts)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_calib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $ROOT/tests/dev/pmc_calib 4096 2 1 > "$OUT/calib_w.json" 2> "$OUT/w.err" || exit 1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $ROOT/tests/dev/pmc_calib 4096 2 1 > "$OUT/calib_f.json" 2> "$OUT/f.err" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
known = json.load(open(out + "/calib_w.json"))
res = {"known": known}
for tag, sub in (("WRITE_SIZE", "write"), ("FETCH_SIZE", "fetch")):
    rows = []
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == tag:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    res[tag] = [{"dispatch": d, "kernel": k, "KiB": v, "bytes": v * 1024} for d, k, v in rows]
json.dump(res, open(out + "/calib.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
