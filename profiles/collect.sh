#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/collect.sh <tag>
# Three separate rocprofv3 runs of the same bench command, as MI355X_MICROARCH.md prescribes:
#   1. --kernel-trace --stats            -> per-kernel durations
#   2. --pmc FETCH_SIZE                  -> HBM read bytes   (own pass: FETCH_SIZE takes 3 of the 4 TCC slots)
#   3. --pmc WRITE_SIZE                  -> HBM write bytes
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py condenses it into profiles/.
set -o pipefail
TAG=${1:-latest}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t --output-format csv -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || exit 1
echo "trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o f --output-format csv -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err" || exit 1
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o w --output-format csv -- $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.err" || exit 1
echo "write done"
python3 "$ROOT/profiles/summarize.py" "$OUT" "$TAG"
