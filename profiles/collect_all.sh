#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root:  bash profiles/collect_all.sh <tag> [config]
# Every rocprofv3 pass of the same bench command, each on its own as MI355X_MICROARCH.md prescribes (counters never together
# with tracing domains, FETCH_SIZE and WRITE_SIZE in separate passes: they do not fit the TCC's four slots together):
#   1. --kernel-trace --stats   per-kernel durations
#   2. --pmc FETCH_SIZE         HBM read bytes (reports 1/2 of wide reads on gfx950: x 2 in the summary)
#   3. --pmc WRITE_SIZE         HBM write bytes
#   4./5. --pmc SQ_*            issue / wait / instruction counters (8 SQ slots per pass)
#   6. --pmc SQ_LDS_*           LDS bank-conflict counters
# Raw output: gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py condenses it into gpurun_out/prof_<tag>/summary.json,
# which is then copied to profiles/<round>_<tag>_summary.json and profiles/pmc_latest.json (bench.py reads the latter).
set -o pipefail
TAG=${1:-latest}
CONFIG=${2:-}                       # a secondary config (c3, c2): bench.py --config <it>; the summary then goes to profiles/pmc_<config>.json, which bench.py reads for THAT config only
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# bench.py asks for 8 hardware queues (an RCCL communicator's streams must not push the two context streams onto one queue);
# under rocprofv3 that setting serialises the two contexts' launches (92 ms a step), the runtime's default of 4 does not (56 ms)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-4}
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency --no-cli --no-secondary --no-upload-leg"
if [ -n "$CONFIG" ]; then BENCH="$BENCH --config $CONFIG"; fi
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" -d "$OUT/$name" -o $name --output-format csv -- $BENCH > "$OUT/bench_$name.json" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }; echo "$name done"; }
run trace --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run lds --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS
python3 "$ROOT/profiles/summarize.py" "$OUT" "$TAG" "$CONFIG"
