#!/bin/bash
# Development aid, ON THE GPU BOX: SQ counters per kernel of a staged-mode launch (the phases of the unit search as separate
# kernels), to see which phase waits and which one issues.  bash tests/dev/prof_staged_pmc.sh [n_reads] [config]
set -o pipefail
N=${1:-10000}; CFG=${2:-headline2k}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_stpmc_${N}_$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MTR_STAGED=1
run() { name=$1; shift; timeout -k 10 300 rocprofv3 "$@" -d "$OUT/$name" -o $name --output-format csv -- python3 $ROOT/tests/dev/gpu_staged_prof.py $N $CFG > "$OUT/$name.out" 2> "$OUT/$name.err" || { tail -5 "$OUT/$name.err"; exit 1; }; echo "$name done"; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_FLAT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "profiles"))
import summarize as S
out = sys.argv[1]
dur = {}
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(p)):
        dur[r["Name"].split("(")[0].replace("void ", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)
g = {n: S.per_kernel(S.counter_rows(os.path.join(out, n))) for n in ("sq1", "sq2", "sq3")}
ks = sorted(dur, key=lambda k: -dur[k][0] * dur[k][1])
print(f"{'kernel':28s} {'calls':>5s} {'ms':>8s} {'waveGcyc':>9s} {'wait%':>6s} {'active%':>7s} {'VALU M':>8s} {'SALU M':>8s} {'LDS M':>7s} {'RD M':>7s} {'WR M':>7s} {'cyc/ins':>7s} {'wait/rd':>8s}")
for k in ks:
    a, b = g["sq1"].get(k, {}), g["sq2"].get(k, {})
    if not a: continue
    wc = a.get("SQ_WAVE_CYCLES", 0) * 4; wt = a.get("SQ_WAIT_INST_ANY", 0) * 4; ac = a.get("SQ_ACTIVE_INST_ANY", 0) * 4
    v, s_, l, rd, wr = (b.get(x, 0) for x in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"))
    print(f"{k[:28]:28s} {dur[k][0]:5d} {dur[k][1]:8.2f} {wc / 1e9:9.2f} {100 * wt / max(wc, 1):6.1f} {100 * ac / max(wc, 1):7.1f} {v / 1e6:8.1f} {s_ / 1e6:8.1f} {l / 1e6:7.1f} {rd / 1e6:7.1f} {wr / 1e6:7.1f} {wc / max(v + s_, 1):7.1f} {wt / max(rd, 1):8.0f}")
print({k: v for k, v in g["sq3"].items()})
PY
