set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_c3_100
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MTR_STAGED=1 MTR_DEBUG=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python3 $ROOT/tests/dev/gpu_staged_prof.py 100 c3 > $OUT/trace.out 2> $OUT/trace.err
python3 - $OUT <<'PY'
import csv,glob,sys
for p in glob.glob(sys.argv[1]+'/trace/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(p)):
        if float(r['Percentage'])>0.3: print(f"  {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:8.2f} ms  {r['Percentage']}%")
PY
grep "staged:" $OUT/trace.err | tail -1
cat $OUT/trace.out
