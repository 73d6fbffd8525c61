# Development aid, ON THE GPU BOX: bash tests/dev/r6/slots_and_lone.sh <tag> - per-kernel durations of lone launches (two passes / one pass) and SQ_WAVE_CYCLES per kernel of the pipelined bench
export TAG=${1:-r06x}
set -o pipefail
cd $GRAFT_REPO_ROOT
bash tests/dev/r4/trace_lone.sh 10000 headline2k ${TAG}_two > gpurun_out/${TAG}_lone_two.txt 2>&1
MTR_TWO_PASS=0 bash tests/dev/r4/trace_lone.sh 10000 headline2k ${TAG}_one > gpurun_out/${TAG}_lone_one.txt 2>&1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp GPU_MAX_HW_QUEUES=4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq1 -o sq1 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency --no-cli --no-secondary --no-upload-leg > $OUT/bench_sq1.json 2> $OUT/sq1.err
echo sq1 rc=$?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,os
acc={}
for p in glob.glob('gpurun_out/prof_'+os.environ['TAG']+'/sq1/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if r['Counter_Name']!='SQ_WAVE_CYCLES': continue
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        a=acc.setdefault(k,{})
        a[r['Dispatch_Id']]=a.get(r['Dispatch_Id'],0.0)+float(r['Counter_Value'])
tot=0
lines=[]
for k,v in acc.items():
    n=len(v); s=sum(v.values())*4
    lines.append((s/ n, n, k))
# per launch of the chain: total over dispatches / number of staged launches (k1_ranges dispatch count)
nl=len(acc.get('mtr_k1_ranges',{1:1}))
with open('gpurun_out/'+os.environ['TAG']+'_wave_cycles.txt','w') as fh:
    for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1].values())):
        s=sum(v.values())*4/nl
        tot+=s if 'mtr_k_reads' not in k and 'wire' not in k else 0
        fh.write(f"{k[:40]:40s} dispatches {len(v):4d}  wave-cycles per chain launch {s/1e9:9.3f} G\n")
    fh.write(f"chain launches {nl}; sum over the chain {tot/1e9:.2f} G\n")
print(open('gpurun_out/'+os.environ['TAG']+'_wave_cycles.txt').read())
PY
