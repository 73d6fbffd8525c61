#!/bin/bash
# Development aid, ON THE GPU BOX: config 5's 15 files through mTR -c -p -g 1 with the host's and the library's stamps (where do the seconds go?)
cd $GRAFT_REPO_ROOT
F=""; for n in 3_5 3_10 3_20 3_50 5_10 5_20 5_50 10_20 10_50 20_50 2_5_10_20_set 2_5_10_20_50_100_200_set worm_chrI worm_chrII_1 worm_chrII_2; do F="$F tests/golden/inputs/$n.fasta"; done
for i in 1 2; do
  /usr/bin/env time -f "wall %e s" mtr_amd/host/mTR -p -g 1 $F > /dev/null 2> gpurun_out/c5_plain_$i.err; tail -1 gpurun_out/c5_plain_$i.err
done
MTR_DEBUG=1 MTR_HOST_TIMING=1 mtr_amd/host/mTR -c -p -g 1 $F > /dev/null 2> gpurun_out/c5_debug.err
grep -n "host +\|\[host\]" gpurun_out/c5_debug.err | head -40
python3 - <<'PY'
import re
prev=None
for ln in open('gpurun_out/c5_debug.err', errors='replace'):
    m=re.search(r'\+\s*([0-9.]+)\s*s', ln)
    if m:
        t=float(m.group(1))
        if prev is not None and t-prev>0.15: print('GAP %.3f s before: %s' % (t-prev, ln.strip()[:200]))
        prev=t
PY
MTR_BENCH_CONTEXTS=2 true
for i in 1 2 3; do ( time mtr_amd/host/mTR -p tests/golden/inputs/10_20.fasta > /dev/null ) 2>&1 | grep real; done
