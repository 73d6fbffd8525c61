#!/bin/bash
# Development aid, ON THE GPU BOX: config 4 (100 000 mixed reads) through mTR -c -g N with the library's stamps: how often does a context's scratch grow?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from mtr_amd import synth
synth.write_fasta('/tmp/c4.fa', synth.make_reads('c4', 100000, 4))
PY
for g in 1 2; do
  MTR_DEBUG=1 MTR_HOST_TIMING=1 mtr_amd/host/mTR -c -g $g /tmp/c4.fa > /dev/null 2> gpurun_out/c4_debug_g$g.err
  echo "g=$g"; grep "scratch ready\|chain buffers ready" gpurun_out/c4_debug_g$g.err | awk '{print $2, $3, $0}' | cut -c1-160 | head -40; grep "^\[host\]" gpurun_out/c4_debug_g$g.err
done
