#!/bin/bash
# Development aid, ON THE GPU BOX: 1 200 reads of config 3's shape (42 kb) through the command line in batches of 100 reads (MTR_CHUNK_BYTES): wall clock and
# stdout against the same file in the default batches of 24 MiB - the host pipeline's six batches in flight for long reads (pipeline.c)
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
from mtr_amd import synth
synth.write_fasta('/tmp/c3_1200.fa', synth.make_reads('c3', 1200, 3))
PY
for i in 1 2; do
  ( time MTR_HOST_TIMING=1 MTR_CHUNK_BYTES=4300000 mtr_amd/host/mTR /tmp/c3_1200.fa > /tmp/c3_small_batches.out 2> /tmp/c3_small.err ) 2>&1 | grep real
  grep "^\[host\]" /tmp/c3_small.err; grep -c "device context created" /tmp/c3_small.err
done
( time mtr_amd/host/mTR /tmp/c3_1200.fa > /tmp/c3_default_batches.out 2>/dev/null ) 2>&1 | grep real
cmp /tmp/c3_small_batches.out /tmp/c3_default_batches.out && echo "stdout identical ($(wc -l < /tmp/c3_default_batches.out) lines)"
