#!/usr/bin/env python3
"""Development aid (GPU box): the library's debug timeline of the command line's first batches (what do the first launch's allocations cost?)"""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("headline2k", 10000, 2)]
td = tempfile.mkdtemp(); fa = os.path.join(td, "r.fa")
synth.write_fasta(fa, [(str(i), reads[i % len(reads)]) for i in range(40000)])
for rep in range(2):
    p = subprocess.run(["mtr_amd/host/mTR", fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MTR_DEBUG="1", MTR_HOST_TIMING="1"))
    keep = [ln for ln in p.stderr.decode().splitlines() if ("[host" in ln or "chain buffers" in ln or "scratch ready" in ln or "mtr_create" in ln or "launch of" in ln or "upload_batch: copies" in ln)]
    print("\n".join(keep[:40])); print("----")
