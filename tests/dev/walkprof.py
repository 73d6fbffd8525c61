import sys, os
sys.path.insert(0, "/root/repo")
import mtr_amd
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("headline2k", 3000, 2)]
eng = mtr_amd.Engine()
for idx in (932, 503, 1301):
    eng.upload([reads[idx]]); eng.run(); c = eng.counters()
    print(idx, "walk", c["cyc_walk"]/1e6, "steps", c["reserved"], "prepass", c["cyc_polish"]/1e6, "scan", c["cyc_revise_vote"]/1e6, "probe", c["cyc_slot_copy"]/1e6, "lookups", c["kmer_lookups"], "tables", c["kmer_tables"], "tab_build", c["cyc_tab_build"]/1e6, "total", c["cyc_total"]/1e6)
