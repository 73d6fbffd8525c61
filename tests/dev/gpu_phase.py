#!/usr/bin/env python3
"""Where K2's cycles go (development aid): phase timers summed over waves, for a single read and a batch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MTR_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "mtr_amd", "libmtr_hip_prof.so"))   # the build with the phase timers
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reads = [c for _, c in synth.make_reads("headline2k", n, 2)]
eng = mtr_amd.Engine()
for label, rs in (("single read", reads[:1]), (f"{n} reads", reads)):
    eng.upload(rs); eng.run(); eng.run()
    c = eng.counters(); kt = eng.kernel_times_ms()
    tot = c["cyc_total"]
    print(f"== {label}: K2 {kt['k2_units']:.1f} ms, K1 {kt['k1_ranges']:.1f} ms, total wave-cycles {tot/1e6:.1f} M (s_memtime ticks)")
    for k in ("cyc_dp_fwd", "cyc_dp_tb", "cyc_dp_fwd_rev", "cyc_dp_tb_rev", "cyc_tab_build", "cyc_seeds", "cyc_walk", "cyc_polish", "cyc_revise_vote", "cyc_slot_copy"):
        print(f"   {k:18s} {c[k]/1e6:10.1f} M  {100.0*c[k]/max(tot,1):5.1f} %")
    k1 = c["cyc_k1_total"]
    print(f"   K1 wave-cycles {k1/1e6:.1f} M: codes {100*c['cyc_k1_codes']/max(k1,1):.1f} %  passes {100*c['cyc_k1_passes']/max(k1,1):.1f} %  extract {100*c['cyc_k1_extract']/max(k1,1):.1f} %  dedup+out {100*c['cyc_k1_dedup']/max(k1,1):.1f} %")
    excl = ("cyc_dp_fwd", "cyc_dp_tb", "cyc_dp_fwd_rev", "cyc_dp_tb_rev", "cyc_tab_build", "cyc_seeds", "cyc_walk", "cyc_polish",
            "cyc_revise_vote", "cyc_slot_copy", "cyc_k1_total")      # disjoint phases (polish contains one table build: counted twice, small)
    rest = tot - sum(c[k] for k in excl)
    print(f"   {'range phase (K1)':18s} {k1/1e6:10.1f} M  {100.0*k1/max(tot,1):5.1f} %")
    print(f"   {'other':18s} {rest/1e6:10.1f} M  {100.0*rest/max(tot,1):5.1f} %")
    print(f"   traceback refills: {c['tb_refills']} taking {c['cyc_tb_refill']/1e6:.1f} M cycles = {c['cyc_tb_refill']/max(c['tb_refills'],1):.0f} per refill; memo hits {c['memo_hits']}, tables skipped {c['tables_skipped']}")
    print(f"   walk steps {c['walk_steps']}, of which {c['walk_slow_steps']} through the general look-ahead taking {c['cyc_walk_slow']/1e6:.1f} M cycles")
    print(f"   walk calls {c['walk_calls']} ({c['walk_closed']} closed a cycle), cycles inside walk_fast {c['cyc_walk_fast']/1e6:.1f} M")
    print(f"   revisions {c['prof44']}, unit unchanged by the votes in {c['prof43']}, accepted: round 0 {c['prof45']}, round 1 {c['prof46']}")
    print(f"   two-parameter forward passes of units <= 16 bases: {c['prof47']/1e6:.1f} M cycles")
    print("   counts:", {k: c[k] for k in ("dp_calls", "dp_rows", "dp_cells", "traceback_steps", "kmer_tables", "kmer_lookups", "ranges_executed")})
