#!/usr/bin/env python3
"""Phase breakdown on the config-3 shape (42 kb reads, unit 200 x 200) - development aid."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MTR_LIB", os.path.join(ROOT, "mtr_amd", "libmtr_hip_prof.so"))
import mtr_amd
from mtr_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reads = [c for _, c in synth.make_reads("c3", n, 3)]
eng = mtr_amd.Engine()
eng.upload(reads); eng.run(); eng.run()
c = eng.counters(); kt = eng.kernel_times_ms()
tot = c["cyc_total"]
print(f"{n} reads of ~42 kb: kernel {kt['k2_units']:.1f} ms, wave-cycles {tot/1e6:.1f} M")
for k in ("cyc_dp_fwd", "cyc_dp_tb", "cyc_dp_fwd_rev", "cyc_dp_tb_rev", "cyc_tab_build", "cyc_seeds", "cyc_walk", "cyc_walk_slow", "cyc_polish", "cyc_revise_vote", "cyc_slot_copy", "cyc_k1_total", "cyc_k1_passes", "cyc_k1_extract", "cyc_k1_dedup", "cyc_tb_refill"):
    print(f"   {k:18s} {c[k]/1e6:10.1f} M  {100.0*c[k]/max(tot,1):5.1f} %")
print({k: c[k] for k in ("dp_calls", "dp_rows", "dp_cells", "revise_dp_calls", "revise_dp_cells", "memo_hits", "memo_cells", "traceback_steps", "tb_refills", "kmer_tables", "tables_skipped", "global_tables", "walk_calls", "walk_steps", "walk_slow_steps", "ranges_executed", "records")})
