import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mtr_amd
from mtr_amd import synth
reads = [c for _, c in synth.make_reads("headline2k", 3000, 2)]
eng = mtr_amd.Engine(); eng.upload(reads); eng.run()
c = eng.counters(); print({k: c[k] for k in ("global_tables", "kmer_tables", "kmer_lookups", "dp_calls")})
eng.upload([reads[503]]); eng.run(); c = eng.counters(); print("read 503:", {k: c[k] for k in ("global_tables", "kmer_tables", "kmer_lookups", "dp_calls", "cyc_walk", "cyc_total")}, eng.kernel_times_ms())
