#!/usr/bin/env python3
"""Development aid: one wavefront per read (fused kernel) against the range-parallel mode on the headline batch — a single
launch, two contexts pipelined, and where the wave-cycles go in each (the build with phase timers)."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import mtr_amd
    from mtr_amd import synth
    n = int(sys.argv[2])
    reads = [c for _, c in synth.make_reads("headline2k", n, 2)]
    engs = [mtr_amd.Engine(), mtr_amd.Engine()]
    for e in engs:
        e.upload(reads)
    e = engs[0]
    ts = []
    for _ in range(3):
        t = time.perf_counter(); e.run(); ts.append((time.perf_counter() - t) * 1e3)
    kt = e.kernel_times_ms(); c = e.counters()
    engs[1].run()
    steps = 8
    t = time.perf_counter()
    engs[0].run_async()
    for s in range(steps):
        if s + 1 < steps:
            engs[(s + 1) & 1].run_async()
        engs[s & 1].wait()
    dt = (time.perf_counter() - t) / steps * 1e3
    print(f"   single launch (host call) {min(ts):.1f} ms, kernels {kt['k2_units']:.1f} ms; pipelined over two contexts {dt:.1f} ms/step = {n / dt:.1f} k reads/s")
    tot = c["cyc_total"]
    if tot:
        names = ("cyc_dp_fwd", "cyc_dp_tb", "cyc_dp_fwd_rev", "cyc_dp_tb_rev", "cyc_tab_build", "cyc_seeds", "cyc_walk", "cyc_polish", "cyc_revise_vote", "cyc_slot_copy", "cyc_k1_total")
        print("   phases (% of wave-cycles %.0f M):" % (tot / 1e6), ", ".join(f"{k[4:]} {100.0 * c[k] / tot:.1f}" for k in names))
    print("   counts:", {k: c[k] for k in ("dp_calls", "dp_cells", "revise_dp_calls", "revise_dp_cells", "memo_hits", "kmer_tables", "ranges_candidate", "ranges_executed", "records", "walk_steps")})
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "10000"
for lib in ("libmtr_hip.so", "libmtr_hip_prof.so"):
    for split in ("0", "1"):
        print(f"== {lib} MTR_SPLIT={split}", flush=True)
        env = dict(os.environ, MTR_SPLIT=split, MTR_LIB=os.path.join(ROOT, "mtr_amd", lib))
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", n], env=env)
