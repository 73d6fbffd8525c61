#!/usr/bin/env python3
"""Development aid (GPU box): phase timers (wave-cycles) of the revision kernel, four slots per wavefront (tests/dev/r5/old4_prof.so, built from the commit before) against eight."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import mtr_amd
    from mtr_amd import synth
    reads = [c for _, c in synth.make_reads("headline2k", 10000, 2)]
    e = mtr_amd.Engine(); e.upload(reads); e.run(); e.run()
    c = e.counters(); k = e.kernel_times_ms()
    print(json.dumps({"lib": os.path.basename(os.environ["MTR_LIB"]), "revise_quads_ms": k.get("kernel_mtr_k_revise_quads"), "chain_revisions_ms": k.get("chain_revisions"),
                      "fwd_rev_Gcyc": c["cyc_dp_fwd_rev"] / 1e9, "tb_rev_Gcyc": c["cyc_dp_tb_rev"] / 1e9, "fwd_dp2_Gcyc": c["cyc_dp_fwd"] / 1e9, "tb_all_Gcyc": c["cyc_dp_tb"] / 1e9,
                      "revise_vote_Gcyc": c["cyc_revise_vote"] / 1e9, "tb_refills": c["tb_refills"], "refill_Gcyc": c["cyc_tb_refill"] / 1e9, "revq_wave_Gcyc": c["reserved"] / 1e9, "revq_fill_Gcyc": c["prof47"] / 1e9, "revq_prepare_Gcyc": c["prof55"] / 1e9, "slot_copy_Gcyc": c["cyc_slot_copy"] / 1e9, "rev_bytes_per_cell": c["qpass_bytes_rev"] / max(1, c["qpass_cells_rev"])}))
else:
    for lib in ("tests/dev/r5/old4_prof.so", "mtr_amd/libmtr_hip_prof.so"):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, MTR_LIB=os.path.join(ROOT, lib)), capture_output=True, text=True)
        print(p.stdout.strip() or p.stderr[-400:], flush=True)
