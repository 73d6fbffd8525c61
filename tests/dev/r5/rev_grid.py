#!/usr/bin/env python3
"""Development aid (GPU box): mtr_k_revise_quads alone (HIP events, mtr_get_kernel_times id 8) for several grid sizes: python tests/dev/r5/rev_grid.py [n_reads] [config]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3:
    import numpy as np, torch
    torch.cuda.init()
    import mtr_amd
    from mtr_amd import synth
    reads = [c for _, c in synth.make_reads(sys.argv[2], int(sys.argv[1]), 2)]
    e = mtr_amd.Engine(); e.upload(reads)
    ts = []
    for _ in range(4):
        e.run(); ts.append(e.kernel_times_ms())
    c = e.counters()
    k = ts[-1]
    print(json.dumps({"knob": os.environ.get("MTR_REV_WAVES_PER_CU"), "revise_quads_ms": min(t.get("kernel_mtr_k_revise_quads", 0) for t in ts[1:]), "dp2_quads_ms": min(t.get("kernel_mtr_k_dp2_quads", 0) for t in ts[1:]),
                      "launch_ms": min(t["k2_units"] for t in ts[1:]), "bytes_per_cell": c["qpass_bytes_rev"] / max(c["qpass_cells_rev"], 1), "records": c["records"]}))
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "10000"; cfg = sys.argv[2] if len(sys.argv) > 2 else "headline2k"
    for knob in ("0", "16", "12", "10", "8", "6", "4"):
        env = dict(os.environ, MTR_REV_WAVES_PER_CU=knob)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), n, cfg, "child"], env=env, capture_output=True, text=True)
        print(p.stdout.strip() or p.stderr[-300:], flush=True)
