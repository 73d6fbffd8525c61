"""Development aid: what a share of a long read costs (the model behind the host's plan, mtr_amd/host/pipeline.c RANGE_PHASE_SHARE).
For every bundled file: the whole read on one context, and share s of g for g = 2, 4, 8 (slowest share), plus the replay on the
reporting context.  python tests/dev/r3/share_cost.py [-p]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import mtr_amd
from tests import golden_util as gu
from tests.test_run_gloo import BUNDLED


def main():
    manhattan = "-p" not in sys.argv
    out = {}
    for name in BUNDLED:
        reads = [c for _, c in gu.read_fasta(gu.input_path(name))]
        L = sum(len(c) for c in reads)
        e = mtr_amd.Engine(manhattan=manhattan)
        e.upload(reads); e.run(); e.fetch()
        ts = []
        for _ in range(3):
            e.upload(reads)
            t0 = time.perf_counter(); e.run(); ts.append(time.perf_counter() - t0)
        row = {"bases": L, "whole_ms": round(min(ts) * 1e3, 2)}
        for g in (2, 4, 8):
            worst = 0.0
            blobs = []
            for s in range(g):
                e.upload(reads)
                t0 = time.perf_counter(); e.run_share(s, g); dt = time.perf_counter() - t0
                worst = max(worst, dt)
                blobs.append(e.export_candidates())
            e.upload(reads); e.run_share(0, g)
            t0 = time.perf_counter(); e.replay_candidates(blobs); rp = time.perf_counter() - t0
            row[f"share_of_{g}_ms"] = round(worst * 1e3, 2)
            row[f"replay_{g}_ms"] = round(rp * 1e3, 2)
            row[f"blob_{g}_bytes"] = sum(len(b) for b in blobs)
        e.close()
        out[name] = row
        print(name, json.dumps(row), flush=True)
    json.dump(out, open("gpurun_out/r3_share_cost%s.json" % ("" if manhattan else "_p"), "w"), indent=1)


if __name__ == "__main__":
    main()
