#!/usr/bin/env python3
"""Level-2 captures (G2 = every search_De_Bruijn_graph call that found a unit or ran a search, consensus.c:507-582) of the
UNMODIFIED reference for two of the golden inputs, one reference process per read as in make_golden.py.  They pin the GPU's
unit search stage by stage (tests/test_gpu_stages.py).  Run where /root/reference exists:

  python tests/golden/make_golden_l2.py      -> tests/golden/<case>.default.l2.jsonl.gz (the G2 lines, tagged with the read index)
"""
import gzip
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_isolated  # noqa: E402

CASES = ["3_5", "synth_2k"]


def main():
    for name in CASES:
        fa = os.path.join(HERE, "inputs", name + (".fasta" if os.path.exists(os.path.join(HERE, "inputs", name + ".fasta")) else ".fa"))
        _, cap = ref_isolated.run_reference_isolated(fa, [], level=2)
        rd, out = -1, []
        for line in cap.decode().splitlines():
            ev = json.loads(line)
            if ev["t"] == "G1":
                rd += 1
            elif ev["t"] == "G2":
                rr = ev["rr"]
                out.append(json.dumps({"rd": rd, "qs": ev["qs"], "qe": ev["qe"], "k": ev["k"], "found": ev["found"], "period": rr["period"],
                                       "rep_start": rr["rep_start"], "rep_end": rr["rep_end"], "repeat_len": rr["repeat_len"], "copies": rr["copies"],
                                       "mat": rr["mat"], "mis": rr["mis"], "ins": rr["ins"], "del": rr["del"], "unit": rr["unit"]}, separators=(",", ":")))
        path = os.path.join(HERE, f"{name}.default.l2.jsonl.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as fh:
            fh.write(("\n".join(out) + "\n").encode())
        print(name, len(out), "G2 lines ->", os.path.relpath(path, ROOT), os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
