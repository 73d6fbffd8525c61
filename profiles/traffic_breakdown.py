#!/usr/bin/env python3
"""HBM traffic of one launch, kernel by kernel, from a summary of profiles/collect_all.sh: python profiles/traffic_breakdown.py <summary.json> > out.json"""
import json
import sys

s = json.load(open(sys.argv[1]))
b = s.get("bench_under_rocprof", {})
out = {"note": "HBM traffic of one launch of the staged chain (10 000 headline reads), kernel by kernel: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
               "passes of bench.py --steps 4 --warmup 1 (profiles/collect_all.sh), bytes = KiB x 1024, per launch = mean per dispatch x dispatches / launches.  "
               "FETCH_SIZE reports half of the bytes of wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md): the corrected total doubles the read side.",
       "kernel_src_sha": s.get("kernel_src_sha"),
       "per_launch": {"fetch_bytes": s.get("k2_fetch_bytes_per_launch"), "write_bytes": s.get("k2_write_bytes_per_launch"), "raw_total": s.get("k2_hbm_bytes_per_launch"),
                      "total_fetch_x2": s.get("k2_hbm_bytes_per_launch_fetch_x2"),
                      "algorithmic_bytes_per_launch": b.get("algorithmic_bytes_per_launch") if isinstance(b, dict) else None},
       "what": {"mtr_k_revise_quads": "writes the cells of every vote DP and re-alignment that runs, TWO rows per byte (round 4), and the revisions' records; reads the tracebacks' windows and the read words",
                "mtr_k_dp2_quads": "writes one byte per cell pair of every computed two-parameter DP (both sets' flags in one byte), reads the windows of two tracebacks per DP",
                "mtr_k1_ranges": "writes the numerator rows of the <= 20 passes (D(i) per position, four or five positions per store since round 4, then D(i) - D(i+w) in place) + code arrays + DI/END/W; reads them back for the extraction",
                "mtr_k_walks / mtr_k_walks_k": "arena blocks (unit slots per (k, direction)), k-mer tables that do not fit the LDS, walk outputs",
                "mtr_k_select / mtr_k_polish": "revision records (3 KB each) into the arena; the polished unit and scores"},
       "kernels": {k: {"ms_under_profiler": r.get("ms_per_launch"), "fetch_bytes": r.get("fetch_bytes"), "write_bytes": r.get("write_bytes"),
                       "valu": r.get("valu"), "salu": r.get("salu")} for k, r in s.get("chain_per_launch", {}).items()}}
print(json.dumps(out, indent=1))
