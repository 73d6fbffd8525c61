#!/usr/bin/env python3
"""Condense one profiles/collect.sh run (gpurun_out/prof_<tag>/) into gpurun_out/prof_<tag>/summary.json.

Copy summary.json to profiles/<round>_<tag>_summary.json (and to profiles/pmc_latest.json, which bench.py reads for
roofline.traffic) and the kernel_stats csv next to it.
FETCH_SIZE / WRITE_SIZE: rocprofv3 reports KiB -> bytes = value * 1024.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced 16 B/lane streaming reads; this kernel's reads are byte- and
dword-granular (traceback windows, k-mer tables, packed words) = an uncalibrated pattern, so both the raw and the
2x figure are given; `traffic` uses the raw one.  WRITE_SIZE is exact for streaming stores.
"""
import csv
import glob
import json
import os
import sys


def counter_rows(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def per_kernel(rows, counter):
    acc = {}
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        k = r.get("Kernel_Name", "?")
        did = r.get("Dispatch_Id", "0")
        acc.setdefault(k, {}).setdefault(did, 0.0)
        acc[k][did] += float(r["Counter_Value"])
    return {k: {"launches": len(v), "bytes_per_launch": sum(v.values()) * 1024.0 / len(v)} for k, v in acc.items()}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    s = {"tag": tag, "command": "bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency under rocprofv3 (three separate runs)"}
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        with open(stats[0]) as fh:
            s["kernel_stats"] = [dict(r) for r in csv.DictReader(fh)]
    try:
        with open(os.path.join(out, "bench_trace.json")) as fh:
            b = json.loads(fh.read().strip().splitlines()[-1])
        s["bench_under_rocprof"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "kernels_ms": b["kernels_ms"],
                                    "kernels_ms_alone": b.get("kernels_ms_alone"), "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"]}
    except Exception as e:      # noqa: BLE001
        s["bench_under_rocprof"] = f"unreadable: {e}"
    f = per_kernel(counter_rows(os.path.join(out, "fetch")), "FETCH_SIZE")
    w = per_kernel(counter_rows(os.path.join(out, "write")), "WRITE_SIZE")
    s["fetch"] = f
    s["write"] = w
    for k in f:
        if "mtr_k_reads" in k and k in w:
            s["k2_fetch_bytes_per_launch"] = f[k]["bytes_per_launch"]
            s["k2_write_bytes_per_launch"] = w[k]["bytes_per_launch"]
            s["k2_hbm_bytes_per_launch"] = f[k]["bytes_per_launch"] + w[k]["bytes_per_launch"]
            s["k2_hbm_bytes_per_launch_fetch_x2"] = 2 * f[k]["bytes_per_launch"] + w[k]["bytes_per_launch"]
    with open(os.path.join(out, "summary.json"), "w") as fh:
        json.dump(s, fh, indent=1)
    print(json.dumps({k: v for k, v in s.items() if k.startswith("k2_")}))


if __name__ == "__main__":
    main()
