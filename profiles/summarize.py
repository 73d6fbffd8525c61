#!/usr/bin/env python3
"""Condense one profiles/collect_all.sh run (gpurun_out/prof_<tag>/) into gpurun_out/prof_<tag>/summary.json.

Copy summary.json to profiles/<round>_<tag>_summary.json and to profiles/pmc_latest.json (bench.py reads roofline.traffic and
roofline.issue from the latter, and only while the kernel sources still have the sha recorded here), and the kernel_stats
csv next to it.
FETCH_SIZE / WRITE_SIZE: rocprofv3 reports KiB -> bytes = value * 1024.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads -> `k2_hbm_bytes_per_launch_fetch_x2` doubles the
read side as the guide prescribes; the raw sum is kept beside it.  WRITE_SIZE is exact for streaming stores.
VALU issue utilisation = SQ_INSTS_VALU x 2 cycles (a wave64 VALU instruction occupies a SIMD-32 for two passes)
/ (SIMDs x launch duration x clock); lanes doing reference work = algorithmic cell-ops / (SQ_INSTS_VALU x 64).
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = "mtr_k"          # every kernel of the library starts with this


def counter_rows(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(p) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def per_kernel(rows):
    """{kernel: {counter: mean per launch}}"""
    acc = {}
    for r in rows:
        k = r.get("Kernel_Name", "?").split("(")[0].replace("void ", "")
        if DOMINANT not in k:
            continue
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], {})
        a[r["Dispatch_Id"]] = a.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} | {"_launches": max(len(v) for v in cs.values())} for k, cs in acc.items()}


def kernel_sources_sha():
    # the sources of the kernels and the flags they are built with (the same function as bench.py's)
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mtr_amd", "csrc")
    for f in ("device_util.hip.inc", "dp_wrap.hip.inc", "dp_quad.hip.inc", "k1_ranges.hip.inc", "k2_units.hip.inc", "k3_staged.hip.inc", "mtr_common.h",
              "min_missing_table.h", "mtr_abi.hip"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "mtr_amd", "build.py"), "rb").read())
    return h.hexdigest()[:16]


# one launch = the staged chain: every kernel of the library except the per-read kernel (bench.py runs it ONCE, outside the timed
# region, for the reference's work counters) and the wire-form kernels of the fetch
NOT_CHAIN = ("mtr_k_reads", "mtr_k_wire")


def main():
    out, tag = sys.argv[1], sys.argv[2]
    config = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else None
    s = {"tag": tag, "config": config or "headline2k", "kernel_src_sha": kernel_sources_sha(),
         "command": "bench.py --steps 4 --warmup 1 --cpu-sample 0 --no-latency --no-cli" + (f" --config {config}" if config else "") + " under rocprofv3 (one pass per counter group)"}
    stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
    dur_ms = {}
    if stats:
        with open(stats[0]) as fh:
            s["kernel_stats"] = [dict(r) for r in csv.DictReader(fh)]
        for r in s["kernel_stats"]:
            if DOMINANT in r.get("Name", ""):
                dur_ms[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"]) / 1e6
    try:
        with open(os.path.join(out, "bench_trace.json")) as fh:
            b = json.loads(fh.read().strip().splitlines()[-1])
        s["bench_under_rocprof"] = {"value": b["value"], "ms_per_step": b["ms_per_step"], "kernels_ms": b["kernels_ms"],
                                    "kernels_ms_alone": b.get("kernels_ms_alone"), "algorithmic_bytes_per_launch": b["roofline"]["hbm"]["algorithmic_bytes_per_launch"],
                                    "cells_per_launch": b["roofline"]["cells_per_launch"], "reads_per_launch": b.get("config", {}).get("reads_per_step")}
    except Exception as e:      # noqa: BLE001
        s["bench_under_rocprof"] = f"unreadable: {e}"
    groups = {g: per_kernel(counter_rows(os.path.join(out, g))) for g in ("fetch", "write", "sq1", "sq2", "lds")}
    s["counters_per_launch"] = groups
    chain = [k for k in dur_ms if not k.startswith(NOT_CHAIN)]
    calls0 = {r["Name"].split("(")[0].replace("void ", ""): int(r["Calls"]) for r in s.get("kernel_stats", [])}
    dom = max(chain, key=lambda k: dur_ms[k] * calls0.get(k, 1)) if chain else None      # by time per LAUNCH (the two-pass chain runs most kernels twice)
    s["dominant_kernel"] = dom
    if dom:
        def launches(grp):            # launches of the chain seen by a counter pass = dispatches of a kernel that runs ONCE per launch (the two-pass chain runs most kernels twice)
            g = groups.get(grp, {})
            return max(1, int((g.get("mtr_k_replay") or g.get("mtr_k_items") or g.get("mtr_k_gather") or {}).get("_launches", 1)))

        def total(grp, c):            # per launch of the chain: sum over its kernels of (mean per dispatch x dispatches) / launches
            t, seen = 0.0, False
            for k, cs in groups.get(grp, {}).items():
                if k.startswith(NOT_CHAIN) or c not in cs:
                    continue
                t += cs[c] * cs["_launches"] / launches(grp); seen = True
            return t if seen else None

        table = {}
        calls = {r["Name"].split("(")[0].replace("void ", ""): int(r["Calls"]) for r in s.get("kernel_stats", [])}
        n_launch_trace = max(1, calls.get("mtr_k_replay", calls.get("mtr_k_items", calls.get("mtr_k_gather", 1))))
        for k in sorted(chain, key=lambda k: -dur_ms[k] * calls.get(k, 1)):
            row = {"ms_per_launch": dur_ms[k] * calls.get(k, 1) / n_launch_trace}
            for grp, c, key, mul in (("sq2", "SQ_INSTS_VALU", "valu", 1.0), ("sq2", "SQ_INSTS_SALU", "salu", 1.0), ("fetch", "FETCH_SIZE", "fetch_bytes", 1024.0),
                                     ("write", "WRITE_SIZE", "write_bytes", 1024.0), ("sq1", "SQ_WAVE_CYCLES", "wave_cycles", 4.0)):
                cs = groups.get(grp, {}).get(k, {})
                if c in cs:
                    row[key] = cs[c] * cs["_launches"] / launches(grp) * mul
            table[k] = row
        s["chain_per_launch"] = table
        s["chain_ms_per_launch_kernels_one_at_a_time"] = sum(r["ms_per_launch"] for r in table.values())
        f, w = total("fetch", "FETCH_SIZE"), total("write", "WRITE_SIZE")
        if f is not None and w is not None:
            s["k2_fetch_bytes_per_launch"] = f * 1024.0
            s["k2_write_bytes_per_launch"] = w * 1024.0
            s["k2_hbm_bytes_per_launch"] = (f + w) * 1024.0
            s["k2_hbm_bytes_per_launch_fetch_x2"] = (2 * f + w) * 1024.0
        valu, salu = total("sq2", "SQ_INSTS_VALU"), total("sq2", "SQ_INSTS_SALU")
        if valu:
            simds, clock = 1024, 2.4e9
            s["SQ_INSTS_VALU_per_launch"] = valu
            s["valu_issue_utilisation"] = valu * 2.0 / (simds * s["chain_ms_per_launch_kernels_one_at_a_time"] * 1e-3 * clock)
            s["insts_valu_plus_salu"] = valu + (salu or 0.0)
            cells = s["bench_under_rocprof"].get("cells_per_launch") if isinstance(s["bench_under_rocprof"], dict) else None
            if cells:
                s["lanes_doing_reference_work"] = cells * 7.0 / (valu * 64.0)
                s["cells_per_instruction"] = cells / (valu + (salu or 0.0))
        bc, ia = total("lds", "SQ_LDS_BANK_CONFLICT"), total("lds", "SQ_LDS_IDX_ACTIVE")
        if bc is not None and ia:
            s["lds_bank_conflict_share_of_lds_cycles"] = bc / ia
            wc = total("sq1", "SQ_WAVE_CYCLES")
            if wc:
                s["lds_array_busy_share_of_wave_cycles"] = ia / (4.0 * wc)
    with open(os.path.join(out, "summary.json"), "w") as fh:
        json.dump(s, fh, indent=1)
    print(json.dumps({k: v for k, v in s.items() if k not in ("kernel_stats", "counters_per_launch", "chain_per_launch")}))


if __name__ == "__main__":
    main()
