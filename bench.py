#!/usr/bin/env python3
"""bench.py — reads/s of the per-read hot path (handle_one_read) on MI355X, next to CPU mTR.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path (the per-read kernel mtr_k_reads: candidate ranges, unit search, wrap-around DPs,
revision; mtr_run_resident) over one batch of synthetic reads that is already resident in HBM (2 bit/base).  Workload = the configuration BASELINE.json's
metric is quoted on: 10 000 synthetic Nanopore-error reads of ~2 kb (unit 100 x 10 copies, 500-base flanks;
mtr_amd.synth "headline2k") per GPU.  With N GPUs every rank holds its own 10 000 reads (weak scaling, no
data-path collective); the step ends with the RCCL gather of the per-read record tables to rank 0 (the one
exchange step of the path).  Rank 0 prints ONE JSON line.

roofline: the only kernel is mtr_k_reads.  `achieved` = algorithmic HBM bytes of one launch (SURVEY.md §8d:
B_alg = ceil(L/4) + 576 R + sum over the REFERENCE's DPs of ceil(cells/2), every DP counted as spilled because this
build keeps all traceback cells in HBM-backed scratch; the DPs the kernel answers from its memo are part of the
reference's work and are counted) / the kernel's average duration measured with HIP events on the launch stream.
The path is NOT HBM-bound (row-serial integer recurrence: instruction issue + cross-lane scan latency): the
VALU-side figure is reported in `roofline.valu` from the reference's DP cell updates.  `traffic` = HBM bytes/launch
from the rocprofv3 PMC passes recorded in profiles/ (null until measured).
cpu_baseline: rank 0, N=1 only — the reference mTR binary (oracle/_ref/mTR_ref, kind "reference") when it
travelled with the repo, else the CPU oracle (kind "port"), on the first reads of the same workload, 1 core.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READS_PER_GPU = 10000
WORKLOAD = "headline2k"
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz = 78.6 T lane-ops/s
OPS_PER_CELL = 7                 # ~7 integer ops per DP cell update


def cpu_baseline(reads, n_sample):
    """Time CPU mTR on the first n_sample reads (1 core).  Returns the cpu_baseline object."""
    from mtr_amd import synth

    sample = [(str(i), reads[i]) for i in range(min(n_sample, len(reads)))]
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "mTR_ref")
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "sample.fa")
        synth.write_fasta(fa, sample)
        if os.path.exists(ref_bin) and os.access(ref_bin, os.X_OK):
            kind, cmd = "reference", [ref_bin, fa]
        else:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
            kind, cmd = "port", [os.path.join(ROOT, "oracle", "mtr_oracle_cli"), fa]
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        one = {"value": len(sample) / dt, "unit": "reads/s", "cores": 1, "kind": kind,
               "sample": f"first {len(sample)} reads of the workload, one process, {dt:.1f} s"}
        # the same binary on every host core of this box's share: one process per core, each on its own reads
        # (SURVEY.md 8d asks for both figures); same number of reads per process as above, so it takes as long
        cores = max(1, min(16, len(os.sched_getaffinity(0))))      # 16 = the CPU share of a one-GPU box on this pool
        if cores > 1:
            per = max(1, min(len(sample), len(reads) // cores))
            files = []
            for c in range(cores):
                f = os.path.join(td, f"part{c}.fa")
                synth.write_fasta(f, [(str(i), reads[i]) for i in range(c * per, (c + 1) * per)])
                files.append(f)
            t0 = time.perf_counter()
            procs = [subprocess.Popen(cmd[:-1] + [f], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for f in files]
            ok = all(p.wait() == 0 for p in procs)
            dt = time.perf_counter() - t0
            if ok:
                one["all_cores"] = {"value": cores * per / dt, "unit": "reads/s", "cores": cores,
                                    "sample": f"{cores} processes x {per} reads, {dt:.1f} s"}
    return one


def measured_traffic():
    """HBM bytes per K2 launch from the committed rocprofv3 PMC summary (profiles/pmc_latest.json), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(p) as fh:
            return float(json.load(fh)["k2_hbm_bytes_per_launch"])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU (default = the headline workload)")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="reads timed on the CPU baseline (0 = skip); 2000 reads ~ 15 s on one core")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--rehearse-exchange", action="store_true", help="development aid, 1 GPU: run the per-step record export of the "
                    "multi-GPU path (compaction kernel + a device copy in place of the RCCL gather) to see what it costs the pipeline")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import mtr_amd
    from mtr_amd import synth
    from mtr_amd.dist import RECORD_BYTES, gather_records

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py: --gpus > 1 needs torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # every rank owns its own block of reads (weak scaling): same distribution, different seed
    reads = [c for _, c in synth.make_reads(WORKLOAD, a.reads, seed=2 + rank)]
    # Two contexts (own stream, own result and scratch buffers) hold the same resident batch: consecutive steps
    # alternate between them, so the kernels of step s+1 are enqueued while the last wavefronts of step s are
    # still finishing (a read is one wavefront's serial chain, the slowest read of a batch takes ~3x the mean).
    # Every step still does all of its work; only the barrier between steps is gone, as in a real multi-batch run.
    # batches in flight: 2 on one GPU (3 and 4 measured: no gain).  With an exchange step (N > 1) one more, so that the next
    # kernel is already enqueued while the host waits for the gather of the previous step, whose RCCL kernels may only get
    # their wave slots when the resident kernel's first waves retire
    NCTX = int(os.environ.get("MTR_BENCH_CONTEXTS", "2" if world == 1 else "3"))
    engs = [mtr_amd.Engine(device=local_rank) for _ in range(NCTX)]
    for e in engs:
        e.upload(reads)                                 # inputs resident in HBM before the timed region
    eng = engs[0]

    rec_buf = [None] * NCTX

    def finish(s):
        e = engs[s % NCTX]
        e.wait()
        if world > 1 or a.rehearse_exchange:
            # exchange step: gather the per-read record tables to rank 0 over RCCL
            total = e.counters()["records"]
            rb = rec_buf[s % NCTX]
            if rb is None or rb.numel() < total * RECORD_BYTES:
                rb = rec_buf[s % NCTX] = torch.empty(max(total, 1) * RECORD_BYTES * 5 // 4, dtype=torch.uint8, device="cuda")
            counts, tot = e.export_records_device(rb.data_ptr(), rb.numel() // RECORD_BYTES)
            if world > 1:
                gather_records(rb[: tot * RECORD_BYTES], torch.from_numpy(counts).cuda(), dst=0)
            else:
                rb[: tot * RECORD_BYTES].clone(); torch.from_numpy(counts).cuda(); torch.cuda.current_stream().synchronize()
        return e.kernel_times_ms()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync_k1, sync_k2 = [], []
    for w in range(a.warmup):                           # warm-up steps run one at a time (un-overlapped kernel times)
        engs[w % NCTX].run_async()
        kt = finish(w)
        sync_k1.append(kt["k1_ranges"]); sync_k2.append(kt["k2_units"])
    k2_ms, k1_ms = [], []
    sync()
    t0 = time.perf_counter()
    depth = NCTX - 1                                    # steps enqueued ahead of the one being finished
    for s in range(min(depth, a.steps)):
        engs[s % NCTX].run_async()
    for s in range(a.steps):
        if s + depth < a.steps:
            engs[(s + depth) % NCTX].run_async()
        kt = finish(s)
        k1_ms.append(kt["k1_ranges"]); k2_ms.append(kt["k2_units"])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        cnt = eng.counters()
        n_local = len(reads)
        total_reads = n_local * world * a.steps
        value = total_reads / dt
        # algorithmic bytes of one launch (this rank's batch): the reference's DP cells = computed + answered from the memo
        sumL4 = sum((len(r) + 3) // 4 for r in reads)
        cells = cnt["dp_cells"] + cnt["revise_dp_cells"] + cnt["memo_cells"]
        b_alg = sumL4 + 576 * cnt["records"] + (cells + 1) // 2
        k2_avg_s = float(np.mean(k2_ms)) / 1e3
        achieved_gbs = b_alg / k2_avg_s / 1e9
        valu_ops = cells * OPS_PER_CELL / k2_avg_s
        out = {
            "metric": "reads/sec, 2 kb Nanopore synthetic",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"{WORKLOAD}: {n_local} synthetic Nanopore reads per GPU, unit 100 x 10 copies, 500-base flanks, "
                                   f"mean L {np.mean([len(r) for r in reads]):.0f}, error profile sub 1.6/ins 9.0/del 3.8 %",
                       "reads_per_gpu": n_local, "parallelism": f"reads sharded over {world} GPU(s), gather to rank 0"},
            "ms_per_read": dt / a.steps * 1e3 / n_local,
            "kernels_ms": {"mtr_k_reads": float(np.mean(k2_ms)),
                           "note": "HIP-event durations over the timed region; consecutive steps overlap on the GPU, so a launch shares the chip with its neighbour"},
            "kernels_ms_alone": {"mtr_k_reads": float(np.mean(sync_k2)) if sync_k2 else None},
            "work_per_launch": {k: cnt[k] for k in ("dp_calls", "dp_cells", "dp_rows", "revise_dp_calls", "revise_dp_cells", "memo_hits", "memo_cells",
                                                    "kmer_tables", "tables_skipped", "kmer_lookups", "ranges_executed", "records", "traceback_steps")},
            "roofline": {"bound": "hbm", "kernel": "mtr_k_reads", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": measured_traffic(),
                         "algorithmic_bytes_per_launch": b_alg,
                         "valu": {"achieved_lane_ops_per_s": valu_ops, "peak_lane_ops_per_s": VALU_PEAK_LANE_OPS,
                                  "frac": valu_ops / VALU_PEAK_LANE_OPS, "cells_per_launch": cells, "ops_per_cell": OPS_PER_CELL},
                         "note": "not HBM-bound: row-serial integer max-plus recurrence (instruction issue + cross-lane scan latency); cells = the reference's DP cells, of which memo_cells were answered without a DP"},
        }
        if world == 1 and not a.no_latency:
            lat = []
            e2 = mtr_amd.Engine(device=local_rank)
            for i in range(33):
                t1 = time.perf_counter()
                e2.process([reads[i]])
                lat.append((time.perf_counter() - t1) * 1e3)
            e2.close()
            out["latency_ms_per_read_p50"] = float(np.median(lat[1:]))
        if world == 1 and a.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(reads, a.cpu_sample)
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
            if "all_cores" in out["cpu_baseline"]:
                out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["all_cores"]["value"]
        print(json.dumps(out))
    for e in engs:
        e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
