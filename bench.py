#!/usr/bin/env python3
"""bench.py — reads/s of the per-read hot path (handle_one_read) on MI355X, next to CPU mTR.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)
  python bench.py --gpus N --strong c4 ...               (strong scaling: ONE set of 100 000 config-4 reads split over the ranks)

A "step" is one pass of the hot path over one batch of synthetic reads that is already resident in HBM (2 bit/base):
the per-read kernels (candidate ranges, unit search, wrap-around DPs, revision; mtr_run_resident) AND the hand-over at
the boundary's lower edge — the record tables compacted on the device to the wire form and copied into pinned host
memory (mtr_fetch_results_packed), i.e. the arguments of insert_an_alignment_into_set where the host's chaining takes
them.  Workload = the configuration BASELINE.json's metric is quoted on: 10 000 synthetic Nanopore-error reads of ~2 kb
(unit 100 x 10 copies, 500-base flanks; mtr_amd.synth "headline2k") per GPU.  Steps take turns on three contexts (three
batches in flight, as the host pipeline keeps them on a long job: mtr_amd/host/pipeline.c), so the fetch of step s
overlaps the kernels of the steps behind it.

  value          reads/s at the boundary (above);            value_kernel  the same steps without the fetch (round 1's figure)
  value_cli      reads/s of the command line mtr_amd/host/mTR on a FASTA of the same reads, wall clock incl. process
                 start, HIP initialisation, parsing, chaining and printing (N = 1 only; NOT resident inputs)
  value_with_upload  the steps with a FRESH batch each, handed over as host buffers: 2-bit packing + copy to the device inside
                 the step (N = 1; PCIe-inclusive, never `value`)
  value_launcher reads/s of the multi-GPU PRODUCT path: mTR -g N (the C host, one process, RCCL gather) on a FASTA of 100 000
                 reads, wall clock incl. process start, HIP + RCCL initialisation, parsing, chaining and printing
With N GPUs (weak scaling) every rank holds its own 10 000 reads and the step ends with the ONE exchange of the path: the
wire-form tables go device-to-device to rank 0 over RCCL (mtr_export_packed_device + gather).  --strong c4: BASELINE
config 4 — one set of 100 000 mixed-unit reads, contiguous blocks balanced by sum of lengths; rank 0 checks the sha256 of
the gathered stream against the known answer of the CPU oracle (tests/golden/c4_100k_wire.json) and reports ranks_seen.

Round 6: every BASELINE config is in the default line, each with its own evidence - `secondary.c2` / `secondary.c3` (configs 2 and 3 as resident-batch steps in child
processes, record streams against the oracle's known answers), `baseline_configs_cli` (configs 2, 4 and 5 as jobs of `mTR -c -g N`, fresh child processes, stdout
against committed known answers), `host_ceiling` (the host pipeline alone behind eight null GPUs), `cpu_baseline.ms_per_read_p50` (the reference, one read per
process); `roofline.issue` by the lone launch and by the step, `roofline.slot_time` = wavefront-slot time per kernel, `roofline.traffic` - all three from the
committed PMC passes of THE SAME workload (profiles/pmc_latest.json for the headline, profiles/pmc_c3.json for config 3), null otherwise.  At N > 1 rank 0 measures
`value_launcher` (default and with every round forced onto RCCL), `baseline_configs_cli` and `host_ceiling` as child processes BEFORE any rank touches a GPU.

roofline: a launch is the staged chain (k3_staged.hip.inc: ranges -> [unit search -> alignments -> selection -> revisions] for the ranges of wide
windows, then for the ranges their records leave -> replay, one stream, no host round trip); its duration and the durations of its phases are measured with HIP events on the launch stream
inside the timed region (mtr_get_kernel_times), the dominant phases are the wrap-around DP kernels mtr_k_dp2_quads (alignments) and
mtr_k_revise_quads (revisions).  The path is bound by VALU instruction issue (row-serial integer max-plus recurrence),
so `bound` is "valu-issue": `achieved` = the REFERENCE's DP cell updates of one launch x 7 integer operations / the
launch's duration, `peak` = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz lane-operations per second; `frac_by_step` prices the same work
against the driver-visible step time.  `hbm` keeps the HBM view of SURVEY.md 8d: algorithmic bytes B_alg = ceil(L/4) + 576 R + sum over
the REFERENCE's DPs of ceil(cells/2) with EVERY DP counted as spilled (the DPs the kernels answer from their memo are the reference's
work and are counted), `traffic` (HBM bytes per launch, FETCH_SIZE x 2 + WRITE_SIZE as the gfx950 guide prescribes) and `issue`
(SQ_INSTS_VALU x 2 cycles / SIMD-cycles of the launch) from the rocprofv3 PMC passes committed under profiles/ (profiles/pmc_latest.json),
reported only while the kernel sources are the ones that were profiled (sha of mtr_amd/csrc), else null.
cpu_baseline: rank 0, N = 1 only — the reference mTR binary (oracle/_ref/mTR_ref, kind "reference") when it travelled
with the repo, else the CPU oracle (kind "port"), on the first reads of the same workload, 1 core (and all cores).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess

# The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4).  An RCCL communicator brings
# streams of its own; with the default, the two context streams of a rank then share ONE hardware queue and their launches no
# longer overlap (measured with one rank under RCCL: 80 ms a step instead of 52; with 8 queues 52 again).  Must be set before the
# runtime initialises, i.e. before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
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


def kernel_sources_sha():
    # the sources of the kernels and the flags they are built with
    h = hashlib.sha256()
    d = os.path.join(ROOT, "mtr_amd", "csrc")
    for f in ("device_util.hip.inc", "dp_wrap.hip.inc", "dp_quad.hip.inc", "k1_ranges.hip.inc", "k2_units.hip.inc", "k3_staged.hip.inc", "mtr_common.h",
              "min_missing_table.h", "mtr_abi.hip"):
        h.update(f.encode())
        h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "mtr_amd", "build.py"), "rb").read())
    return h.hexdigest()[:16]


def profiled_counters(config=None):
    """The committed rocprofv3 PMC summary of THIS workload - profiles/pmc_latest.json for the headline, profiles/pmc_<config>.json for a secondary config
    (profiles/collect_all.sh <tag> [--config c3]) - or None when there is none or the kernel sources changed since it was taken.  Round 5 read one file for every
    config: the config-3 line carried the headline's traffic and issue figures."""
    name = "pmc_latest.json" if config is None else f"pmc_{config}.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as fh:
            p = json.load(fh)
    except Exception:
        return None, f"no profile of this workload (profiles/{name})"
    if p.get("kernel_src_sha") != kernel_sources_sha():
        return None, f"stale: profiles/{name} was taken at kernel sources {p.get('kernel_src_sha')}"
    return p, p.get("tag", "")


def slot_time(prof):
    """Wavefront-slot time of one launch of the chain from the profile's SQ_WAVE_CYCLES pass (x 4: the counter ticks once per four cycles): the sum over the chain's
    kernels, what it is in ms on the chip's 4 096 slots of sixteen per CU, and every kernel's share.  The pipelined step is made of exactly this (round 5:
    359.7 G = 36.6 ms against a 36.4 ms step), which is why a kernel that holds wavefronts without issuing anything costs the step its residence time."""
    if not prof or "chain_per_launch" not in prof:
        return None
    ks = {k: v.get("wave_cycles", 0.0) for k, v in prof["chain_per_launch"].items()}
    tot = sum(ks.values())
    if tot <= 0:
        return None
    service = ("mtr_k_gather", "mtr_k_select", "mtr_k_rev_share", "mtr_k_finish", "mtr_k_polish", "mtr_k_replay", "mtr_k_pass_mark", "mtr_k_item_table")
    return {"sum_wave_cycles": tot, "ms_on_4096_slots_at_2.4GHz": tot / (4096 * 2.4e9) * 1e3,
            "service_kernels_wave_cycles": sum(v for k, v in ks.items() if k in service),
            "service_kernels": list(service),
            "kernels_G": {k: round(v / 1e9, 3) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]) if v >= 1e7},
            "shares": {k: round(v / tot, 4) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]) if v / tot >= 0.002},
            "source": "SQ_WAVE_CYCLES x 4 per kernel of the chain, rocprofv3 --pmc pass of the committed profile (pipelined launches)"}


def cpu_baseline(reads, n_sample, p50_budget_s=20.0):
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
        # ms/read p50 (BASELINE.json's metric is reads/sec + ms/read p50): the same binary, ONE READ PER PROCESS - the isolated semantics the parity contract is
        # stated in, and the only way the reference's own per-read timers (handle_one_read.c:217-258, main.c:100-111) can be read per read -
        # on the first reads of the sample, wall clock of each process
        lat = []
        n_lat = min(len(sample), 200 if p50_budget_s is None else 200)
        t_budget = time.perf_counter()
        for i in range(n_lat):
            f1 = os.path.join(td, "one.fa")
            synth.write_fasta(f1, [sample[i]])
            t1 = time.perf_counter()
            subprocess.run(cmd[:-1] + [f1], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            lat.append((time.perf_counter() - t1) * 1e3)
            if p50_budget_s is not None and time.perf_counter() - t_budget > p50_budget_s:
                break
        if lat:
            one["ms_per_read_p50"] = float(np.median(lat))
            one["ms_per_read_p50_sample"] = f"{len(lat)} reads, one process per read (process start included: ~1 ms)"
        # the same binary on every host core of this box's share: one process per core, each on its own reads
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


def empty_hip_program():
    """What a process pays on THIS box before its first kernel has run - a program that initialises the HIP runtime, creates a stream and runs an
    empty kernel (tests/dev/hip_init_probe.hip, built by __graft_entry__.build()): the floor under the command line's start-up, three runs."""
    exe = os.path.join(ROOT, "tests", "dev", "hip_init_probe")
    if not os.path.exists(exe):
        return None
    out = []
    for _ in range(3):
        try:
            p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
            ms = [float(ln.split(" ms")[0].split()[-1]) for ln in p.stdout.splitlines() if ln.startswith("first kernel")]
            if ms:
                out.append(ms[0] / 1e3)
        except Exception:
            pass
    return {"seconds_until_first_kernel_done": out, "min": min(out), "max": max(out),
            "note": "hipInit + stream + first kernel of an EMPTY program on this box in this run; the runtime's own start-up, which varies from process to process"} if out else None


def cli_rate(reads, n, flags=()):
    """wall clock of the command line on a FASTA of the first n reads (page cache warm), best of 3"""
    from mtr_amd import synth

    exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "mTR"], check=True)
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "reads.fa")
        synth.write_fasta(fa, [(str(i), reads[i % len(reads)]) for i in range(n)])
        best, stamps = None, None
        for _ in range(3):
            t0 = time.perf_counter()
            p = subprocess.run([exe, *flags, fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MTR_HOST_TIMING="1"))
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode()[-200:]}
            if best is None or dt < best:
                best = dt
                # where the wall clock went (seconds since main): the HIP runtime's start-up, the first batch's latency, the batches at the kernels' pace
                stamps = {ln.split("] ", 1)[1]: float(ln.split("+")[1].split(" s")[0]) for ln in p.stderr.decode().splitlines() if ln.startswith("[host +")}
    return {"reads": n, "seconds": best, "reads_per_s": n / best, "stamps_s": stamps}


def launcher_rate(reads, n_total, n_gpus, force_rccl=False):
    """wall clock of the multi-GPU PRODUCT path - mtr_amd/host/mTR -g N (one process, a run per GPU, the record tables gathered to the first GPU over
    RCCL, chained and printed there) - on a FASTA of n_total reads, process start, HIP / RCCL initialisation, parsing and printing included; best of 2"""
    from mtr_amd import synth

    exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "mTR"], check=True)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MTR_LIB")}
    env.update(MTR_HOST_TIMING="1", GPU_MAX_HW_QUEUES="8")
    if force_rccl:      # every round over RCCL: the job waits for librccl + ncclCommInitAll before its first batch (on one GPU the tables go to the GPU itself)
        env.update(MTR_GATHER="rccl", MTR_GATHER_SELF="1")
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "reads.fa")
        synth.write_fasta(fa, [(str(i), reads[i % len(reads)]) for i in range(n_total)])
        best, gather, err = None, None, None
        for _ in range(2):
            t0 = time.perf_counter()
            try:
                p = subprocess.run([exe, "-c", "-g", str(n_gpus), fa], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=600)
            except subprocess.TimeoutExpired:
                return {"error": "timeout after 600 s"}
            dt = time.perf_counter() - t0
            if p.returncode != 0:
                return {"error": p.stderr.decode(errors="replace")[-300:]}
            if best is None or dt < best:
                best = dt
                gather = [ln for ln in p.stderr.decode(errors="replace").splitlines() if "\tgather " in ln]
    out = {"command": f"mTR -c -g {n_gpus} <fasta of {n_total} reads> > /dev/null" + (" with MTR_GATHER=rccl" if force_rccl else ""), "reads": n_total, "gpus": n_gpus,
           "seconds": best, "reads_per_s": n_total / best, "gather": gather[0] if gather else None}
    if gather and "gather rccl, " in gather[0]:
        try:
            out["exchanges_over_rccl"] = int(gather[0].split("gather rccl, ")[1].split(" exchange")[0])
            out["exchanges_straight_to_the_host"] = int(gather[0].split(" over RCCL + ")[1].split(" straight")[0])
        except Exception:
            pass
    return out


C5_FILES = ["3_5", "3_10", "3_20", "3_50", "5_10", "5_20", "5_50", "10_20", "10_50", "20_50", "2_5_10_20_set", "2_5_10_20_50_100_200_set",
            "worm_chrI", "worm_chrII_1", "worm_chrII_2"]                     # test_multiple_TRs/test.sh:8-31, in its order


def _child_env(extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MTR_LIB")}
    env.update(GPU_MAX_HW_QUEUES="8")
    if extra:
        env.update(extra)
    return env


def _run_job(cmd, env, repeats=2, timeout=600):
    """a command line as a fresh child process: best wall clock of `repeats`, its stdout's sha256 and line count, its -c block"""
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=timeout)
        except subprocess.TimeoutExpired:
            return {"error": f"timeout after {timeout} s"}
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            return {"error": p.stderr.decode(errors="replace")[-300:], "returncode": p.returncode}
        if best is None or dt < best["seconds"]:
            err = p.stderr.decode(errors="replace")
            block = {}
            for ln in err.splitlines():
                parts = ln.strip().split("\t")
                if len(parts) == 2 and parts[0].replace(".", "", 1).isdigit():
                    block[parts[1]] = float(parts[0])
            gather = [ln for ln in err.splitlines() if "\tgather " in ln]
            best = {"seconds": dt, "stdout_sha256": hashlib.sha256(p.stdout).hexdigest(), "stdout_lines": p.stdout.count(b"\n"),
                    "c_block_s": block, "gather": gather[0] if gather else None}
    if best.get("gather") and "exchange(s) over RCCL" in best["gather"]:
        try:
            best["exchanges_over_rccl"] = int(best["gather"].split("gather rccl, ")[1].split(" exchange")[0])
        except Exception:
            pass
    return best


def baseline_configs_through_the_command_line(n_gpus, with_per_file=True):
    """BASELINE configs 2, 4 and 5 as the jobs users run: `mTR -c -g N` (the C host, one process, a run per GPU, the record tables gathered to the first GPU) as a
    fresh child process each - wall clock incl. process start, HIP (and RCCL) initialisation, parsing, chaining and printing - with the stdout checked against a
    committed known answer of the CPU oracle's command line / the reference's goldens.  Round 5's driver record had numbers for the headline and config 3 only."""
    from mtr_amd import synth

    exe = os.path.join(ROOT, "mtr_amd", "host", "mTR")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe), "mTR"], check=True)
    env = _child_env({"MTR_HOST_TIMING": "1"})
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for cfg, n in (("c2", 1000), ("c4", 100000)):
            try:
                with open(os.path.join(ROOT, "tests", "golden", f"{cfg}_{n}_stdout.json")) as fh:
                    known = json.load(fh)
            except Exception:
                known = None
            fa = os.path.join(td, f"{cfg}.fa")
            synth.write_fasta(fa, synth.make_reads(cfg, n, synth.CONFIGS[cfg][4]))
            _run_job([exe, "-g", str(n_gpus), fa], env, repeats=1)                    # (page cache and the driver's memory pools warm)
            j = _run_job([exe, "-c", "-g", str(n_gpus), fa], env)
            if "error" not in j:
                j.update(command=f"mTR -c -g {n_gpus} <fasta of the {n} reads of synth config {cfg}>", reads=n, gpus=n_gpus, reads_per_s=n / j["seconds"],
                         matches_oracle=(known is not None and known["sha256"] == j["stdout_sha256"] and known["stdout_lines"] == j["stdout_lines"]) if known else None,
                         known_answer=f"tests/golden/{cfg}_{n}_stdout.json (oracle/mtr_oracle_cli, pinned to the reference)" if known else None)
            out[cfg] = j
            os.unlink(fa)
        # config 5: the 15 files of test_multiple_TRs (one read each, 2.6-140 kb), -p: longest first over the GPUs, one round, output in command-line order
        files = [os.path.join(ROOT, "tests", "golden", "inputs", f"{n}.fasta") for n in C5_FILES]
        if all(os.path.exists(f) for f in files):
            want = hashlib.sha256()
            for n in C5_FILES:
                want.update(open(os.path.join(ROOT, "tests", "golden", f"{n}.p.stdout"), "rb").read())
            _run_job([exe, "-p", "-g", str(n_gpus)] + files, env, repeats=1)
            j = _run_job([exe, "-c", "-p", "-g", str(n_gpus)] + files, env)
            if "error" not in j:
                j.update(command=f"mTR -c -p -g {n_gpus} <the 15 files of test_multiple_TRs/test.sh:8-31>", files=len(files), gpus=n_gpus,
                         sum_bases=int(sum(os.path.getsize(f) for f in files)), matches_reference_goldens=j["stdout_sha256"] == want.hexdigest(),
                         known_answer="tests/golden/<file>.p.stdout (the reference compiled here, -p), concatenated in command-line order")
                if with_per_file:
                    per = {}
                    for n, f in zip(C5_FILES, files):
                        k = _run_job([exe, "-c", "-p", f], env, repeats=1)
                        b = k.get("c_block_s", {})
                        per[n] = {"wall_s": k.get("seconds"), "device_ms": round(1e3 * max(0.0, b.get("ranges", 0.0) + b.get("Computing periods", 0.0) - b.get("chaining", 0.0)), 3) if b else None}
                    j["per_file_alone_on_one_gpu"] = per
            out["c5"] = j
    return out


def host_ceiling_leg(n_reads=1000000):
    """The host pipeline ALONE behind `mTR -g 8`: tests/null_engine.c answers every batch at once (two records per read), so what is timed is cutting, parsing + 2-bit
    packing, the runs, the exchange, wire unpack + chaining + formatting and the writer - what one process must sustain for eight GPUs (8 x the headline rate = 2.3 M
    reads/s).  No GPU is touched.  Stage costs: the same job with no records (parsing + packing + pipeline only) against the full one."""
    from mtr_amd import synth
    from tests import host_util

    try:
        reads = [c for _, c in synth.make_reads(WORKLOAD, 10000, seed=2)]
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > (5 << 30) else tempfile.gettempdir()
        one = os.path.join(d, f"mtr_ceiling_{os.getpid()}_10k.fa")
        big = os.path.join(d, f"mtr_ceiling_{os.getpid()}.fa")
        synth.write_fasta(one, [(str(i), reads[i]) for i in range(len(reads))])
        blob = open(one, "rb").read()
        os.unlink(one)
        with open(big, "wb") as fh:
            for _ in range(max(1, n_reads // len(reads))):
                fh.write(blob)
        n = max(1, n_reads // len(reads)) * len(reads)
        try:
            full = host_util.host_ceiling(big, n, n_gpus=8, records=2, repeats=3)
            bare = host_util.host_ceiling(big, n, n_gpus=8, records=0, repeats=2)
            g1 = host_util.host_ceiling(big, n, n_gpus=1, records=2, repeats=2)
        finally:
            os.unlink(big)
        cpu_full, cpu_bare = full["user_s"] + full["sys_s"], bare["user_s"] + bare["sys_s"]
        return {"command": "mTR -g 8 <fasta> > /dev/null with MTR_LIB=tests/libmtr_null.so (every batch answered at once with 2 records per read)",
                "reads": n, "fasta_bytes": len(blob) * (n // len(reads)), "seconds": full["seconds"], "reads_per_s": full["reads_per_s"],
                "host_cores": full["host_cores"], "cores_used": full["cores_used"], "cpu_us_per_read": cpu_full / n * 1e6,
                "reads_per_s_per_core_used": full["reads_per_s_per_core_used"],
                "stages_cpu_us_per_read": {"cut + parse + 2-bit pack + batches through the runs": cpu_bare / n * 1e6,
                                           "exchange + wire unpack + chain + format + write": (cpu_full - cpu_bare) / n * 1e6,
                                           "of which system time (page faults, mmap)": full["sys_s"] / n * 1e6},
                "without_records": {"seconds": bare["seconds"], "reads_per_s": bare["reads_per_s"]},
                "one_gpu": {"seconds": g1["seconds"], "reads_per_s": g1["reads_per_s"], "cores_used": g1["cores_used"]},
                "needed_for_8_gpus_at_the_headline_rate": "8 x value reads/s",
                "note": "CPU only (no GPU touched).  Round 5: 238 k reads/s on 8 vCPU through the replay engine (one manager thread formatted a result, waited, wrote)"}
    except Exception as ex:
        return {"error": repr(ex)[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)            # (1.5 s of timed steps: the timed region starts on an idle GPU, and filling three batches in flight is
    ap.add_argument("--warmup", type=int, default=4)           #  one launch's worth of time that a short run spreads over few steps)
    ap.add_argument("--reads", type=int, default=READS_PER_GPU, help="reads per GPU (default = the headline workload)")
    ap.add_argument("--strong", default=None, choices=["c4"], help="strong scaling: one fixed read set split over the ranks")
    ap.add_argument("--strong-reads", type=int, default=100000)
    ap.add_argument("--cpu-sample", type=int, default=2000, help="reads timed on the CPU baseline (0 = skip); 2000 reads ~ 15 s on one core")
    ap.add_argument("--config", default=None, choices=["c2", "c3"], help="a secondary line for another BASELINE config: c2 = 1 000 reads of unit 100 x 10 copies (L ~ 1.25 kb); c3 = 100 reads of unit 200 x 200 copies (L ~ 42 kb), -a on in the command-line leg")
    ap.add_argument("--no-secondary", action="store_true", help="the default line without its secondary objects (BASELINE configs 2 and 3 measured in child processes, configs 2 / 4 / 5 through the command line, the host's ceiling)")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--no-cli", action="store_true")
    ap.add_argument("--no-upload-leg", action="store_true", help="skip value_with_upload (profiling passes: only launches of the resident headline batch)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import mtr_amd
    from mtr_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        print("bench.py: --gpus > 1 needs torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    # value_launcher at N > 1 (what users run on N GPUs: the C command line with -g N on 100 000 reads, wall clock incl. start-up): measured FIRST, on a node
    # whose GPUs nobody has touched yet - rank 0 runs the command as a child process before any rank initialises its GPU, the others wait for a marker file
    # (no process group exists yet).  Run after the timed steps it shared every GPU with a rank's idle context and waited for the driver to reclaim the
    # ranks' freed memory (2.8 s instead of 0.9 s in the two-rank rehearsal on one card).
    launcher_pre = None
    pre = {}
    t_start = time.time()
    if world > 1 and not a.no_cli and not a.strong and a.config is None:
        # (the marker: named after the launcher's process and port, and only believed when it was written AFTER this process started - a stale one from a run that
        #  died would let the other ranks initialise their GPUs while rank 0 is still timing)
        marker = os.path.join(tempfile.gettempdir(), f"mtr_bench_launcher_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}.done")
        if rank == 0:
            try:
                os.remove(marker)
            except OSError:
                pass
            try:
                hreads = [c for _, c in synth.make_reads(WORKLOAD, a.reads, seed=2)]
                launcher_pre = launcher_rate(hreads, 100000, world)
                # the same job with every round forced onto RCCL (the default lets RCCL come up in the background: a 100 000-read job on 8 GPUs is over before it is up)
                pre["launcher_rccl_forced"] = launcher_rate(hreads, 100000, world, force_rccl=True)
                if not a.no_secondary:
                    pre["baseline_configs_cli"] = baseline_configs_through_the_command_line(world, with_per_file=False)
                    pre["host_ceiling"] = host_ceiling_leg()
            except Exception as ex:          # the bench line must not die of its optional legs
                launcher_pre = launcher_pre or {"error": repr(ex)[:300]}
                pre.setdefault("error", repr(ex)[:300])
            with open(marker, "w") as fh:
                fh.write("done")
        else:
            t_wait = time.perf_counter()
            while time.perf_counter() - t_wait < 2400:
                try:
                    if os.path.getmtime(marker) >= t_start - 1.0:
                        break
                except OSError:
                    pass
                time.sleep(0.2)
    # MTR_BENCH_BACKEND=gloo: a rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share GPUs, the exchange
    # goes through host memory); the driver's runs use RCCL ("nccl"), one GPU per rank
    backend = os.environ.get("MTR_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    xdev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    # MTR_BENCH_FORCE_DIST=1 (under torch.distributed.run with ONE rank): the N > 1 code path - process group, size exchange,
    # gather to rank 0 - on a single GPU, to exercise the RCCL calls where only one GPU is available
    dist_on = world > 1 or os.environ.get("MTR_BENCH_FORCE_DIST") == "1"
    report = sys.stdout
    if dist_on:
        # stdout carries ONE JSON line: RCCL prints its version banner on descriptor 1 at the first collective, so the line keeps a descriptor of its own and
        # descriptor 1 points at stderr from here on
        sys.stdout.flush()
        report = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if a.strong:
        # one data set for the whole job, a contiguous block of it per rank (the reads of this config are all ~2 kb: equal counts are
        # equal sums of lengths to within a per cent).  A rank generates only ITS block: the seeded stream is entered at the nearest
        # committed generator checkpoint (tests/golden/c4_rng_checkpoints.npz, every 5 000 reads) instead of at read 0.
        n_job = a.strong_reads
        lo, hi = rank * n_job // world, (rank + 1) * n_job // world
        ck = None
        try:
            z = np.load(os.path.join(ROOT, "tests", "golden", f"{a.strong}_rng_checkpoints.npz"))
            ck = (z["idx"], z["keys"], z["pos"])
        except Exception:
            pass
        reads = [c for _, c in synth.make_reads_range(a.strong, lo, hi, 4, ck)]
    elif a.config is not None:
        reads = [c for _, c in synth.make_reads(a.config, {"c2": 1000, "c3": 100}[a.config], seed=synth.CONFIGS[a.config][4] + rank)]
        n_job = len(reads) * world
    else:
        # every rank owns its own block of reads (weak scaling): same distribution, different seed
        reads = [c for _, c in synth.make_reads(WORKLOAD, a.reads, seed=2 + rank)]
        n_job = len(reads) * world
    # The contexts (own stream, own result and scratch buffers; three: round 5 measured 36.64 / 36.13 / 36.02 ms a step with two / three / four) hold the same
    # resident batch: consecutive steps take turns on them, so the kernels of step s+1 are enqueued while step s is fetched (and while its last wavefronts finish:
    # a read is one wavefront's serial chain).  Every step does all of its work.  With an exchange step (N > 1) one more
    # context, so that the next kernel is already enqueued while the host waits for the gather of the previous step.
    # --strong: ONE data set is the whole job, so a step is the whole job - kernels, then the gather, then the copy to the host, one after the other on
    # one context (repetitions of a job do not overlap each other; round 4 pipelined them and measured the next repetition's persistent kernels holding
    # the wavefront slots the gather's kernels were waiting for: 687 ms a step for a 335 ms launch).  Config 3: six batches in flight (round 5: three), as the host
    # pipeline runs long reads (launches bound by their longest work items leave most of the chip idle: 78.5 ms a step with two, 53.8 with three).
    # batches in flight: three (the host pipeline's choice for long jobs of short reads), six for long reads (mtr_amd/host/pipeline.c); a strong step is the whole job: one
    # (config 3: six, as the host pipeline runs long reads since round 6 - 46.6 / 38.9 / 36.4 / 32.8 / 31.6 / 34.6 ms a step with 3 / 4 / 5 / 6 / 8 / 10 in flight)
    NCTX = int(os.environ.get("MTR_BENCH_CONTEXTS", "1" if a.strong else "6" if a.config == "c3" else "3"))
    engs = [mtr_amd.Engine(device=local_rank) for _ in range(NCTX)]
    for e in engs:
        e.upload(reads)                                 # inputs resident in HBM before the timed region
    eng = engs[0]
    wire_buf = [None] * NCTX
    gathered = {}

    out_bufs = [None] * NCTX                            # rank 0: where the ranks' tables land, per context slot
    works = [None] * NCTX                               # the gather still in flight on a context slot's buffers
    agreed = {"cap": None}                              # bytes every rank sends per step, agreed once (the first exchange)

    host_bufs = [None] * NCTX                           # rank 0: the gathered tables in pinned host memory (where N = 1 ends too)

    def to_host(i):
        """rank 0: the tables a finished gather left on its GPU go to pinned host memory (one copy per rank's buffer, enqueued)"""
        if rank != 0 or out_bufs[i] is None or backend != "nccl":
            return                                      # (gloo: the gather's output already is host memory)
        cap = out_bufs[i][0].numel()
        if host_bufs[i] is None or host_bufs[i].numel() != cap * world:
            host_bufs[i] = torch.empty(cap * world, dtype=torch.uint8, pin_memory=True)
        for r in range(world):
            host_bufs[i][r * cap:(r + 1) * cap].copy_(out_bufs[i][r], non_blocking=True)

    def exchange(s, keep=False):
        """the path's one exchange: wire-form tables device-to-device to rank 0 (RCCL gather).
        The first exchange (a warm-up step) all-gathers the table sizes and fixes the per-rank capacity; after that a step
        ENQUEUES one gather of fixed-size buffers and goes on: a collective's kernel only gets onto the chip when wavefront slots
        free up - the per-read kernel holds all of them until a launch drains - so a host that waits for it (or for a size
        exchange in front of it) every step stalls the pipeline (measured with one rank: 63 instead of 52 ms a step)."""
        i = s % NCTX
        e = engs[i]
        tdbg = [time.perf_counter()]
        if works[i] is not None:                        # the previous gather out of this slot's buffer (NCTX steps ago): it must
            works[i].wait(); works[i] = None            # have READ the buffer before the engine's own stream writes it again
            to_host(i)
            torch.cuda.current_stream().synchronize()
        tdbg.append(time.perf_counter())
        wb = wire_buf[i]
        need = agreed["cap"] or max(1 << 20, 700 * max(1, e.counters()["records"]))
        if wb is None or wb.numel() < need:
            wb = wire_buf[i] = torch.zeros(need * 5 // 4 if agreed["cap"] is None else need, dtype=torch.uint8, device="cuda")
        tdbg.append(time.perf_counter())
        counts, total, nbytes = e.export_packed_device(wb.data_ptr(), wb.numel())
        tdbg.append(time.perf_counter())
        first = agreed["cap"] is None
        all_sizes = None
        if first or keep or backend != "nccl":
            sizes = torch.tensor([nbytes, total, len(counts)], dtype=torch.int64, device=xdev)
            all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
            dist.all_gather(all_sizes, sizes)
        if first:
            agreed["cap"] = max(1 << 20, max(int(x[0]) for x in all_sizes) * 3 // 2)
            if wb.numel() < agreed["cap"]:
                nb = torch.zeros(agreed["cap"], dtype=torch.uint8, device="cuda"); nb[: wb.numel()] = wb; wb = wire_buf[i] = nb
        cap = agreed["cap"]
        if nbytes > cap:
            raise RuntimeError(f"record table of {nbytes} bytes exceeds the capacity agreed at the first exchange ({cap})")
        pad = wb[:cap].to(xdev)
        if rank == 0 and (out_bufs[i] is None or out_bufs[i][0].numel() != cap):
            out_bufs[i] = [torch.empty(cap, dtype=torch.uint8, device=xdev) for _ in range(world)]
        out = out_bufs[i] if rank == 0 else None
        if os.environ.get("MTR_BENCH_XMODE") == "export" and not keep:
            pass
        elif backend == "nccl" and not keep and not a.strong:
            works[i] = dist.gather(pad, out, dst=0, async_op=True)
        else:
            dist.gather(pad, out, dst=0)
            to_host(i)
            torch.cuda.current_stream().synchronize()
        tdbg.append(time.perf_counter())
        if os.environ.get("MTR_BENCH_DEBUG") and rank == 0:
            print("exchange step", s, "ms: wait prev gather %.2f, buffers %.2f, export %.2f, sizes+gather enqueue %.2f" %
                  tuple((tdbg[j + 1] - tdbg[j]) * 1e3 for j in range(4)), file=sys.stderr)
        if rank == 0 and keep:
            gathered["blobs"] = [out[r][: int(all_sizes[r][0])].cpu().numpy().tobytes() for r in range(world)]
            gathered["records"] = [int(x[1]) for x in all_sizes]
            gathered["reads"] = [int(x[2]) for x in all_sizes]

    def drain():
        for i in range(NCTX):
            if works[i] is not None:
                works[i].wait(); works[i] = None
                to_host(i)
        torch.cuda.current_stream().synchronize()

    def finish(s, fetch=True, keep=False):
        e = engs[s % NCTX]
        t_w = time.perf_counter()
        e.wait()
        if os.environ.get("MTR_BENCH_DEBUG") and rank == 0:
            print("step", s, "waited %.2f ms for the kernels" % ((time.perf_counter() - t_w) * 1e3), file=sys.stderr)
        xmode = os.environ.get("MTR_BENCH_XMODE", "full")          # development: which part of the exchange step runs
        if dist_on and xmode != "fetch":
            exchange(s, keep)
        elif fetch:
            e.fetch_packed_nocopy()                     # wire form -> pinned host memory of the context
        return e.kernel_times_ms()

    def sync():
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, fetch=True):
        k2 = []
        sync()
        t0 = time.perf_counter()
        depth = max(NCTX - 1, 0)                        # steps enqueued ahead of the one being finished
        for s in range(min(depth, steps)):
            engs[s % NCTX].run_async()
        for s in range(steps):
            if s + depth < steps:
                engs[(s + depth) % NCTX].run_async()
            k2.append(finish(s, fetch, keep=(s == steps - 1)))
        if dist_on:
            drain()                                     # every gather of the timed steps has completed inside the timed region
        sync()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device=xdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, k2

    # the REFERENCE's work on this batch (its DP calls and cells, what it answers from a repeated call included): one launch of the
    # per-read kernel, which runs the reference's own sequential range loop (the chain also searches ranges that loop removes)
    staged_before = os.environ.get("MTR_STAGED")
    os.environ["MTR_STAGED"] = "0"
    try:
        engs[0].run()
        ref_cnt = engs[0].counters()
    finally:
        if staged_before is None:
            del os.environ["MTR_STAGED"]
        else:
            os.environ["MTR_STAGED"] = staged_before     # the caller's choice for the timed steps stays
    for e in engs[1:]:                                  # every context has run once before anything is timed: a context's first launch allocates its
        e.run()                                         # scratch and chain buffers (GBs; hipMalloc inside the timed region is not a step's work)
    sync_k2 = []
    lone_mode = None
    for w in range(a.warmup):                           # warm-up steps run one at a time (un-overlapped kernel times)
        engs[w % NCTX].run_async()
        sync_k2.append(finish(w))
        lone_mode = engs[w % NCTX].last_mode()
    if dist_on:
        drain()
    dt, k2_ms = timed(a.steps, fetch=True)
    dt_kernel = None
    dt_upload = None
    if world == 1:
        dt_kernel, _ = timed(a.steps, fetch=False)      # round 1's figure: the kernels alone, results left on the device
        if not a.strong and a.config is None and not a.no_upload_leg and NCTX >= 2:
            # the boundary handing over HOST buffers: every step takes a FRESH batch (three different read sets in turn) as concatenated base
            # codes + offsets + lengths in host memory (mtr_upload_batch: 2-bit packing on the calling thread + the copy to the device), runs
            # it and fetches its tables; the upload of step s+1 overlaps the kernels of step s (two contexts), as in the host pipeline
            flats = [mtr_amd._flatten(reads)] + [mtr_amd._flatten([c for _, c in synth.make_reads(WORKLOAD, a.reads, seed=1000 + k)]) for k in range(2)]
            def with_upload(steps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for s in range(steps):
                    e = engs[s % 2]
                    e.upload_flat(*flats[s % 3])
                    e.run_async()
                    if s > 0:
                        engs[(s - 1) % 2].wait(); engs[(s - 1) % 2].fetch_packed_nocopy()
                engs[(steps - 1) % 2].wait(); engs[(steps - 1) % 2].fetch_packed_nocopy()
                torch.cuda.synchronize()
                return time.perf_counter() - t0
            with_upload(2)
            dt_upload = with_upload(a.steps)
            for e in engs[:2]:
                e.upload(reads)                          # (the legs below expect the headline batch resident,
            engs[0].run()                                #  and the counters of its last launch)

    if rank == 0:
        cnt = eng.counters()
        n_local = len(reads)
        value = n_job * a.steps / dt
        # algorithmic bytes of one launch (this rank's batch): the reference's DP cells = computed + answered from the memo
        sumL4 = sum((len(r) + 3) // 4 for r in reads)
        cells = ref_cnt["dp_cells"] + ref_cnt["revise_dp_cells"] + ref_cnt["memo_cells"]
        b_alg = sumL4 + 576 * ref_cnt["records"] + (cells + 1) // 2
        k2_avg_s = float(np.mean([k["k2_units"] for k in k2_ms])) / 1e3
        achieved_gbs = b_alg / k2_avg_s / 1e9
        valu_ops = cells * OPS_PER_CELL / k2_avg_s
        step_s = dt / a.steps
        phases = sorted({p for k in k2_ms for p in k if p.startswith("chain_")})
        phase_ms = {p[6:]: float(np.mean([k.get(p, 0.0) for k in k2_ms])) for p in phases}
        phase_ms_alone = {p[6:]: float(np.mean([k.get(p, 0.0) for k in sync_k2])) for p in phases} if sync_k2 else None
        prof, prof_tag = profiled_counters(a.config)
        traffic = prof.get("k2_hbm_bytes_per_launch_fetch_x2") if prof else None
        valu_insts = prof.get("SQ_INSTS_VALU_per_launch") if prof else None
        wire_bytes = len(eng.fetch_packed()[0]) if world == 1 else None
        if a.strong:
            workload = (f"{a.strong}: ONE set of {n_job} synthetic Nanopore reads (unit 50-200 x 10 copies, L ~ 2 kb), contiguous blocks of equal "
                        f"read count over {world} GPU(s)")
        elif a.config == "c3":
            workload = (f"c3 (BASELINE config 3 shape; the Badread sets are absent): {n_local} synthetic Nanopore reads per GPU, unit 200 x 200 copies, "
                        f"200-base flanks, mean L {np.mean([len(r) for r in reads]):.0f}")
        elif a.config == "c2":
            workload = (f"c2 (BASELINE config 2): {n_local} synthetic Nanopore reads per GPU, unit 100 x 10 copies, 100-base flanks, "
                        f"mean L {np.mean([len(r) for r in reads]):.0f}, error profile sub 1.6/ins 9.0/del 3.8 %")
        else:
            workload = (f"{WORKLOAD}: {n_local} synthetic Nanopore reads per GPU, unit 100 x 10 copies, 500-base flanks, "
                        f"mean L {np.mean([len(r) for r in reads]):.0f}, error profile sub 1.6/ins 9.0/del 3.8 %")
        # ---- roofline of the DOMINANT kernel (mtr_k_revise_quads: four revisions per wavefront), by its own duration: HIP events around its launches
        # on the launch stream (mtr_get_kernel_times id 8), (i) in the warm-up launches, which have the GPU to themselves - the figure
        # `rocprofv3 --kernel-trace --stats` of a lone launch reproduces (profiles/) and the one `frac` uses -, (ii) in the pipelined timed region,
        # where a launch shares the chip with its neighbour (reported, not used).  Its algorithmic work: the cells (rows x unit) of the vote DPs and
        # re-alignments it ran (no padding: what a pass computes beyond them shows as bytes per cell in quad_passes) x 7 integer operations
        def kavg(ks, name):
            v = [k[name] for k in ks if name in k]
            return float(np.mean(v)) if v else None
        dom = "mtr_k_revise_quads"
        dom_lone_ms, dom_timed_ms = kavg(sync_k2, "kernel_" + dom), kavg(k2_ms, "kernel_" + dom)
        dp2_lone_ms = kavg(sync_k2, "kernel_mtr_k_dp2_quads")
        dom_cells = cnt["qpass_cells_rev"]
        launch_lone_s = (float(np.mean([k["k2_units"] for k in sync_k2])) / 1e3) if sync_k2 else None
        computed_cells = cnt["dp_cells"] + cnt["revise_dp_cells"]
        if dom_lone_ms or dom_timed_ms:
            dms = dom_lone_ms or dom_timed_ms
            dom_ops = dom_cells * OPS_PER_CELL / (dms / 1e3)
            roofline = {"bound": "valu-issue", "kernel": dom,
                        "achieved": dom_ops / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": dom_ops / VALU_PEAK_LANE_OPS,
                        "duration_ms": dms, "duration_source": ("HIP events around the kernel's launches, warm-up launches that have the GPU to themselves" if dom_lone_ms
                                                                else "HIP events around the kernel's launches in the timed region (no warm-up launch ran)"),
                        "duration_ms_in_timed_region": dom_timed_ms,
                        "cells_per_launch": dom_cells, "ops_per_cell": OPS_PER_CELL,
                        "frac_definition": "rounds 5-6: the dominant kernel's own cells x 7 / its lone duration / VALU peak (VERDICT r4's definition); the whole-launch "
                                           "figure that rounds 1-4 called frac is whole_launch.frac_lone_launch / frac_by_step (ADVICE r5: not comparable across that change)",
                        "cells_note": "cells (rows x unit) of the vote DPs and re-alignments the kernel ran in its passes of up to eight DPs per wavefront"}
        else:       # a batch whose chain runs one DP per wavefront (small batches) or the per-read kernel: no kernel of its own to name - the launch
            roofline = {"bound": "valu-issue", "kernel": "one launch (the chain ran one DP per wavefront: no dominant several-DPs-per-wavefront kernel)",
                        "achieved": valu_ops / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s", "frac": valu_ops / VALU_PEAK_LANE_OPS,
                        "duration_ms": k2_avg_s * 1e3, "cells_per_launch": cells, "ops_per_cell": OPS_PER_CELL}
        roofline.update({
            "second_kernel": ({"kernel": "mtr_k_dp2_quads", "duration_ms": dp2_lone_ms, "cell_pairs_per_launch": cnt["qpass_cells_dp2"],
                               "frac": 2 * cnt["qpass_cells_dp2"] * OPS_PER_CELL / (dp2_lone_ms / 1e3) / VALU_PEAK_LANE_OPS,
                               "note": "two-parameter alignments: a cell pair = the cell of both parameter sets, 2 x 7 operations"} if dp2_lone_ms else None),
            "whole_launch": {"kernel": "the staged chain (mtr_k1_ranges; per pass mtr_k_walks, mtr_k_walks_k, mtr_k_gather, mtr_k_dp2_quads, mtr_k_select, mtr_k_polish, "
                                       "mtr_k_rev_share, mtr_k_revise_quads, mtr_k_finish; mtr_k_pass_mark between the passes; mtr_k_replay)",
                             "cells_reference": cells, "cells_computed": computed_cells,
                             "frac_by_step": cells * OPS_PER_CELL / step_s / VALU_PEAK_LANE_OPS,
                             "frac_lone_launch": (cells * OPS_PER_CELL / launch_lone_s / VALU_PEAK_LANE_OPS) if launch_lone_s else None,
                             "frac_computed_cells": computed_cells * OPS_PER_CELL / step_s / VALU_PEAK_LANE_OPS,
                             "note": "cells_reference = the reference's DP cells of the launch (memo_cells of them answered without a DP); by the driver-visible step, by a "
                                     "launch that has the GPU to itself, and the cells the kernels computed by the step"},
            "issue": ({"by_lone_launch": (valu_insts * 2 / (1024 * launch_lone_s * 2.4e9)) if launch_lone_s else None,
                       "by_step": valu_insts * 2 / (1024 * step_s * 2.4e9),
                       "SQ_INSTS_VALU_per_launch": valu_insts,
                       "note": "VALU issue utilisation = SQ_INSTS_VALU of one launch (committed PMC pass of THIS workload) x 2 cycles / (1 024 SIMDs x duration x 2.4 GHz), by the "
                               "duration of a launch that has the GPU to itself and by the driver-visible step - not by the profiler's serialised launches (round 5's 0.170)"}
                      if valu_insts else None),
            "slot_time": slot_time(prof),
            "hbm": {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                    "frac_by_step": b_alg / step_s / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": b_alg, "spilled": "every DP of the reference is counted as spilled (SURVEY.md 8d's upper figure)"},
            "traffic": traffic, "traffic_source": prof_tag,
            "note": "bound by VALU instruction issue: row-serial integer max-plus recurrence; issue, slot_time and traffic (HBM bytes per launch of the whole chain, "
                    "FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes) come from the committed profile of THIS workload and are null without one"})
        out = {
            "metric": "reads/sec, 2 kb Nanopore synthetic" if a.config is None else "reads/sec, config 3 (unit 200 x 200 copies, L ~ 42 kb; secondary line)" if a.config == "c3"
                      else "reads/sec, config 2 (1 000 reads, unit 100 x 10 copies, L ~ 1.25 kb; secondary line)",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if a.strong else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "exchange": ({"backend": backend, "ranks": world, "collectives": "all_gather of the table sizes (first exchange), gather of fixed-size wire-form buffers to rank 0 per step",
                          "forced_on_one_rank": world == 1} if dist_on else None),
            "config": {"workload": workload, "reads_per_gpu": n_local,
                       "parallelism": f"reads sharded over {world} GPU(s), wire-form record tables gathered to rank 0" if world > 1
                                      else "1 GPU, record tables fetched to pinned host memory in wire form",
                       "batches_in_flight": NCTX},
            "value_definition": "steps end with the record tables in host memory (wire form, mtr_fetch_results_packed)" if not dist_on
                                else "steps end with the record tables of every rank in rank 0's pinned host memory (wire form: RCCL gather to rank 0's GPU, one copy to the host)",
            "ms_per_read": dt / a.steps * 1e3 / max(n_job, 1),
            "kernels_ms": {"launch": k2_avg_s * 1e3, "phases": phase_ms,
                           "note": "HIP-event durations over the timed region; consecutive steps overlap on the GPU, so a launch shares the chip with its neighbour"},
            "kernels_ms_alone": {"launch": float(np.mean([k["k2_units"] for k in sync_k2])) if sync_k2 else None, "phases": phase_ms_alone, "mode": lone_mode,
                                 "note": "a launch that has the GPU to itself (the warm-up steps)"},
            "kernels_mode_timed": eng.last_mode(),
            "work_per_launch": {k: cnt[k] for k in ("dp_calls", "dp_cells", "dp_rows", "revise_dp_calls", "revise_dp_cells", "memo_hits", "memo_cells",
                                                    "kmer_tables", "tables_skipped", "kmer_lookups", "ranges_executed", "ranges_searched", "records", "traceback_steps", "revisions_shared", "reads_sent_back")},
            "quad_passes": {"note": "the DP passes that carry several DPs per wavefront (dp_quad.hip.inc: four two-parameter alignments, eight one-parameter revision DPs): bytes of cell "
                                    "matrix written (every row of every DP of the pass up to the pass's longest member) per cell of the DPs the pass was run for; alignments: one byte "
                                    "holds the cell of both parameter sets; revisions: a byte holds two rows",
                            "alignments_bytes_per_cell_pair": (cnt["qpass_bytes_dp2"] / cnt["qpass_cells_dp2"]) if cnt["qpass_cells_dp2"] else None,
                            "revisions_bytes_per_cell": (cnt["qpass_bytes_rev"] / cnt["qpass_cells_rev"]) if cnt["qpass_cells_rev"] else None,
                            "alignments_bytes": cnt["qpass_bytes_dp2"], "revisions_bytes": cnt["qpass_bytes_rev"]},
            "reference_work_per_launch": {k: ref_cnt[k] for k in ("dp_calls", "dp_cells", "revise_dp_calls", "revise_dp_cells", "memo_hits", "memo_cells", "ranges_executed", "records")},
            "roofline": roofline,
        }
        out["chain_health"] = {"reads_sent_back": cnt["reads_sent_back"], "ok": cnt["reads_sent_back"] == 0,
                               "note": "reads the two-pass chain had to send back to the per-read kernel (its mark pass missed a range the reference's loop reaches): must be 0"}
        if cnt["reads_sent_back"] != 0:
            print(f"bench.py: WARNING: the chain sent {cnt['reads_sent_back']} reads back to the per-read kernel on the default path", file=sys.stderr)
        if world == 1:
            out["value_kernel"] = n_job * a.steps / dt_kernel
            if dt_upload:
                out["value_with_upload"] = n_job * a.steps / dt_upload
                out["value_with_upload_definition"] = ("every step takes a fresh batch as host buffers (base codes, offsets, lengths): 2-bit packing on the calling thread + "
                                                       "copy to the device (mtr_upload_batch) + kernels + tables fetched to pinned host memory; PCIe-inclusive, never `value`")
            out["boundary"] = {"wire_bytes_per_step": wire_bytes, "record_struct_bytes_per_step": 2560 * cnt["records"],
                               "fetch_cost_ms_per_step": (dt - dt_kernel) / a.steps * 1e3}
        if a.strong and "blobs" in gathered:
            h = hashlib.sha256()
            for bl in gathered["blobs"]:
                h.update(bl)
            known = None
            try:
                with open(os.path.join(ROOT, "tests", "golden", f"c4_{'100k' if n_job == 100000 else n_job}_wire.json")) as fh:
                    known = json.load(fh)
            except Exception:
                pass
            out["strong"] = {"ranks_seen": sum(1 for n in gathered["reads"] if n > 0), "reads_per_rank": gathered["reads"],
                             "records": sum(gathered["records"]), "sha256": h.hexdigest(),
                             "matches_oracle": (known is not None and known["sha256"] == h.hexdigest() and known["records"] == sum(gathered["records"]))
                                               if known else None,
                             "known_answer": "tests/golden/c4_100k_wire.json (CPU oracle)" if known else None}
        elif a.strong and world == 1:
            data, counts = eng.fetch_packed()
            known = None
            try:
                with open(os.path.join(ROOT, "tests", "golden", f"c4_{'100k' if n_job == 100000 else n_job}_wire.json")) as fh:
                    known = json.load(fh)
            except Exception:
                pass
            hx = hashlib.sha256(data).hexdigest()
            out["strong"] = {"ranks_seen": 1, "reads_per_rank": [n_local], "records": int(counts.sum()), "sha256": hx,
                             "matches_oracle": (known["sha256"] == hx) if known else None,
                             "known_answer": "tests/golden/c4_100k_wire.json (CPU oracle)" if known else None}
        if a.config is not None and world == 1:
            # the launch's record stream against the CPU oracle's known answer (tests/golden/make_c4_wire_hash.py --config c3 -n 100 / --config c2 -n 1000)
            data, counts = eng.fetch_packed()
            known = None
            try:
                with open(os.path.join(ROOT, "tests", "golden", f"{a.config}_{n_local}_wire.json")) as fh:
                    known = json.load(fh)
            except Exception:
                pass
            hx = hashlib.sha256(data).hexdigest()
            out["matches_oracle"] = (known["sha256"] == hx and known["records"] == int(counts.sum())) if known else None
            out["record_stream"] = {"records": int(counts.sum()), "wire_bytes": len(data), "sha256": hx,
                                    "known_answer": f"tests/golden/{a.config}_{n_local}_wire.json (CPU oracle, pinned to the reference)" if known else None}
        if world == 1 and not a.no_latency:
            lat = []
            e2 = mtr_amd.Engine(device=local_rank)
            for i in range(33):
                t1 = time.perf_counter()
                e2.process([reads[i]])
                lat.append((time.perf_counter() - t1) * 1e3)
            e2.close()
            out["latency_ms_per_read_p50"] = float(np.median(lat[1:]))
        if world == 1 and not a.no_cli and not a.strong and a.config == "c3":
            for e in engs:
                e.close()
            engs.clear()
            torch.cuda.empty_cache()
            ca = cli_rate(reads, len(reads), ["-a"])
            out["value_cli"] = ca.get("reads_per_s")
            out["cli"] = {"note": "mtr_amd/host/mTR -a <fasta> > /dev/null (alignment of every reported repeat on), wall clock incl. process start and HIP initialisation; best of 3",
                          "with_alignments": ca}
        elif world == 1 and not a.no_cli and not a.strong and a.config == "c2":
            for e in engs:
                e.close()
            engs.clear()
            torch.cuda.empty_cache()
            ca = cli_rate(reads, len(reads))
            out["value_cli"] = ca.get("reads_per_s")
            out["cli"] = {"note": "mtr_amd/host/mTR <fasta> > /dev/null, wall clock incl. process start and HIP initialisation; best of 3", "one_batch": ca}
        elif world == 1 and not a.no_cli and not a.strong:
            for e in engs:                              # the command line brings its own contexts: free this process's memory first
                e.close()
            engs.clear()
            torch.cuda.empty_cache()
            # this process has just given ~50 GB of device memory back: the driver reclaims it in the background, and a child that allocates meanwhile waits
            # for it (measured: the second context's first launch of a run 0.64 s instead of 0.01 s, all three runs of the leg).  A pause and one untimed run first.
            time.sleep(2.0)
            cli_rate(reads, 2000)
            c1 = cli_rate(reads, len(reads))
            c10 = cli_rate(reads, 10 * len(reads))
            out["value_cli"] = c10.get("reads_per_s")
            out["cli"] = {"note": "mtr_amd/host/mTR <fasta> > /dev/null, wall clock incl. process start and HIP initialisation; best of 3",
                          "one_batch": c1, "ten_batches": c10, "empty_hip_program": empty_hip_program()}
            lr = launcher_rate(reads, 10 * len(reads), 1)
            out["launcher"] = lr
            out["value_launcher"] = lr.get("reads_per_s")
            out["launcher_rccl_forced"] = launcher_rate(reads, 10 * len(reads), 1, force_rccl=True)
            if not a.no_secondary:
                # BASELINE configs 2, 4 and 5 as jobs of the command line (fresh child processes), and the host pipeline's ceiling behind eight GPUs (no GPU touched)
                try:
                    out["baseline_configs_cli"] = baseline_configs_through_the_command_line(1)
                except Exception as ex:
                    out["baseline_configs_cli"] = {"error": repr(ex)[:300]}
                out["host_ceiling"] = host_ceiling_leg()
            try:        # the command line's own part: from the first device context (the runtime is up) to the end of the run
                st = c10["stamps_s"]
                own = st.get("run stopped", st["everything printed"]) - st["first device context created"]
                out["cli"]["ten_batches_after_runtime_start"] = {"seconds": own, "reads_per_s": c10["reads"] / own}
            except Exception:
                pass
        if world == 1 and a.cpu_sample > 0 and not a.strong:
            out["cpu_baseline"] = cpu_baseline(reads, a.cpu_sample if a.config is None else min(a.cpu_sample, 6 if a.config == "c3" else 1000))     # (a config-3 read is ~2.6 s of CPU)
            if out["cpu_baseline"].get("ms_per_read_p50") and out.get("latency_ms_per_read_p50"):
                out["latency_p50_vs_cpu"] = out["cpu_baseline"]["ms_per_read_p50"] / out["latency_ms_per_read_p50"]
            out["speedup_vs_cpu_1core"] = value / out["cpu_baseline"]["value"]
            if "all_cores" in out["cpu_baseline"]:
                out["speedup_vs_cpu_all_cores"] = value / out["cpu_baseline"]["all_cores"]["value"]
            if out.get("value_cli"):
                out["speedup_cli_vs_cpu_1core"] = out["value_cli"] / out["cpu_baseline"]["value"]
        if world == 1 and not dist_on and not a.strong and a.config is None and not a.no_secondary:
            # BASELINE config 3 (unit 200 x 200 copies, L ~ 42 kb, -a on in its command-line leg) in the driver-run line: measured by a
            # child process after this one has given its contexts back, its whole JSON line kept under secondary.c3
            for e in engs:
                e.close()
            engs.clear()
            torch.cuda.empty_cache()
            out["secondary"] = {}
            for cfg, steps in (("c2", "40"), ("c3", "24")):
                t0 = time.perf_counter()
                try:
                    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", steps, "--warmup", "3", "--cpu-sample", str(min(a.cpu_sample, 6 if cfg == "c3" else 1000))]
                                       + (["--no-cli"] if a.no_cli else []), capture_output=True, text=True, cwd=ROOT, timeout=420)
                    try:
                        sec = json.loads(p.stdout.strip().splitlines()[-1])
                    except Exception:
                        sec = {"error": (p.stderr or p.stdout)[-400:], "returncode": p.returncode}
                except subprocess.TimeoutExpired as ex:         # a stuck child must not take the driver-run line with it
                    sec = {"error": "timeout after 420 s: " + ((ex.stderr or b"")[-300:].decode(errors="replace") if isinstance(ex.stderr, bytes) else str(ex.stderr or "")[-300:])}
                sec["wall_s"] = time.perf_counter() - t0
                out["secondary"][cfg] = sec
    for e in engs:
        e.close()
    engs.clear()
    if rank == 0 and launcher_pre is not None:
        out["launcher"] = launcher_pre
        out["value_launcher"] = launcher_pre.get("reads_per_s")
        out.update(pre)                                 # launcher_rccl_forced, baseline_configs_cli, host_ceiling: measured before any rank touched its GPU
        try:
            os.remove(os.path.join(tempfile.gettempdir(), f"mtr_bench_launcher_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}.done"))
        except OSError:
            pass
    if rank == 0:
        print(json.dumps(out), file=report)
        report.flush()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
