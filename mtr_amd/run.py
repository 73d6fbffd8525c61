"""python -m mtr_amd.run --gpus N [-a] [-p] [-m ratio] [-B] <fasta> [<fasta> ...] — mTR over the N GPUs of one node.

What the reference does per file (handle_one_file.c:281-287: read after read, handle_one_read, chaining, print) is cut
into chunks of whole reads and spread over one process per GPU (BASELINE configs 4 and 5):

  * every rank maps the file(s), plans the same chunks (mtr_amd/host/pipeline.c) and parses + runs only its own through
    the C-ABI (libmtr_hip.so: two contexts per GPU, packed uploads, wire-form fetch);
  * ONE exchange per round: the ranks' results (IDs, counts, record tables in wire form; with -a also chains, alignment
    paths and 2-bit bases) are gathered to rank 0 — torch.distributed, backend "nccl" = RCCL over xGMI on the GPU box,
    "gloo" in CPU tests; two size all-gathers + one padded gather, no other collective;
  * rank 0 chains and prints in input order (mtr_amd/host/chain.c, print.c).  stdout = the reference's, byte for byte.

One file: chunk c -> rank c % N, round c / N (streams; the next round runs on the GPUs while this one is gathered and
printed).  Several files (test_multiple_TRs/test.sh: 15 files of one read each, 2.6-140 kb): chunks go longest-first to
the least loaded rank (LPT) in one round, output in command-line order.  -B (the reference's file-order behaviour): a rank
replays the reads before its chunks through mtr_file_state_skip.

All logic is in C (mtr_amd/host/libmtr_host.so); this file starts the processes BEFORE anything touches a GPU and moves
bytes.  Without --gpus, or under torchrun (RANK set), it runs as one rank of an existing job.
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_DIR = os.path.join(HERE, "host")
HOST_LIB = os.path.join(HOST_DIR, "libmtr_host.so")


class Opts(C.Structure):
    _fields_ = [("print_alignment", C.c_int), ("manhattan", C.c_int), ("file_order", C.c_int), ("device", C.c_int),
                ("min_match_ratio", C.c_float), ("rank", C.c_int), ("world", C.c_int), ("lpt", C.c_int),
                ("chunk_bytes", C.c_size_t), ("parse_threads", C.c_int), ("print_threads", C.c_int),
                ("engine_lib", C.c_char_p), ("gather", C.c_void_p), ("contexts", C.c_int)]


def load_host():
    """libmtr_host.so = mtr_amd/host/*.c (built with make; plain C, no GPU code)."""
    if not os.path.exists(HOST_LIB) or any(os.path.getmtime(os.path.join(HOST_DIR, f)) > os.path.getmtime(HOST_LIB)
                                           for f in os.listdir(HOST_DIR) if f.endswith((".c", ".h"))):
        subprocess.run(["make", "-s", "-C", HOST_DIR, "libmtr_host.so"], check=True)
    lib = C.CDLL(HOST_LIB)
    lib.mtrh_run_start.restype = C.c_void_p
    lib.mtrh_run_start.argtypes = [C.POINTER(Opts), C.POINTER(C.c_char_p), C.c_int]
    for f in ("mtrh_run_n_chunks", "mtrh_run_n_rounds"):
        getattr(lib, f).restype = C.c_int
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.mtrh_run_owner.restype = C.c_int
    lib.mtrh_run_owner.argtypes = [C.c_void_p, C.c_int]
    lib.mtrh_run_round_blob.restype = C.c_void_p
    lib.mtrh_run_round_blob.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)]
    lib.mtrh_run_stop.restype = None
    lib.mtrh_run_stop.argtypes = [C.c_void_p]
    lib.mtrh_printer_start_stdout.restype = C.c_void_p
    lib.mtrh_printer_start_stdout.argtypes = [C.c_int]
    lib.mtrh_printer_start_fd.restype = C.c_void_p
    lib.mtrh_printer_start_fd.argtypes = [C.c_int, C.c_int]
    lib.mtrh_print_round.restype = C.c_int
    lib.mtrh_print_round.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int]
    lib.mtrh_printer_finish.restype = C.c_int
    lib.mtrh_printer_finish.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    return lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="python -m mtr_amd.run", add_help=True,
                                 description="reference mTR's command line over N GPUs (one process per GPU)")
    ap.add_argument("--gpus", type=int, default=0, help="start this many ranks (omit under torchrun / for a single rank)")
    ap.add_argument("-a", dest="align", action="store_true", help="print the alignment of every reported repeat")
    ap.add_argument("-p", dest="pearson", action="store_true", help="Pearson distance instead of Manhattan")
    ap.add_argument("-m", dest="ratio", type=float, default=0.6, help="minimum match ratio (0..1)")
    ap.add_argument("-B", dest="file_order", action="store_true", help="the reference's whole-file behaviour (file-order mode)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default: nccl with a GPU, else gloo)")
    ap.add_argument("--engine-lib", default=None, help="library implementing include/mtr_hip.h (default: mtr_amd/libmtr_hip.so)")
    ap.add_argument("--chunk-bytes", type=int, default=0, help="FASTA bytes per chunk (default 24 MiB)")
    ap.add_argument("--stats", action="store_true", help="rank 0 reports ranks seen / bytes gathered on stderr")
    ap.add_argument("--force-dist", action="store_true",
                    help="a single rank too joins a process group and goes through the collectives (under torchrun with one rank: RCCL on a one-GPU box)")
    ap.add_argument("fasta", nargs="+")
    a = ap.parse_args(argv)
    if not 0 <= a.ratio <= 1:
        sys.stderr.write("The input minimum match ratio must range from 0 to 1.\n")
        sys.exit(1)
    return a


def spawn(n, argv):
    """One child per rank, started before this process has touched a GPU; the exit status is the ranks' worst.

    Rendezvous: a FileStore in a directory of this launch (init_method file://) - one node is the whole machine, and no port is ever
    chosen ahead of the ranks: a port found by bind-and-close belongs to whoever binds it next, and the ranks need seconds to import
    torch before rank 0 would listen (the EADDRINUSE that reddened round 3's GPU suite)."""
    import shutil
    import tempfile
    import time
    rdzv_dir = tempfile.mkdtemp(prefix=f"mtr_run_{os.getpid()}_")
    procs = []
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MTR_RDZV_FILE=os.path.join(rdzv_dir, "store"))
            for k in ("MASTER_ADDR", "MASTER_PORT"):            # the children rendezvous through the file, whatever the caller's environment says
                env.pop(k, None)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("GPU_MAX_HW_QUEUES", "8")        # RCCL's streams + the two context streams: more than the runtime's default of 4 hardware queues (bench.py)
            procs.append(subprocess.Popen([sys.executable, "-m", "mtr_amd.run", *argv], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        # A rank that dies (a refused allocation in the host library exits the process, a crash, a malformed blob on rank 0) leaves the
        # others waiting in a collective until the communicator times out - minutes.  So: once any rank has ended badly the others get
        # a few seconds to end by themselves, then they are ended.
        codes = [None] * n
        failed_at = None
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    rc = p.poll()
                    if rc is not None:
                        codes[i] = rc
                        if rc != 0 and failed_at is None:
                            failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > 5.0:
                for i, p in enumerate(procs):
                    if codes[i] is None:
                        p.kill()
                        codes[i] = p.wait()
            time.sleep(0.02)
        return max((c if c >= 0 else 128 - c) for c in codes)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
        shutil.rmtree(rdzv_dir, ignore_errors=True)


def join_process_group(dist, backend, rank, world):
    """The ranks of one node meet through a file when this launcher started them (MTR_RDZV_FILE) or when nobody named a master
    (--force-dist in a lone process); under torchrun through the store its agent already listens on (MASTER_ADDR / MASTER_PORT)."""
    delay = float(os.environ.get("MTR_TEST_RDZV_DELAY", "0") or 0)    # tests: rank 0 arrives late (the round-3 failure needed that)
    if delay > 0 and rank == 0:
        import time
        time.sleep(delay)
    path = os.environ.get("MTR_RDZV_FILE")
    own_dir = None
    if not path and "MASTER_PORT" not in os.environ:
        if world > 1:
            # ranks somebody else started (mpirun, srun, a hand-rolled loop) with RANK / WORLD_SIZE but no meeting point: a private store per
            # rank would leave every rank waiting for the others until the store's timeout, without a word
            raise SystemExit("mtr_amd.run: RANK/WORLD_SIZE are set (world size %d) but neither MTR_RDZV_FILE nor MASTER_ADDR/MASTER_PORT say where the "
                             "ranks meet; start the ranks with --gpus N, under torchrun, or export one of the two" % world)
        import tempfile
        own_dir = tempfile.mkdtemp(prefix=f"mtr_run_{os.getpid()}_")      # the lone --force-dist process
        path = os.path.join(own_dir, "store")
    if path:
        dist.init_process_group(backend, init_method="file://" + path, rank=rank, world_size=world)
    else:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return own_dir


def leave_process_group(dist, own_dir):
    dist.destroy_process_group()
    if own_dir:
        import shutil
        shutil.rmtree(own_dir, ignore_errors=True)


def gather_bytes(dist, torch, payload: bytes, rank, world, dev):
    """Variable-length byte strings to rank 0: an all_gather of the sizes, then one padded gather."""
    size = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(s.item()) for s in sizes]
    width = max(max(sizes), 1)
    buf = torch.zeros(width, dtype=torch.uint8, device=dev)
    if payload:
        buf[: len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if out is None:
        return None, sizes
    return [out[r][: sizes[r]].cpu().numpy().tobytes() for r in range(world)], sizes


def worker(a):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # before the HIP runtime starts (ranks started by torchrun rather than by main())
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    lib = load_host()
    dist = torch = dev = None
    device_ordinal = 0
    report_fd = None
    rdzv_own_dir = None
    dist_on = world > 1 or a.force_dist
    backend = None
    if dist_on:
        import torch
        import torch.distributed as dist
        backend = a.backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            ndev = torch.cuda.device_count()
            device_ordinal = local % max(ndev, 1)          # more ranks than GPUs (a rehearsal on one GPU): ranks share
            torch.cuda.set_device(device_ordinal)
            dev = torch.device("cuda", device_ordinal)
        else:
            dev = torch.device("cpu")
            if torch.cuda.is_available():
                device_ordinal = local % max(torch.cuda.device_count(), 1)
        # stdout belongs to the report: whatever a communication library prints when it connects goes to stderr - and RCCL
        # connects (and prints its version banner on fd 1) at the first COLLECTIVE, not here.  So fd 1 stays pointed at stderr for the
        # rest of the process; the report is written to the saved descriptor.
        sys.stdout.flush()
        report_fd = os.dup(1)
        os.dup2(2, 1)
        rdzv_own_dir = join_process_group(dist, backend, rank, world)
    o = Opts(print_alignment=int(a.align), manhattan=int(not a.pearson), file_order=int(a.file_order), device=device_ordinal,
             min_match_ratio=a.ratio, rank=rank, world=world, lpt=int(len(a.fasta) > 1), chunk_bytes=a.chunk_bytes,
             parse_threads=0, print_threads=0, engine_lib=(a.engine_lib.encode() if a.engine_lib else None))
    paths = (C.c_char_p * len(a.fasta))(*[p.encode() for p in a.fasta])
    if a.pearson and rank == 0:
        sys.stderr.write("Pearson's correlation coefficient distance in place of Manhattan distance.\n")
    run = lib.mtrh_run_start(C.byref(o), paths, len(a.fasta))
    ok = 1 if run else 0
    if dist_on:                                             # a rank that could not start (bad file) stops everyone
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    if not ok:
        if run:
            lib.mtrh_run_stop(run)
        if dist_on:
            leave_process_group(dist, rdzv_own_dir)
        return 1
    n_print = min(8, max(1, (os.cpu_count() or 2) // 2))
    printer = None
    if rank == 0:
        printer = lib.mtrh_printer_start_fd(report_fd, n_print) if report_fd is not None else lib.mtrh_printer_start_stdout(n_print)
    n_rounds = lib.mtrh_run_n_rounds(run)
    n_chunks = lib.mtrh_run_n_chunks(run)
    gathered = 0
    for t in range(n_rounds):
        nbytes = C.c_size_t()
        ptr = lib.mtrh_run_round_blob(run, t, C.byref(nbytes))
        payload = C.string_at(ptr, nbytes.value) if nbytes.value else b""
        _libc.free(ptr)
        if dist_on:
            blobs, sizes = gather_bytes(dist, torch, payload, rank, world, dev)
        else:
            blobs, sizes = [payload], [len(payload)]
        round_ok = 1
        if rank == 0:
            gathered += sum(sizes)
            keep = [C.create_string_buffer(b, len(b)) if b else C.create_string_buffer(1) for b in blobs]
            arr = (C.c_void_p * world)(*[C.cast(k, C.c_void_p) for k in keep])
            szs = (C.c_size_t * world)(*sizes)
            if lib.mtrh_print_round(printer, arr, szs, world) < 0:
                sys.stderr.write("internal error: malformed result blob\n")
                round_ok = 0
        if dist_on:                                         # rank 0's verdict on the round reaches everyone: nobody waits in a collective for a rank that left
            okf = torch.tensor([round_ok], dtype=torch.int64, device=dev)
            dist.broadcast(okf, src=0)
            round_ok = int(okf.item())
        if not round_ok:
            lib.mtrh_run_stop(run)
            if dist_on:
                leave_process_group(dist, rdzv_own_dir)
            return 2
    status = 0
    if rank == 0:
        status = lib.mtrh_printer_finish(printer, None)
        if a.stats:
            owners = sorted({lib.mtrh_run_owner(run, c) for c in range(n_chunks)})
            sys.stderr.write(f"[mtr_amd.run] ranks={world} ranks_with_chunks={len(owners)} chunks={n_chunks} rounds={n_rounds} gathered_bytes={gathered}"
                             f" backend={backend or 'none'}\n")
    lib.mtrh_run_stop(run)
    if dist_on:
        st = torch.tensor([status], dtype=torch.int64, device=dev)
        dist.broadcast(st, src=0)
        status = int(st.item())
        dist.barrier()
        leave_process_group(dist, rdzv_own_dir)
    return status


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    a = parse_args(argv)
    if "RANK" not in os.environ and a.gpus > 1:
        child_argv = [x for i, x in enumerate(argv) if not (x == "--gpus" or (i > 0 and argv[i - 1] == "--gpus") or x.startswith("--gpus="))]
        sys.exit(spawn(a.gpus, child_argv))
    sys.stdout.flush()
    sys.exit(worker(a))


if __name__ == "__main__":
    main()
