"""mtr_amd — MI355X (gfx950) implementation of reference mTR's per-read hot path.

Python host-side mirror of the C-ABI in include/mtr_hip.h (ctypes; plain pointers, no torch types).
The product path is libmtr_hip.so only: importing works without a GPU, but creating an Engine
without the library or without a HIP device raises — there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, NamedTuple, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTR_LIB", os.path.join(HERE, "libmtr_hip.so"))   # MTR_LIB: A/B another build of the library
MAX_PERIOD = 500

STATUS = {0: "MTR_OK", 1: "MTR_ERR_NO_DEVICE", 2: "MTR_ERR_BAD_ARG", 3: "MTR_ERR_OOM", 4: "MTR_ERR_HIP",
          5: "MTR_ERR_OVERFLOW", 6: "MTR_ERR_DP_TOO_LARGE"}
COUNTER_NAMES = ["dp_calls", "dp_cells", "dp_rows", "revise_dp_calls", "revise_dp_cells", "kmer_tables", "kmer_lookups",
                 "ranges_candidate", "ranges_executed", "records", "di_passes", "di_positions", "traceback_steps",
                 "undefined_guards", "global_tables", "reserved",
                 "cyc_total", "cyc_dp_fwd", "cyc_dp_tb", "cyc_tab_build", "cyc_seeds", "cyc_walk", "cyc_polish", "cyc_revise_vote",
                 "cyc_slot_copy", "cyc_dp_fwd_rev", "cyc_dp_tb_rev", "cyc_k1_codes", "cyc_k1_passes", "cyc_k1_extract",
                 "cyc_k1_dedup", "cyc_k1_total",
                 "memo_hits", "memo_cells", "tables_skipped", "cyc_tb_refill", "tb_refills", "walk_steps", "walk_slow_steps", "cyc_walk_slow",
                 "walk_calls", "walk_closed", "cyc_walk_fast", "prof43", "prof44", "prof45", "prof46", "prof47",
                 "qpass_bytes_dp2", "qpass_cells_dp2", "qpass_bytes_rev", "qpass_cells_rev", "revisions_shared", "reads_sent_back", "ranges_searched", "prof55"]
# prof43..47, prof55: scratch counters of the profiling builds (-DMTR_PROFILE...); what they count depends on the kernel that wrote them
# (per-read kernel: revisions / unchanged units / accepted rounds; walk kernels: cycles by window width; see the CNT_SPARE* uses in csrc)
EXPORTS = ["mtr_create", "mtr_destroy", "mtr_last_error", "mtr_abi_version", "mtr_process_batch", "mtr_free_results",
           "mtr_upload_batch", "mtr_run_resident", "mtr_fetch_results", "mtr_get_kernel_times", "mtr_get_counters",
           "mtr_test_ranges", "mtr_test_wrap_dp", "mtr_test_last_mode", "mtr_set_trace", "mtr_get_trace",
           "mtr_run_resident_async", "mtr_wait", "mtr_alignments",
           "mtr_file_state_create", "mtr_file_state_destroy", "mtr_upload_batch_in_file", "mtr_file_state_skip",
           "mtr_get_bases_after_read", "mtr_upload_batch_packed", "mtr_fetch_results_packed", "mtr_export_packed_device",
           "mtr_unpack_records", "mtr_pack_records", "mtr_get_first_failed_read"]


class MtrError(RuntimeError):
    pass


class FileState:
    """mtr_file_state: host shadow of what the reads of ONE file leave behind for the reads after them (file-order mode)."""

    def __init__(self):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.mtr_file_state_create(C.byref(h))
        if st != 0:
            raise MtrError(f"mtr_file_state_create: {STATUS.get(st, st)}")
        self.h = h

    def skip(self, reads):
        bases, offs, lens = _flatten(reads)
        st = self.lib.mtr_file_state_skip(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads))
        if st != 0:
            raise MtrError(f"mtr_file_state_skip: {STATUS.get(st, st)}")

    def close(self):
        if self.h:
            self.lib.mtr_file_state_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CRecord(C.Structure):
    _fields_ = [("rep_start", C.c_int32), ("rep_end", C.c_int32), ("repeat_len", C.c_int32), ("rep_period", C.c_int32),
                ("num_freq_unit", C.c_int32), ("num_matches", C.c_int32), ("num_mismatches", C.c_int32),
                ("num_insertions", C.c_int32), ("num_deletions", C.c_int32), ("kmer", C.c_int32), ("match_gain", C.c_int32),
                ("mismatch_penalty", C.c_int32), ("indel_penalty", C.c_int32), ("reserved", C.c_int32),
                ("unit", C.c_char * (MAX_PERIOD + 4)), ("unit_score", C.c_int32 * MAX_PERIOD)]


class CKernelTime(C.Structure):
    _fields_ = [("ms", C.c_float), ("launches", C.c_int32)]


class Record(NamedTuple):
    """The 15 per-repeat arguments of insert_an_alignment_into_set (reference mTR.h:151-168)."""
    rep_start: int
    rep_end: int
    repeat_len: int
    rep_period: int
    num_freq_unit: int
    num_matches: int
    num_mismatches: int
    num_insertions: int
    num_deletions: int
    kmer: int
    match_gain: int
    mismatch_penalty: int
    indel_penalty: int
    unit: str
    unit_score: tuple


_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen libmtr_hip.so; raises if it has not been built (python -m mtr_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise MtrError(f"{path} is missing: build it with `python -m mtr_amd.build` (hipcc, gfx950); there is no CPU fallback")
    lib = C.CDLL(path)
    P = C.POINTER
    lib.mtr_create.argtypes = [C.c_int, C.c_int, C.c_float, P(C.c_void_p)]
    lib.mtr_create.restype = C.c_int
    lib.mtr_destroy.argtypes = [C.c_void_p]
    lib.mtr_destroy.restype = None
    lib.mtr_last_error.argtypes = [C.c_void_p]
    lib.mtr_last_error.restype = C.c_char_p
    lib.mtr_abi_version.restype = C.c_int
    lib.mtr_process_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, P(P(CRecord)), P(P(C.c_int32)), P(C.c_int64)]
    lib.mtr_process_batch.restype = C.c_int
    lib.mtr_free_results.argtypes = [C.c_void_p, C.c_void_p]
    lib.mtr_free_results.restype = None
    lib.mtr_upload_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mtr_upload_batch.restype = C.c_int
    lib.mtr_run_resident.argtypes = [C.c_void_p]
    lib.mtr_run_resident.restype = C.c_int
    lib.mtr_file_state_create.argtypes = [P(C.c_void_p)]
    lib.mtr_file_state_create.restype = C.c_int
    lib.mtr_file_state_destroy.argtypes = [C.c_void_p]
    lib.mtr_file_state_destroy.restype = None
    lib.mtr_upload_batch_in_file.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mtr_upload_batch_in_file.restype = C.c_int
    lib.mtr_file_state_skip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mtr_file_state_skip.restype = C.c_int
    lib.mtr_run_resident_async.argtypes = [C.c_void_p]
    lib.mtr_run_resident_async.restype = C.c_int
    lib.mtr_wait.argtypes = [C.c_void_p]
    lib.mtr_wait.restype = C.c_int
    lib.mtr_test_last_mode.argtypes = [C.c_void_p]
    lib.mtr_test_last_mode.restype = C.c_int32
    lib.mtr_fetch_results.argtypes = [C.c_void_p, P(P(CRecord)), P(P(C.c_int32)), P(C.c_int64)]
    lib.mtr_fetch_results.restype = C.c_int
    lib.mtr_get_kernel_times.argtypes = [C.c_void_p, P(CKernelTime), C.c_int32]
    lib.mtr_get_kernel_times.restype = C.c_int
    lib.mtr_alignments.argtypes = [C.c_void_p, C.c_int32, P(C.c_int32), C.c_void_p, P(P(C.c_uint8)), P(P(C.c_int64)), P(P(C.c_int32))]
    lib.mtr_alignments.restype = C.c_int
    lib.mtr_get_counters.argtypes = [C.c_void_p, P(C.c_int64), C.c_int32]
    lib.mtr_get_counters.restype = C.c_int
    lib.mtr_test_ranges.argtypes = [C.c_void_p, P(P(C.c_int32)), P(P(C.c_int32)), P(P(C.c_int32)), P(P(C.c_int32)), P(P(C.c_uint64)), P(C.c_int64)]
    lib.mtr_test_ranges.restype = C.c_int
    lib.mtr_test_wrap_dp.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 9
    lib.mtr_test_wrap_dp.restype = C.c_int
    lib.mtr_upload_batch_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mtr_upload_batch_packed.restype = C.c_int
    lib.mtr_fetch_results_packed.argtypes = [C.c_void_p, C.c_int32, P(C.c_void_p), P(C.c_int64), P(C.c_void_p), P(C.c_int64)]
    lib.mtr_fetch_results_packed.restype = C.c_int
    lib.mtr_export_packed_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, P(C.c_int64), P(C.c_int64)]
    lib.mtr_export_packed_device.restype = C.c_int
    lib.mtr_unpack_records.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.mtr_unpack_records.restype = C.c_int
    lib.mtr_pack_records.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    lib.mtr_pack_records.restype = C.c_int64
    lib.mtr_get_first_failed_read.argtypes = [C.c_void_p, P(C.c_int32)]
    lib.mtr_get_first_failed_read.restype = C.c_int
    lib.mtr_set_trace.argtypes = [C.c_void_p, C.c_int32]
    lib.mtr_set_trace.restype = C.c_int
    lib.mtr_get_trace.argtypes = [C.c_void_p, P(P(C.c_int32)), P(C.c_int64)]
    lib.mtr_get_trace.restype = C.c_int
    _lib = lib
    return lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _flatten(reads: Sequence[np.ndarray]):
    lens = np.array([len(r) for r in reads], dtype=np.int32)
    offs = np.zeros(len(reads), dtype=np.int64)
    if len(reads) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    bases = np.concatenate([np.asarray(r, dtype=np.uint8) for r in reads]) if len(reads) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(bases), offs, lens


class Engine:
    """One context per GPU (mtr_create).  manhattan=False is the reference's -p; min_match_ratio its -m."""

    def __init__(self, device: int = 0, manhattan: bool = True, min_match_ratio: float = 0.6):
        self.lib = load_library()
        h = C.c_void_p()
        st = self.lib.mtr_create(device, 1 if manhattan else 0, C.c_float(min_match_ratio), C.byref(h))
        if st != 0:
            raise MtrError(f"mtr_create failed: {STATUS.get(st, st)} (a HIP device is required; there is no CPU fallback)")
        self.h = h
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.mtr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int, what: str):
        if st != 0:
            raise MtrError(f"{what}: {STATUS.get(st, st)}: {self.lib.mtr_last_error(self.h).decode(errors='replace')}")

    # ---- the batch edge -------------------------------------------------------------------------------
    def upload(self, reads: Sequence[np.ndarray], file_state: "FileState | None" = None):
        """file_state: the reads are the next reads of that file (file-order mode, mtr_upload_batch_in_file)"""
        bases, offs, lens = _flatten(reads)
        self._keep = (bases, offs, lens)
        if file_state is None:
            self._check(self.lib.mtr_upload_batch(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads)), "mtr_upload_batch")
        else:
            self._check(self.lib.mtr_upload_batch_in_file(self.h, file_state.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads)),
                        "mtr_upload_batch_in_file")
        self.n_reads = len(reads)

    def upload_flat(self, bases: np.ndarray, offs: np.ndarray, lens: np.ndarray):
        """mtr_upload_batch on host buffers the caller already holds in the boundary's form (concatenated base codes, offsets, lengths):
        the library packs them to 2 bit/base on the calling thread and copies the image to the device"""
        self._keep = (bases, offs, lens)
        self._check(self.lib.mtr_upload_batch(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(lens)), "mtr_upload_batch")
        self.n_reads = len(lens)

    def process_in_file(self, reads: Sequence[np.ndarray], file_state: "FileState") -> List[List[Record]]:
        """the next reads of a file under the reference's whole-file behaviour (include/mtr_hip.h, file-order mode)"""
        self.upload(reads, file_state)
        self.run()
        return self.fetch()

    def run(self):
        self._check(self.lib.mtr_run_resident(self.h), "mtr_run_resident")

    def run_async(self):
        """enqueue K1 + K2 on the context's stream without waiting (mtr_run_resident_async)"""
        self._check(self.lib.mtr_run_resident_async(self.h), "mtr_run_resident_async")

    def wait(self):
        self._check(self.lib.mtr_wait(self.h), "mtr_wait")

    def last_mode(self) -> str:
        """how the last launch ran (mtr_test_last_mode)"""
        return {0: "per-read kernel", 2: "staged chain"}.get(int(self.lib.mtr_test_last_mode(self.h)), "?")

    def fetch(self) -> List[List[Record]]:
        recs = C.POINTER(CRecord)()
        cnts = C.POINTER(C.c_int32)()
        total = C.c_int64()
        self._check(self.lib.mtr_fetch_results(self.h, C.byref(recs), C.byref(cnts), C.byref(total)), "mtr_fetch_results")
        try:
            return self._unpack(recs, cnts, self.n_reads)
        finally:
            self.lib.mtr_free_results(recs, cnts)

    def process(self, reads: Sequence[np.ndarray]) -> List[List[Record]]:
        """mtr_process_batch: reads = sequences of base codes 0..3; returns per read its records in insertion order."""
        bases, offs, lens = _flatten(reads)
        recs = C.POINTER(CRecord)()
        cnts = C.POINTER(C.c_int32)()
        total = C.c_int64()
        self.n_reads = len(reads)                       # (before the call: after MTR_ERR_DP_TOO_LARGE the reads before the failing one are still fetched)
        self._check(self.lib.mtr_process_batch(self.h, bases.ctypes.data, offs.ctypes.data, lens.ctypes.data, len(reads),
                                               C.byref(recs), C.byref(cnts), C.byref(total)), "mtr_process_batch")
        try:
            return self._unpack(recs, cnts, len(reads))
        finally:
            self.lib.mtr_free_results(recs, cnts)

    @staticmethod
    def _unpack(recs, cnts, n) -> List[List[Record]]:
        out, p = [], 0
        for i in range(n):
            lst = []
            for _ in range(cnts[i]):
                r = recs[p]
                p += 1
                per = r.rep_period
                lst.append(Record(r.rep_start, r.rep_end, r.repeat_len, per, r.num_freq_unit, r.num_matches, r.num_mismatches,
                                  r.num_insertions, r.num_deletions, r.kmer, r.match_gain, r.mismatch_penalty, r.indel_penalty,
                                  r.unit.decode(), tuple(r.unit_score[:max(0, min(per, MAX_PERIOD))])))
            out.append(lst)
        return out

    def upload_packed(self, reads: Sequence[np.ndarray]):
        """the host's own packing (mtr_upload_batch_packed): reads -> the 2-bit device image, packed here with numpy"""
        lens = np.array([len(r) for r in reads], dtype=np.int32)
        nw = lens.astype(np.int64) // 16 + 4
        woff = np.zeros(len(reads), np.int64)
        if len(reads) > 1:
            woff[1:] = np.cumsum(nw[:-1])
        packed = np.zeros(int(nw.sum()), np.uint32)
        for i, r in enumerate(reads):
            packed[woff[i]: woff[i] + nw[i]] = pack_read(np.asarray(r, np.uint8))
        self._keep = (packed, woff, lens)
        self._check(self.lib.mtr_upload_batch_packed(self.h, packed.ctypes.data, len(packed), woff.ctypes.data, lens.ctypes.data, len(reads)),
                    "mtr_upload_batch_packed")
        self.n_reads = len(reads)

    def fetch_packed(self, limit: int = -1):
        """mtr_fetch_results_packed: (wire blob bytes, counts int32[n]); the blob is copied out of the context's pinned staging"""
        blob, cnts = C.c_void_p(), C.c_void_p()
        nbytes, total = C.c_int64(), C.c_int64()
        self._check(self.lib.mtr_fetch_results_packed(self.h, limit, C.byref(blob), C.byref(nbytes), C.byref(cnts), C.byref(total)), "mtr_fetch_results_packed")
        n = self.n_reads if limit < 0 else min(limit, self.n_reads)
        counts = np.ctypeslib.as_array(C.cast(cnts, C.POINTER(C.c_int32)), shape=(max(n, 1),))[:n].copy() if n > 0 and cnts.value else np.zeros(0, np.int32)
        data = C.string_at(blob, nbytes.value) if nbytes.value else b""
        return data, counts

    def fetch_packed_nocopy(self):
        """mtr_fetch_results_packed, leaving the wire form where the boundary hands it over (the context's pinned host
        buffer): returns (bytes, records)"""
        blob, cnts = C.c_void_p(), C.c_void_p()
        nbytes, total = C.c_int64(), C.c_int64()
        self._check(self.lib.mtr_fetch_results_packed(self.h, -1, C.byref(blob), C.byref(nbytes), C.byref(cnts), C.byref(total)), "mtr_fetch_results_packed")
        return int(nbytes.value), int(total.value)

    def fetch_via_wire(self) -> List[List[Record]]:
        """the records of the last run through the wire form and mtr_unpack_records (must equal fetch())"""
        data, counts = self.fetch_packed()
        total = int(counts.sum())
        recs = (CRecord * max(total, 1))()
        buf = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
        st = self.lib.mtr_unpack_records(buf, len(data), total, recs)
        if st != 0:
            raise MtrError(f"mtr_unpack_records: {STATUS.get(st, st)}")
        return self._unpack(recs, counts, len(counts))

    def export_packed_device(self, device_ptr: int, capacity_bytes: int):
        """wire-form records into caller-owned device memory (for RCCL); returns (counts int32[n_reads], records, bytes)"""
        counts = np.zeros(self.n_reads, np.int32)
        total, nbytes = C.c_int64(), C.c_int64()
        self._check(self.lib.mtr_export_packed_device(self.h, C.c_void_p(device_ptr), capacity_bytes, counts.ctypes.data, C.byref(total), C.byref(nbytes)),
                    "mtr_export_packed_device")
        return counts, int(total.value), int(nbytes.value)

    def first_failed_read(self) -> int:
        v = C.c_int32()
        self.lib.mtr_get_first_failed_read(self.h, C.byref(v))
        return int(v.value)

    # ---- measurements ------------------------------------------------------------------------------------
    def kernel_times_ms(self):
        kt = (CKernelTime * 10)()
        self._check(self.lib.mtr_get_kernel_times(self.h, kt, 10), "mtr_get_kernel_times")
        out = {"k1_ranges": float(kt[0].ms), "k2_units": float(kt[1].ms)}
        for i, name in enumerate(("ranges", "unit_search", "alignments", "selection", "revisions", "finish_replay")):
            if kt[2 + i].launches:
                out["chain_" + name] = float(kt[2 + i].ms)
        # the two dominant kernels by themselves (HIP events around each launch on the launch stream; both passes of the chain summed)
        for i, name in enumerate(("mtr_k_revise_quads", "mtr_k_dp2_quads")):
            if kt[8 + i].launches:
                out["kernel_" + name] = float(kt[8 + i].ms)
        return out

    def counters(self):
        c = (C.c_int64 * len(COUNTER_NAMES))()
        self._check(self.lib.mtr_get_counters(self.h, c, len(COUNTER_NAMES)), "mtr_get_counters")
        return {n: int(c[i]) for i, n in enumerate(COUNTER_NAMES)}

    # ---- building blocks (parity tests) ----------------------------------------------------------------------
    def test_ranges(self):
        """K1 alone on the uploaded batch -> per read list of (start, end, w, di_bits)."""
        P = C.POINTER
        pc, ps, pe, pw, pd, tot = P(C.c_int32)(), P(C.c_int32)(), P(C.c_int32)(), P(C.c_int32)(), P(C.c_uint64)(), C.c_int64()
        self._check(self.lib.mtr_test_ranges(self.h, C.byref(pc), C.byref(ps), C.byref(pe), C.byref(pw), C.byref(pd), C.byref(tot)), "mtr_test_ranges")
        out, p = [], 0
        for i in range(self.n_reads):
            out.append([(ps[p + t], pe[p + t], pw[p + t], pd[p + t]) for t in range(pc[i])])
            p += pc[i]
        for ptr in (pc, ps, pe, pw, pd):
            _libc.free(C.cast(ptr, C.c_void_p))
        return out

    def test_wrap_dp(self, tasks):
        """tasks: list of (read_idx, qs, qe, unit codes, G, MM, D) -> int32 [n,8] as wrap_around_DP_sub returns."""
        n = len(tasks)
        rd = np.array([t[0] for t in tasks], np.int32)
        qs = np.array([t[1] for t in tasks], np.int32)
        qe = np.array([t[2] for t in tasks], np.int32)
        uo = np.zeros(n + 1, np.int32)
        uo[1:] = np.cumsum([len(t[3]) for t in tasks])
        units = np.ascontiguousarray(np.concatenate([np.asarray(t[3], np.uint8) for t in tasks]))
        g = np.array([t[4] for t in tasks], np.int32)
        m = np.array([t[5] for t in tasks], np.int32)
        d = np.array([t[6] for t in tasks], np.int32)
        out = np.zeros((n, 8), np.int32)
        self._check(self.lib.mtr_test_wrap_dp(self.h, n, rd.ctypes.data, qs.ctypes.data, qe.ctypes.data, units.ctypes.data, uo.ctypes.data,
                                              g.ctypes.data, m.ctypes.data, d.ctypes.data, out.ctypes.data), "mtr_test_wrap_dp")
        return out

    def set_trace(self, max_events: int):
        self._check(self.lib.mtr_set_trace(self.h, max_events), "mtr_set_trace")

    def get_trace(self) -> np.ndarray:
        ev = C.POINTER(C.c_int32)()
        n = C.c_int64()
        self._check(self.lib.mtr_get_trace(self.h, C.byref(ev), C.byref(n)), "mtr_get_trace")
        arr = np.ctypeslib.as_array(ev, shape=(max(n.value, 1), 16))[: n.value].copy()
        _libc.free(C.cast(ev, C.c_void_p))
        return arr


def pack_read(codes: np.ndarray) -> np.ndarray:
    """one read in the device layout of include/mtr_hip.h ("the host's own packing"): len//16 + 4 words, first base in the
    top bits of word 0, zero behind the read"""
    n = len(codes)
    nw = n // 16 + 4
    padded = np.zeros(nw * 16, np.uint32)
    padded[:n] = codes
    shifts = (30 - 2 * np.arange(16, dtype=np.uint32)).astype(np.uint32)
    return np.bitwise_or.reduce(padded.reshape(nw, 16) << shifts, axis=1).astype(np.uint32)


def codes_from_str(s: str) -> np.ndarray:
    """ACGT/acgt -> 0..3 (reference handle_one_file.c:169-188); anything else raises like the reference exits."""
    lut = np.full(256, 255, np.uint8)
    for ch, v in zip("ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ord(ch)] = v
    a = lut[np.frombuffer(s.encode(), np.uint8)]
    if (a == 255).any():
        bad = s[int(np.argmax(a == 255))]
        raise ValueError(f"Invalid character: {bad}")
    return a
