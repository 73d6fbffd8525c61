"""ctypes binding of oracle/liboracle_mtr.so — TEST INFRASTRUCTURE ONLY (the CPU oracle).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker; never by mtr_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle_mtr.so")
MAX_PERIOD = 500


class ORecord(C.Structure):
    _fields_ = [("rep_start", C.c_int32), ("rep_end", C.c_int32), ("repeat_len", C.c_int32), ("rep_period", C.c_int32),
                ("num_freq_unit", C.c_int32), ("num_matches", C.c_int32), ("num_mismatches", C.c_int32),
                ("num_insertions", C.c_int32), ("num_deletions", C.c_int32), ("kmer", C.c_int32), ("match_gain", C.c_int32),
                ("mismatch_penalty", C.c_int32), ("indel_penalty", C.c_int32),
                ("unit", C.c_char * (MAX_PERIOD * 2 + 4)), ("unit_score", C.c_int32 * MAX_PERIOD)]


class OStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("dp_calls", "dp_cells", "dp_rows", "dp_max_cells", "revise_dp_calls", "revise_dp_cells",
                                         "kmer_tables", "kmer_lookups", "searches_passing_maxfreq", "ranges_candidate",
                                         "ranges_executed", "records", "di_passes", "di_positions")]


class ODpResult(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rep_start", "rep_end", "repeat_len", "num_freq_unit", "num_matches",
                                         "num_mismatches", "num_insertions", "num_deletions")]


_lib = None
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def load(build: bool = True):
    global _lib
    if _lib is not None:
        return _lib
    if build:
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    lib = C.CDLL(LIB)
    lib.mtro_create.argtypes = [C.c_int, C.c_float]
    lib.mtro_create.restype = C.c_void_p
    lib.mtro_destroy.argtypes = [C.c_void_p]
    lib.mtro_set_file_order.argtypes = [C.c_void_p, C.c_int]
    lib.mtro_set_file_order.restype = None
    lib.mtro_process_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.POINTER(ORecord))]
    lib.mtro_process_read.restype = C.c_int
    lib.mtro_get_stats.argtypes = [C.c_void_p]
    lib.mtro_get_stats.restype = C.POINTER(OStats)
    lib.mtro_reset_stats.argtypes = [C.c_void_p]
    lib.mtro_ranges.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mtro_ranges.restype = C.c_int
    lib.mtro_wrap_dp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ODpResult)]
    lib.mtro_wrap_dp.restype = C.c_int
    lib.mtro_mt_bases.argtypes = [C.c_void_p, C.c_int]
    lib.mtro_chain.argtypes = [C.POINTER(ORecord), C.c_int, C.c_void_p]
    lib.mtro_chain.restype = C.c_int
    _lib = lib
    return lib


class Oracle:
    def __init__(self, manhattan: bool = True, min_match_ratio: float = 0.6):
        self.lib = load()
        self.h = C.c_void_p(self.lib.mtro_create(1 if manhattan else 0, C.c_float(min_match_ratio)))

    def set_file_order(self, on: bool = True):
        """the reference's behaviour on a multi-read file: process() must then be called in file order"""
        self.lib.mtro_set_file_order(self.h, 1 if on else 0)

    def close(self):
        if self.h:
            self.lib.mtro_destroy(self.h)
            self.h = None

    def process(self, codes: np.ndarray):
        """-> list of 15-tuples in the field order of mtr_amd.Record"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        out = C.POINTER(ORecord)()
        n = self.lib.mtro_process_read(self.h, b"", codes.ctypes.data, len(codes), C.byref(out))
        if n < 0:
            raise RuntimeError("oracle failed")
        res = []
        for i in range(n):
            r = out[i]
            per = r.rep_period
            res.append((r.rep_start, r.rep_end, r.repeat_len, per, r.num_freq_unit, r.num_matches, r.num_mismatches,
                        r.num_insertions, r.num_deletions, r.kmer, r.match_gain, r.mismatch_penalty, r.indel_penalty,
                        r.unit.decode(), tuple(r.unit_score[:max(0, min(per, MAX_PERIOD))])))
        if n > 0:
            _libc.free(C.cast(out, C.c_void_p))
        return res

    def stats(self):
        s = self.lib.mtro_get_stats(self.h).contents
        return {n: int(getattr(s, n)) for n, _ in OStats._fields_}

    def ranges(self, codes: np.ndarray):
        """-> list of (start, end, w, di_bits) of the usable ranges after de-duplication"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        L = len(codes)
        di = np.zeros(L, np.float64)
        end = np.zeros(L, np.int32)
        w = np.zeros(L, np.int32)
        self.lib.mtro_ranges(self.h, codes.ctypes.data, L, di.ctypes.data, end.ctypes.data, w.ctypes.data)
        bits = di.view(np.uint64)
        idx = np.nonzero((end > -1) & (end < L) & (di != -1.0))[0]
        return [(int(i), int(end[i]), int(w[i]), int(bits[i])) for i in idx]

    def wrap_dp(self, codes: np.ndarray, qs: int, qe: int, unit: np.ndarray, G: int, MM: int, D: int):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        unit = np.ascontiguousarray(unit, dtype=np.uint8)
        r = ODpResult()
        rc = self.lib.mtro_wrap_dp(codes.ctypes.data, len(codes), qs, qe, unit.ctypes.data, len(unit), G, MM, D, C.byref(r))
        if rc != 0:
            raise RuntimeError("oracle DP too large")
        return (r.rep_start, r.rep_end, r.repeat_len, r.num_freq_unit, r.num_matches, r.num_mismatches, r.num_insertions, r.num_deletions)


def mt_bases(n: int) -> np.ndarray:
    out = np.zeros(n, np.uint8)
    load().mtro_mt_bases(out.ctypes.data, n)
    return out
