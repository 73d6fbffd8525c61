"""The C-ABI library on CPU: it builds for gfx950, loads, exports every symbol include/mtr_hip.h declares, keeps the
record layout of the header, and refuses to run without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import mtr_amd
from mtr_amd import build as mbuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    mbuild.build()
    return mtr_amd.load_library()


def test_exports_every_declared_symbol(lib):
    declared = set()
    for h in ("mtr_hip.h", "mtr_hip_test.h"):
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                       # comments mention functions too
        inline = set(re.findall(r"static\s+inline\s+[a-z_0-9]+\s+(mtr_[a-z_0-9]+)\s*\(", hdr))
        declared |= set(re.findall(r"^[a-z_0-9 \*]*?\b(mtr_[a-z_0-9]+)\s*\(", hdr, flags=re.M)) - inline
    assert declared >= set(mtr_amd.EXPORTS), set(mtr_amd.EXPORTS) - declared
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libmtr_hip.so"
    assert lib.mtr_abi_version() == 5


def test_record_layout_matches_header():
    assert C.sizeof(mtr_amd.CRecord) == 14 * 4 + 504 + 500 * 4 == 2560
    assert mtr_amd.CRecord.unit.offset == 56 and mtr_amd.CRecord.unit_score.offset == 560


def test_library_carries_gfx950_code_object():
    data = open(mtr_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"mtr_k_reads" in data and b"mtr_k1_ranges" in data


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.mtr_create(0, 1, C.c_float(0.6), C.byref(h)) == 1          # MTR_ERR_NO_DEVICE
    assert not h.value
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_NO_DEVICE"):
        mtr_amd.Engine()


def test_product_does_not_reference_the_oracle():
    """the product path may not import, link or call anything under oracle/ (it is the checker, not the product)"""
    for base, _, files in os.walk(os.path.join(ROOT, "mtr_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".inc", "Makefile")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "oracle/" not in txt and "mtr_oracle" not in txt and "liboracle" not in txt, os.path.join(base, f)


def test_codes_from_str():
    assert mtr_amd.codes_from_str("ACGTacgt").tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    with pytest.raises(ValueError, match="Invalid character: N"):
        mtr_amd.codes_from_str("ACNT")


def test_synthetic_generator_is_seeded_and_shaped():
    from mtr_amd import synth
    a = synth.make_reads("c2", 5, 9)
    b = synth.make_reads("c2", 5, 9)
    assert all(np.array_equal(x[1], y[1]) for x, y in zip(a, b))
    lens = [len(c) for _, c in synth.make_reads("headline2k", 50, 1)]
    assert 1950 < np.mean(lens) < 2150                                 # unit 100 x 10 (+ins -del) + 2 x 500
    mixed = [len(c) for _, c in synth.make_reads("c4", 50, 1)]
    assert 1800 < np.mean(mixed) < 2400


def test_wire_form_round_trip_on_the_host(lib):
    """mtr_pack_records / mtr_unpack_records need no device: mtr_record <-> wire form (56-byte header, rep_period unit bytes
    padded to 4, rep_period int32 scores), the layout tests/host_util.wire_record writes for the replay tables."""
    from tests import golden_util as gu
    from tests import host_util as hu
    cap = gu.capture_by_read("3_5", "default") + gu.capture_by_read("synth_c2", "default")
    tuples = [gu.g4_tuple(ev) for per_read in cap for ev in per_read["G4"]]
    assert len(tuples) > 30
    recs = (mtr_amd.CRecord * len(tuples))()
    for r, t in zip(recs, tuples):
        (r.rep_start, r.rep_end, r.repeat_len, r.rep_period, r.num_freq_unit, r.num_matches, r.num_mismatches, r.num_insertions,
         r.num_deletions, r.kmer, r.match_gain, r.mismatch_penalty, r.indel_penalty) = t[:13]
        r.unit = t[13].encode()
        for i, v in enumerate(t[14][: t[3]]):
            r.unit_score[i] = v
    want = b"".join(hu.wire_record(t) for t in tuples)
    buf = C.create_string_buffer(len(want) + 16)
    n = lib.mtr_pack_records(recs, len(tuples), buf, len(want) + 16)
    assert n == len(want) and buf.raw[:n] == want
    assert lib.mtr_pack_records(recs, len(tuples), buf, len(want) - 1) == -1                     # too small
    back = (mtr_amd.CRecord * len(tuples))()
    assert lib.mtr_unpack_records(buf, n, len(tuples), back) == 0
    for b, t in zip(back, tuples):
        assert (b.rep_start, b.rep_end, b.repeat_len, b.rep_period, b.num_matches, b.kmer, b.indel_penalty) == (t[0], t[1], t[2], t[3], t[5], t[9], t[12])
        assert b.unit.decode() == t[13] and tuple(b.unit_score[: t[3]]) == tuple(t[14][: t[3]])
    assert lib.mtr_unpack_records(buf, n - 4, len(tuples), back) == 2                            # truncated blob: MTR_ERR_BAD_ARG
    assert lib.mtr_unpack_records(buf, n, len(tuples) - 1, back) == 2                            # count does not cover the blob


def test_packed_image_of_a_read():
    """mtr_pack_read (inline of the header) and the Python mirror agree on the device image: len // 16 + 4 words, first base
    in the top bits, zero behind the read"""
    codes = np.array([0, 1, 2, 3] * 9 + [3, 3], np.uint8)
    w = mtr_amd.pack_read(codes)
    assert len(w) == len(codes) // 16 + 4 and w.dtype == np.uint32
    for i, c in enumerate(codes):
        assert (int(w[i >> 4]) >> (30 - 2 * (i & 15))) & 3 == c
    assert int(w[2]) & ((1 << (32 - 2 * (len(codes) & 15))) - 1) == 0 and not w[3:].any()
