"""The built library's kernels against the resources their launches assume (CPU; reads the gfx950 code object out of mtr_amd/libmtr_hip.so).

Round 5 lost a wavefront per CU in the revision kernel for half a round (a 512-byte __shared__ array took the workgroup over a sixteenth of the CU's LDS) and, later,
put the walk kernels' whole context struct into scratch memory with one dynamically indexed member - nothing fails when that happens, the kernels are just slower.
This test fails."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mtr_amd", "libmtr_hip.so")
READELF = next((p for p in ("/opt/rocm/lib/llvm/bin/llvm-readelf", shutil.which("llvm-readelf")) if p and os.path.exists(p)), None)

LDS_PER_CU = 160 * 1024


def _code_object(tmp_path):
    blob = open(LIB, "rb").read()
    at = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert at >= 0, "no offload bundle in libmtr_hip.so"
    n = struct.unpack_from("<Q", blob, at + 24)[0]
    p = at + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", blob, p)
        p += 24
        triple = blob[p:p + tl].decode()
        p += tl
        if "gfx950" in triple:
            out = tmp_path / "mtr_gfx950.co"
            out.write_bytes(blob[at + off:at + off + size])
            return str(out)
    raise AssertionError("no gfx950 code object in libmtr_hip.so")


def _kernels(tmp_path):
    txt = subprocess.run([READELF, "--notes", _code_object(tmp_path)], capture_output=True, text=True, check=True).stdout
    out, cur = {}, {}
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip("'\"")
        if key in ("group_segment_fixed_size", "private_segment_fixed_size", "vgpr_count", "sgpr_count", "vgpr_spill_count", "name"):
            if key in cur:                 # the next kernel's block begins
                if "name" in cur:
                    out[cur["name"]] = cur
                cur = {}
            cur[key] = val if key == "name" else int(val)
    if "name" in cur:
        out[cur["name"]] = cur
    return out


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if READELF is None:
        pytest.skip("llvm-readelf not found")
    if not os.path.exists(LIB):
        import mtr_amd.build
        mtr_amd.build.build()
    ks = _kernels(tmp_path_factory.mktemp("co"))
    assert len(ks) >= 25, sorted(ks)
    return ks


def _find(kernels, stem):
    hits = [v for k, v in kernels.items() if k.startswith(f"_Z{len(stem)}{stem}")]        # (the mangled name carries the length: mtr_k_walks is not mtr_k_walks_k)
    assert len(hits) >= 1, (stem, sorted(kernels))
    return hits


@pytest.mark.parametrize("stem", ["mtr_k1_ranges", "mtr_k_walks", "mtr_k_walks_k", "mtr_k_select", "mtr_k_polish", "mtr_k_revise", "mtr_k_revise_quads", "mtr_k_reads"])
def test_sixteen_wavefronts_per_cu_by_lds(kernels, stem):
    """one 64-lane workgroup per wavefront: 16 of them per CU need at most a sixteenth of its LDS each"""
    for k in _find(kernels, stem):
        assert k["group_segment_fixed_size"] <= LDS_PER_CU // 16, k


@pytest.mark.parametrize("stem", ["mtr_k1_ranges", "mtr_k_walks", "mtr_k_walks_k", "mtr_k_select", "mtr_k_polish", "mtr_k_revise_quads", "mtr_k_dp2_quads", "mtr_k_revise", "mtr_k_reads"])
def test_four_wavefronts_per_simd_by_registers(kernels, stem):
    for k in _find(kernels, stem):
        assert k["vgpr_count"] <= 128, k


@pytest.mark.parametrize("stem", ["mtr_k_walks", "mtr_k_walks_k", "mtr_k_select", "mtr_k_polish", "mtr_k_gather", "mtr_k_finish", "mtr_k_replay"])
def test_no_scratch_memory_in_the_kernels_that_align_nothing(kernels, stem):
    """no spills and no struct or array kept in scratch (a dynamically indexed member of the context struct did that to the walk kernels)"""
    for k in _find(kernels, stem):
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, k


def test_dp_kernels_spill_little(kernels):
    for stem, limit in (("mtr_k_dp2_quads", 40), ("mtr_k_revise_quads", 48)):
        for k in _find(kernels, stem):
            assert k["vgpr_spill_count"] <= limit, k
