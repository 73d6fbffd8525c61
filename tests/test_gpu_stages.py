"""Stage-level parity of the unit phase on the GPU (-m gpu): what the kernels compute INSIDE a range, against capture
points of the unmodified reference (tests/golden), so that a regression in the walks or the revision is localised and
does not only show up as "record differs":
  G2   search_De_Bruijn_graph (consensus.c:507-582): per (window, k) the found flag and the unit chosen among the two walk
       directions with its alignment  <-> trace event 2;
  G3p  polish_repeat (consensus.c:610-704): unit in / unit out                                  <-> trace event 4;
  G3r  revise_representative_unit_sub (consensus.c:851-1046): unit in, scores, revised unit     <-> trace event 5.
Units are compared through their length and an FNV-1a checksum of their base codes (what the trace carries).
Plus the builder-side sweeps promoted into the suite: BASELINE config 2 at 1 000 reads and the config-3 shape at 16 reads,
every read against the oracle; -a on a config-3-shaped read through the command line; the accuracy table of
test_single_TR/test.sh on 200 reads per unit length."""
import gzip
import json
import os
import subprocess
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

import mtr_amd
from mtr_amd import synth
from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LUT = {"A": 0, "C": 1, "G": 2, "T": 3}


def fnv(unit: str) -> int:
    h = 2166136261
    for ch in unit:
        h = ((h ^ LUT[ch]) * 16777619) & 0xFFFFFFFF
    return h & 0x7FFFFFFF


def traced_run(name, mask):
    os.environ["MTR_STAGED"] = "0"                         # the per-read kernel: the reference's own order of ranges (the sequential loop)
    os.environ["MTR_TRACE_MASK"] = str(mask)
    try:
        e = mtr_amd.Engine()
        reads = gu.read_fasta(gu.input_path(name))
        e.set_trace(2_000_000)
        e.upload([c for _, c in reads])
        e.run()
        ev = e.get_trace()
        e.close()
    finally:
        del os.environ["MTR_STAGED"], os.environ["MTR_TRACE_MASK"]
    assert len(ev) < 2_000_000
    return ev


@pytest.mark.parametrize("name", ["3_5", "synth_2k"])
def test_search_stage_matches_reference_G2(name):
    ev = traced_run(name, 1 << 2)
    got = {}
    for e in ev[ev[:, 0] == 2]:
        rd, qs, qe, k, found = int(e[1]), int(e[2]), int(e[3]), int(e[4]), int(e[5])
        got.setdefault((rd, qs, qe, k), []).append((found, tuple(int(x) for x in e[7:16])))
    n = 0
    with gzip.open(os.path.join(gu.GOLDEN, f"{name}.default.l2.jsonl.gz"), "rt") as fh:
        for line in fh:
            g = json.loads(line)
            key = (g["rd"], g["qs"], g["qe"], g["k"])
            assert key in got, f"the GPU never searched {key}"
            founds = [f for f, _ in got[key]]
            assert g["found"] in founds, (key, g["found"], founds)
            if g["found"]:
                want = (g["period"], g["rep_start"], g["rep_end"], g["repeat_len"], g["copies"], g["mat"], g["mis"], g["ins"], g["del"])
                assert any(f == 1 and v == want for f, v in got[key]), (key, want, got[key])
                n += 1
    assert n > 20


@pytest.mark.parametrize("name", ["synth_2k", "synth_c2", "10_50", "20_50"])
def test_polish_and_revision_stages_match_reference_G3p_G3r(name):
    ev = traced_run(name, (1 << 4) | (1 << 5))
    cap = gu.capture_by_read(name, "default")
    n_p = n_r = 0
    for rd, per_read in enumerate(cap):
        mine = ev[ev[:, 1] == rd]
        # polish: one call per revision, in the reference's order
        got_p = [(int(e[2]), int(e[3]), int(e[4]), int(e[5]), int(e[6]), int(e[7])) for e in mine[mine[:, 0] == 4]]
        want_p = [(g["rep_start"], g["rep_end"], g["k"], len(g["in"]), len(g["out"]), fnv(g["out"])) for g in per_read["G3p"]]
        assert got_p == want_p, f"{name} read {rd}: polish_repeat calls differ"
        n_p += len(want_p)
        # revision rounds: the kernel answers a round it has already run from its memo, so it runs a subset of the
        # reference's calls - every one it runs must be one of the reference's, and every distinct one of the reference's
        # must have been run
        got_r = {(int(e[2]), int(e[3]), int(e[4]), int(e[5]), int(e[6]), int(e[7]), int(e[8]), int(e[9])) for e in mine[mine[:, 0] == 5]}
        want_r = {(g["rep_start"], g["rep_end"], len(g["in"]), g["G"], g["MM"], g["D"], g["out_period"],
                   fnv(g["out"]) if 0 < g["out_period"] < 1024 else 0) for g in per_read["G3r"]}
        assert got_r == want_r, f"{name} read {rd}: revise_representative_unit_sub results differ: {sorted(got_r ^ want_r)[:4]}"
        n_r += len(want_r)
    assert n_p > 0 and n_r > 0


def _oracle_chunk(args):
    reads, manhattan = args
    from tests.oracle_binding import Oracle
    o = Oracle(manhattan=manhattan)
    out = [o.process(c) for c in reads]
    o.close()
    return out


def _against_oracle(reads, manhattan=True, workers=8):
    chunks = [reads[i::workers] for i in range(workers)]
    with ProcessPoolExecutor(workers) as ex:
        parts = list(ex.map(_oracle_chunk, [(c, manhattan) for c in chunks]))
    want = [None] * len(reads)
    for w, part in enumerate(parts):
        want[w::workers] = part
    e = mtr_amd.Engine(manhattan=manhattan)
    got = e.process(reads)
    e.close()
    return [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]


@pytest.mark.timeout(900)
def test_config2_at_full_size_every_read_against_the_oracle():
    """BASELINE config 2: 1 000 reads, unit 100 x 10 copies, L ~ 1.25 kb"""
    reads = [c for _, c in synth.make_reads("c2")]
    assert len(reads) == 1000
    assert _against_oracle(reads) == []


@pytest.mark.timeout(900)
def test_config3_shape_16_reads_against_the_oracle():
    """BASELINE config 3 shape: unit 200 x 200 copies, L ~ 42 kb (DPs of 4 M cells)"""
    reads = [c for _, c in synth.make_reads("c3", 16, 3)]
    assert _against_oracle(reads) == []


@pytest.mark.timeout(900)
def test_config3_at_full_size_record_stream_is_the_oracles():
    """BASELINE config 3 as bench.py measures it (its default line's secondary.c3 runs the same child): 100 reads of unit 200 x 200 copies,
    the launch's record stream in wire form against the CPU oracle's known answer (tests/golden/c3_100_wire.json)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("MTR_LIB", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c3", "--steps", "1", "--warmup", "1", "--no-cli", "--no-latency", "--cpu-sample", "0"],
                       capture_output=True, env=env, timeout=800, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-800:]
    line = json.loads(p.stdout.decode().strip().splitlines()[-1])
    assert line["matches_oracle"] is True and line["record_stream"]["records"] == 1724, line.get("record_stream")
    assert "config 3" in line["metric"] and line["config"]["reads_per_gpu"] == 100


@pytest.mark.timeout(900)
def test_cli_alignments_on_a_config3_shaped_read(tmp_path):
    from tests.oracle_binding import ORACLE_DIR
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "mtr_amd", "host"), "mTR"], check=True)
    fa = tmp_path / "c3.fa"
    synth.write_fasta(str(fa), synth.make_reads("c3", 2, 7))
    want = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), "-a", str(fa)], capture_output=True, check=True).stdout
    p = subprocess.run([os.path.join(ROOT, "mtr_amd", "host", "mTR"), "-a", str(fa)], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[:500]
    assert p.stdout == want and len(want) > 100000


@pytest.mark.timeout(900)
def test_accuracy_table_equals_the_oracles(tmp_path):
    """test_single_TR/test.sh:35-63 with count_match / comp_mTR_DP (tools/accuracy.py, tools/unit_score.c): unit lengths
    2..200 x 10 copies, 200 seeded reads each.  The GPU driver's report must score exactly as the oracle CLI's report on the
    same files (it is the same report), and the predictions must be good: the table is an accuracy harness, not only a diff."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import accuracy
    from tests.oracle_binding import ORACLE_DIR
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "mtr_amd", "host"), "mTR"], check=True)
    lib = accuracy.load_scorer()
    for u in (2, 5, 10, 20, 50, 100, 200):
        rng = np.random.RandomState(1000 + u)
        reads, truth = [], []
        for i in range(200):
            codes, unit = synth.make_read(rng, u, 10, u * 10, u * 10)
            reads.append((str(i), codes))
            truth.append("".join("ACGT"[int(x)] for x in unit))
        fa = tmp_path / f"u{u}.fa"
        synth.write_fasta(str(fa), reads)
        gpu = subprocess.run([os.path.join(ROOT, "mtr_amd", "host", "mTR"), str(fa)], capture_output=True, check=True).stdout.decode()
        orc = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), str(fa)], capture_output=True, check=True).stdout.decode()
        sg, so = accuracy.score(lib, gpu, truth), accuracy.score(lib, orc, truth)
        assert sg == so and gpu == orc, u
        assert sg["report_lines"] >= 150, (u, sg)
        assert sg["ratio>=0.94"] >= 0.8 * len(reads), (u, sg)               # the reference's own level on this error profile (90-100 %)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("profile", ["sub_heavy", "sub_del"])
def test_the_references_other_error_profiles_against_the_oracle(profile):
    """test_single_TR/test.sh:12-18 carries two alternates to the Nanopore profile (substitution 12.7 / insertion 3.2 / deletion 4.7 %
    and 9.7 / 2.9 / 7.5 %): substitution-heavy reads give a different mix of DPs, revisions and memo hits.  300 reads of each shape
    (the headline's unit 100 x 10 and mixed units 50-200), every read against the oracle, in the chain with and without the
    four-per-wavefront passes."""
    prof = synth.PROFILES[profile]
    reads = [c for _, c in synth.make_reads("headline2k", 150, 41, prof)] + [c for _, c in synth.make_reads("c4", 150, 42, prof)]
    assert _against_oracle(reads) == []
    os.environ["MTR_QUAD_MIN"] = "1"
    try:
        assert _against_oracle(reads) == []
    finally:
        del os.environ["MTR_QUAD_MIN"]
