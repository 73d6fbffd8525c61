"""Parity of the HIP path (through the C-ABI of include/mtr_hip.h) — run on the MI355X box with -m gpu.

Three kinds of evidence:
  1. against the golden vectors recorded from the unmodified reference (tests/golden): candidate ranges
     (G1), wrap-around DP calls (G3) and the inserted records (G4 = the 17 arguments of
     insert_an_alignment_into_set), in Manhattan and Pearson (-p) mode — bit-exact;
  2. against the CPU oracle (oracle/mtr_oracle.c, itself pinned to the same vectors) on seeded synthetic
     reads at sizes the oracle finishes in seconds — bit-exact;
  3. at BASELINE.json's full size (10 000 reads of ~2 kb) through size-independent properties: batch
     independence (a read's records do not depend on which reads share its batch, nor on their order),
     run-to-run determinism, and a random sample against the oracle.
"""
import os
import subprocess

import numpy as np
import pytest

import mtr_amd
from mtr_amd import synth
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = mtr_amd.Engine()
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_p():
    e = mtr_amd.Engine(manhattan=False)
    yield e
    e.close()


@pytest.fixture(scope="module")
def oracle():
    from tests.oracle_binding import Oracle
    return Oracle()


def _diff_msg(i, want, got):
    msg = [f"read {i}: {len(want)} records expected, {len(got)} produced"]
    for a, b in zip(want, got):
        if a != b:
            for k, (x, y) in enumerate(zip(a, b)):
                if x != y:
                    msg.append(f"  field {k}: want {str(x)[:120]} got {str(y)[:120]}")
            break
    return "\n".join(msg)


# ---- 1. golden vectors of the reference ----------------------------------------------------------------------
@pytest.mark.parametrize("name,mode", gu.cases())
def test_records_match_reference_golden(eng, eng_p, name, mode):
    e = eng if mode == "default" else eng_p
    reads = gu.read_fasta(gu.input_path(name))
    cap = gu.capture_by_read(name, mode)
    assert len(cap) == len(reads)
    got = e.process([c for _, c in reads])
    for i, (per_read, g) in enumerate(zip(cap, got)):
        want = [gu.g4_tuple(ev) for ev in per_read["G4"]]
        assert [tuple(r) for r in g] == want, _diff_msg(i, want, [tuple(r) for r in g])


@pytest.mark.parametrize("name,mode", gu.cases())
def test_ranges_match_reference_golden(eng, eng_p, name, mode):
    e = eng if mode == "default" else eng_p
    reads = gu.read_fasta(gu.input_path(name))
    cap = gu.capture_by_read(name, mode)
    e.upload([c for _, c in reads])
    got = e.test_ranges()
    for i, per_read in enumerate(cap):
        assert got[i] == gu.g1_usable(per_read["G1"]), f"read {i} of {name} [{mode}]"


def test_dp_calls_match_reference_golden(eng):
    """Every wrap_around_DP_sub call the reference made on the fixture reads (G3), replayed on the DP kernel."""
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    total = 0
    for name in ("synth_c2", "synth_2k", "synth_c4", "edge", "3_5", "2_5_10_20_set", "10_50", "20_50"):
        reads = gu.read_fasta(gu.input_path(name))
        cap = gu.capture_by_read(name, "default")
        eng.upload([c for _, c in reads])
        tasks, want = [], []
        for i, per_read in enumerate(cap):
            for ev in per_read["G3"]:
                if len(ev["unit"]) == 0:
                    continue
                tasks.append((i, ev["qs"], min(ev["qe"], len(reads[i][1]) - 1), np.array([lut[ch] for ch in ev["unit"]], np.uint8), ev["G"], ev["MM"], ev["D"]))
                want.append((tuple(ev["out"]), ev["qe"]))
        # the revision DP may run one base past the read; the test entry point takes in-read windows only
        keep = [k for k, (_, qe) in enumerate(want) if qe == tasks[k][2]]
        tasks = [tasks[k] for k in keep]
        want = [want[k][0] for k in keep]
        out = eng.test_wrap_dp(tasks)
        bad = [k for k in range(len(tasks)) if tuple(int(x) for x in out[k]) != want[k]]
        assert not bad, f"{name}: {len(bad)} of {len(tasks)} DP calls differ, first: task {tasks[bad[0]][:3]} want {want[bad[0]]} got {out[bad[0]]}"
        total += len(tasks)
    assert total > 5000


# ---- 2. the oracle on seeded synthetic reads -------------------------------------------------------------------
@pytest.mark.parametrize("config,n,seed,manhattan", [("c2", 300, 11, True), ("headline2k", 150, 12, True), ("c4", 300, 14, True),
                                                      ("c2", 150, 21, False), ("c4", 150, 24, False)])
def test_records_match_oracle(eng, eng_p, config, n, seed, manhattan):
    from tests.oracle_binding import Oracle
    orc = Oracle(manhattan=manhattan)
    reads = [c for _, c in synth.make_reads(config, n, seed)]
    got = (eng if manhattan else eng_p).process(reads)
    bad = []
    for i, codes in enumerate(reads):
        want = orc.process(codes)
        if [tuple(r) for r in got[i]] != want:
            bad.append(_diff_msg(i, want, [tuple(r) for r in got[i]]))
    orc.close()
    assert not bad, f"{len(bad)} of {n} reads differ\n" + "\n".join(bad[:3])


@pytest.mark.parametrize("ratio", [0.0, 0.8, 1.0])
def test_min_match_ratio_option(ratio):
    """-m (main.c:66-73, min_match_ratio of handle_one_read.c:139 / consensus.c:563): 0 lets every candidate through the ratio gate, 1 only
    perfect alignments - exact repeats are in the set so that -m 1 reports something"""
    from tests.oracle_binding import Oracle
    orc = Oracle(min_match_ratio=ratio)
    e = mtr_amd.Engine(min_match_ratio=ratio)
    rng = np.random.RandomState(31)
    reads = [c for _, c in synth.make_reads("c2", 40, 31)]
    reads += [synth.make_read(rng, u, c, 60, 60, profile=(0, 0, 0))[0] for u, c in ((3, 40), (7, 30), (24, 12), (100, 8))]
    got = e.process(reads)
    n_rec = 0
    for i, codes in enumerate(reads):
        want = orc.process(codes)
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])
        n_rec += len(want)
    assert n_rec > 0
    e.close()
    orc.close()


def test_long_read_config3_shape(eng, oracle):
    """unit 200 x 200 copies (L ~ 42 kb): multi-chunk DP rows, DPs of millions of cells, large windows -> global tables."""
    reads = [c for _, c in synth.make_reads("c3", 1, 3)]
    got = eng.process(reads)
    want = oracle.process(reads[0])
    assert [tuple(r) for r in got[0]] == want, _diff_msg(0, want, [tuple(r) for r in got[0]])
    assert eng.counters()["global_tables"] > 0


def test_mixed_length_batch(eng, oracle):
    """very short and long reads in one batch: scratch is sized by the longest, results by each read alone"""
    rng = np.random.RandomState(5)
    reads = [c for _, c in synth.make_reads("c2", 6, 41)]
    reads += [rng.randint(0, 4, size=n).astype(np.uint8) for n in (1, 2, 9, 10, 11, 12, 31, 100, 999, 1000, 1001)]
    reads += [np.tile(np.array([0, 1], np.uint8), 300), np.zeros(500, np.uint8), np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 1500)]
    got = eng.process(reads)
    for i, codes in enumerate(reads):
        want = oracle.process(codes)
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])


# ---- 3. full size: properties -----------------------------------------------------------------------------------
def test_full_size_batch_independence_and_sample(eng, oracle):
    reads = [c for _, c in synth.make_reads("headline2k", 10000, 2)]
    eng.upload(reads)
    eng.run()
    full = eng.fetch()
    cnt = eng.counters()
    assert cnt["records"] == sum(len(r) for r in full)
    # the DEFAULT selection for a batch of 20 M bases (no MTR_* override in this test): the chain in two passes with every wide range first
    # (k3_staged.hip.inc: mtr_k_pass_mark).  The mark pass's list of ranges is not provably a superset of what the reference's loop reaches; a miss
    # is a silent per-read re-run, so it must show here: no read may have been sent back, and the chain searches what the loop reaches (+ a few %)
    assert not any(k in os.environ for k in ("MTR_TWO_PASS", "MTR_STAGED", "MTR_QUAD_MIN"))
    assert eng.last_mode() == "staged chain"
    assert cnt["reads_sent_back"] == 0, cnt["reads_sent_back"]
    assert cnt["ranges_searched"] < 1.05 * cnt["ranges_executed"], (cnt["ranges_searched"], cnt["ranges_executed"])
    kt = eng.kernel_times_ms()
    assert all(kt.get("chain_" + p, 0) > 0 for p in ("ranges", "unit_search", "alignments", "revisions")), kt
    # determinism: a second run of the same resident batch gives the same records
    eng.run()
    again = eng.fetch()
    assert again == full
    # batch independence: a sample processed alone, in reverse order, gives the same per-read records
    rng = np.random.RandomState(99)
    idx = rng.choice(len(reads), 96, replace=False)
    sub = eng.process([reads[i] for i in idx[::-1]])
    for k, i in enumerate(idx[::-1]):
        assert sub[k] == full[i], f"read {i} depends on its batch"
    # and the sample agrees with the oracle
    for i in idx[:48]:
        want = oracle.process(reads[i])
        assert [tuple(r) for r in full[i]] == want, _diff_msg(i, want, [tuple(r) for r in full[i]])
    # every record satisfies the reference's own acceptance gates (handle_one_read.c:139-146, :239-240)
    for recs in full:
        for r in recs:
            tot = r.num_matches + r.num_mismatches + r.num_insertions + r.num_deletions
            assert r.repeat_len > 0 and r.rep_start + 10 < r.rep_end
            assert np.float32(r.num_matches) / np.float32(tot) >= np.float32(0.6)
            assert r.num_freq_unit > 5 and 2 <= r.rep_period < 500 and len(r.unit) == r.rep_period
            assert r.repeat_len == r.num_matches + r.num_mismatches + r.num_insertions


# ---- error behaviour of the boundary -------------------------------------------------------------------------------
def test_bad_arguments(eng):
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_BAD_ARG"):
        eng.process([np.array([0, 1, 2, 4], np.uint8)])          # the reference exits on a non-ACGT character
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_BAD_ARG"):
        eng.process([np.zeros(0, np.uint8)])
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_BAD_ARG"):
        eng.process([])
    # the context stays usable after an error
    assert eng.process([np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 80)])[0]     # TTAGGG x 80


# ---- the 32-bit DP kernels (what reads longer than 64 kb fall back to), forced through a test knob ------------------
def test_32bit_dp_fallbacks(monkeypatch, oracle):
    """MTR_DP16_MAX_ROWS=0 sends every wrap-around DP through dp_forward / dp_forward2 (32-bit values, byte cells)
    instead of the packed 16-bit kernels: same records, and the golden DP calls of the reference again."""
    monkeypatch.setenv("MTR_DP16_MAX_ROWS", "0")
    e = mtr_amd.Engine()
    reads = [c for _, c in synth.make_reads("headline2k", 48, 77)] + [c for _, c in synth.make_reads("c4", 48, 78)]
    got = e.process(reads)
    for i, codes in enumerate(reads):
        want = oracle.process(codes)
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    name = "synth_2k"
    rd = gu.read_fasta(gu.input_path(name))
    cap = gu.capture_by_read(name, "default")
    e.upload([c for _, c in rd])
    tasks, want = [], []
    for i, per_read in enumerate(cap):
        for ev in per_read["G3"]:
            if len(ev["unit"]) == 0 or ev["qe"] > len(rd[i][1]) - 1:
                continue
            tasks.append((i, ev["qs"], ev["qe"], np.array([lut[ch] for ch in ev["unit"]], np.uint8), ev["G"], ev["MM"], ev["D"]))
            want.append(tuple(ev["out"]))
    out = e.test_wrap_dp(tasks)
    bad = [k for k in range(len(tasks)) if tuple(int(x) for x in out[k]) != want[k]]
    assert not bad, f"{len(bad)} of {len(tasks)} DP calls differ through the 32-bit kernels"
    e.close()


def test_very_long_read_with_several_repeats(eng, oracle):
    """~110 kb read with repeats of 7-, 150- and 300-base units: DP rows beyond the 16-bit kernels' limit for the large
    windows (32-bit two-parameter pass), units > 256 bases (8 chunks of columns), k-mer tables in global memory."""
    rng = np.random.RandomState(2024)
    parts = []
    for unit_len, copies in ((300, 30), (7, 500), (150, 100), (40, 60)):
        parts.append(rng.randint(0, 4, size=18000).astype(np.uint8))
        body, _ = synth.make_read(rng, unit_len, copies, 0, 0)
        parts.append(body)
    parts.append(rng.randint(0, 4, size=18000).astype(np.uint8))
    read = np.concatenate(parts)
    assert 100000 < len(read) < 140000
    got = eng.process([read])
    want = oracle.process(read)
    assert [tuple(r) for r in got[0]] == want, _diff_msg(0, want, [tuple(r) for r in got[0]])
    assert len(want) >= 3


def test_read_of_the_maximum_length(eng, oracle):
    """833 333 bases = the longest read whose padded buffer (L + 2r) still fits the reference's MAX_INPUT_LENGTH
    (mTR.h:31): 20 passes over 10^6 positions, 95 000 candidate ranges, 620 records; one more base is refused."""
    rng = np.random.RandomState(99)
    parts = []
    for unit_len, copies in ((180, 40), (3, 300), (60, 150)):
        parts.append(rng.randint(0, 4, size=270000).astype(np.uint8))
        parts.append(synth.make_read(rng, unit_len, copies, 0, 0)[0])
    read = np.concatenate(parts)
    read = np.concatenate([read, rng.randint(0, 4, size=833333 - len(read)).astype(np.uint8)])
    got = eng.process([read])
    want = oracle.process(read)
    assert [tuple(r) for r in got[0]] == want, _diff_msg(0, want, [tuple(r) for r in got[0]])
    assert len(want) > 500
    with pytest.raises(mtr_amd.MtrError):
        eng.process([np.zeros(833334, np.uint8)])


# ---- the kernel modes: same records as one wavefront per read -----------------------------------------------------------
MODES = {"per_read": {"MTR_STAGED": "0"},
         "staged": {"MTR_STAGED": "1", "MTR_QUAD_MIN": "0"}, "staged_quads": {"MTR_STAGED": "1", "MTR_QUAD_MIN": "1"},
         # the lists of a big batch in 64 sub-lists with a counter each (k3_staged.hip.inc), forced on a small one
         "staged_sublists": {"MTR_STAGED": "1", "MTR_QUAD_MIN": "0", "MTR_TEST_STAGED_CAPS": "nsub=64"},
         "staged_sublists_quads": {"MTR_STAGED": "1", "MTR_QUAD_MIN": "1", "MTR_TEST_STAGED_CAPS": "nsub=64"},
         # the chain in TWO passes (the ranges of wide windows first, then what their records leave; the default for deep batches of long reads),
         # and with every wide range in the first pass
         "staged_two_pass": {"MTR_STAGED": "1", "MTR_TWO_PASS": "1"}, "staged_two_pass_quads": {"MTR_STAGED": "1", "MTR_TWO_PASS": "1", "MTR_QUAD_MIN": "1"},
         "staged_two_pass_all_wide_first": {"MTR_STAGED": "1", "MTR_TWO_PASS": "2", "MTR_TEST_STAGED_CAPS": "nsub=64"},
         # two passes whose mark pass lists nothing: the replay reaches ranges nobody searched and sends those reads to the per-read kernel
         "staged_sent_back": {"MTR_STAGED": "1", "MTR_TWO_PASS": "1", "MTR_TEST_STAGED_FLAGS": "1"},
         "staged_overflow_sublist": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "nsub=64,kc=640"},       # 10 blocks per sub-list
         "staged_overflow_arena": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "arena=200000"},
         "staged_overflow_kc": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "kc=40"},
         "staged_overflow_cont": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "cont=10"},
         "staged_overflow_dp": {"MTR_STAGED": "1", "MTR_QUAD_MIN": "1", "MTR_TEST_STAGED_CAPS": "dp=50"},
         "staged_overflow_rev": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "rev=3"},
         "staged_overflow_cand": {"MTR_STAGED": "1", "MTR_TEST_STAGED_CAPS": "cand=5"}}


@pytest.mark.parametrize("mode", sorted(MODES))
def test_kernel_modes_match_the_oracle(monkeypatch, oracle, mode):
    """per_read: one wavefront per read (the reference's sequential range loop; the fallback).  staged (the default arrangement):
    ranges -> walks -> every two-parameter DP its own work item
    (one wavefront each; staged_quads: units of 17..128 bases four per wavefront whatever the batch size) -> selection/revision
    -> replay; staged_overflow_*: one of the chain's capacities far too small, so the library must fall back to the per-read kernel
    (the kernels behind the overflow must not walk the half-written work lists).  All must give the oracle's records."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    e = mtr_amd.Engine()
    rng = np.random.RandomState(7)
    reads = [c for _, c in synth.make_reads("headline2k", 40, 91)] + [c for _, c in synth.make_reads("c4", 40, 92)]
    reads += [rng.randint(0, 4, size=n).astype(np.uint8) for n in (1, 9, 31, 1000)]
    reads += [np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 700), c3_read := synth.make_reads("c3", 1, 5)[0][1]]
    got = e.process(reads)
    if mode.startswith("staged"):
        assert e.last_mode() == ("per-read kernel" if "overflow" in mode else "staged chain")
        cnt = e.counters()
        if mode == "staged_sent_back":
            assert cnt["reads_sent_back"] > 20
        elif "overflow" not in mode:
            assert cnt["reads_sent_back"] == 0
            if "two_pass" in mode:
                assert cnt["ranges_searched"] < 1.1 * cnt["ranges_executed"], (cnt["ranges_searched"], cnt["ranges_executed"])      # what the reference's loop reaches, + a few per cent
    for i, codes in enumerate(reads):
        want = oracle.process(codes)
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])
    e.close()


@pytest.mark.parametrize("mode", ["per_read", "staged", "staged_quads"])
def test_unit_lengths_around_the_two_column_pass(monkeypatch, oracle, mode):
    """dp_forward2p_2c takes the two-parameter alignments of units of 65..128 bases (two columns per lane, even row stride);
    64 and 129 are its neighbours' (one chunk / four chunks).  Units at and next to every boundary, odd and even, with few and
    with many copies (rows), flanked and flush with the read's ends."""
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    rng = np.random.RandomState(20260)
    reads = []
    for u in (63, 64, 65, 66, 67, 99, 100, 101, 126, 127, 128, 129, 130, 255, 256):
        for copies, pre, post in ((6, 40, 40), (12, 0, 300), (25, 500, 0)):
            reads.append(synth.make_read(rng, u, copies, pre, post)[0])
    e = mtr_amd.Engine()
    got = e.process(reads)
    for i, codes in enumerate(reads):
        want = oracle.process(codes)
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])
    e.close()


def test_the_chain_runs_every_batch_and_overlapping_launches_agree(monkeypatch):
    """One arrangement: every batch runs as the staged chain, alone or while another context's launch is in flight (a big batch
    with its alignments and revisions four per wavefront, a small one with one wavefront each); the per-read kernel only takes a
    batch that outgrew a buffer, or MTR_STAGED=0.  The records are the same in every case."""
    for k in ("MTR_STAGED", "MTR_QUAD_MIN"):
        monkeypatch.delenv(k, raising=False)
    reads = [c for _, c in synth.make_reads("headline2k", 5600, 17)]
    a, b = mtr_amd.Engine(), mtr_amd.Engine()
    a.upload(reads); b.upload(reads)
    a.run()
    assert a.last_mode() == "staged chain"
    lone = [[tuple(r) for r in g] for g in a.fetch()]
    kt = a.kernel_times_ms()
    assert all(kt.get("chain_" + p, 0) > 0 for p in ("ranges", "unit_search", "alignments", "revisions")), kt
    a.run_async(); b.run_async()                                   # b is launched while a's launch is pending
    a.wait(); b.wait()
    assert a.last_mode() == "staged chain" and b.last_mode() == "staged chain"
    assert [[tuple(r) for r in g] for g in b.fetch()] == lone
    small = reads[:500]
    a.upload(small); a.run()
    assert a.last_mode() == "staged chain"
    assert [[tuple(r) for r in g] for g in a.fetch()] == lone[:500]
    monkeypatch.setenv("MTR_STAGED", "0")
    a.run()
    assert a.last_mode() == "per-read kernel"
    assert [[tuple(r) for r in g] for g in a.fetch()] == lone[:500]
    a.close(); b.close()


@pytest.mark.parametrize("mode", ["per_read", "staged", "staged_quads", "staged_two_pass", "staged_two_pass_quads"])
def test_kernel_modes_golden(monkeypatch, mode):
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    engines = {"default": mtr_amd.Engine(manhattan=True), "p": mtr_amd.Engine(manhattan=False)}
    checked = 0
    for name, mode in gu.cases():
        if name not in ("2_5_10_20_set", "10_50", "synth_2k", "edge", "3_5", "synth_c3"):
            continue
        reads = gu.read_fasta(gu.input_path(name))
        cap = gu.capture_by_read(name, mode)
        got = engines[mode].process([c for _, c in reads])
        for i, (per_read, g) in enumerate(zip(cap, got)):
            want = [gu.g4_tuple(ev) for ev in per_read["G4"]]
            assert [tuple(r) for r in g] == want, f"{name} [{mode}] " + _diff_msg(i, want, [tuple(r) for r in g])
        checked += 1
    assert checked >= 5
    for e in engines.values():
        e.close()


def test_more_records_than_slots(eng, oracle):
    """A read made of many short distinct repeats reports more records than its 16 + Lmax/100 slots: the library runs
    such reads again with room for all of them (resolve_overflow) - the reference has no limit."""
    rng = np.random.RandomState(31)
    parts = []
    for _ in range(70):
        u = rng.randint(0, 4, size=int(rng.randint(3, 9))).astype(np.uint8)
        parts.append(np.tile(u, 9)); parts.append(rng.randint(0, 4, size=6).astype(np.uint8))
    crowded = np.concatenate(parts)
    reads = [c for _, c in synth.make_reads("c2", 5, 77)] + [crowded] + [c for _, c in synth.make_reads("c2", 5, 78)]
    got = eng.process(reads)
    want = [oracle.process(c) for c in reads]
    slots = 16 + max(len(r) for r in reads) // 100
    assert len(want[5]) > slots, f"the crowded read should overflow its {slots} slots, it reports {len(want[5])}"
    for i in range(len(reads)):
        assert [tuple(r) for r in got[i]] == want[i], _diff_msg(i, want[i], [tuple(r) for r in got[i]])


def test_a_read_sent_back_that_also_has_more_records_than_slots(monkeypatch, oracle):
    """A read the staged chain sends back to the per-read kernel is first given max_rec + 1 slots (the mark it carries says nothing
    about its count); when its true count is larger, resolve_overflow runs it once more with room for all of them."""
    for k, v in MODES["staged_sent_back"].items():
        monkeypatch.setenv(k, v)
    rng = np.random.RandomState(32)
    parts = []
    for _ in range(70):
        u = rng.randint(0, 4, size=int(rng.randint(3, 9))).astype(np.uint8)
        parts.append(np.tile(u, 9)); parts.append(rng.randint(0, 4, size=6).astype(np.uint8))
    crowded = np.concatenate(parts)
    reads = [c for _, c in synth.make_reads("headline2k", 20, 79)] + [crowded] + [c for _, c in synth.make_reads("c2", 10, 80)]
    e = mtr_amd.Engine()
    got = e.process(reads)
    assert e.last_mode() == "staged chain" and e.counters()["reads_sent_back"] > 5
    want = [oracle.process(c) for c in reads]
    slots = 16 + max(len(r) for r in reads) // 100
    assert len(want[20]) > slots + 1, f"the crowded read should overflow its {slots} slots, it reports {len(want[20])}"
    for i in range(len(reads)):
        assert [tuple(r) for r in got[i]] == want[i], _diff_msg(i, want[i], [tuple(r) for r in got[i]])
    e.close()


def _oracle_chunk(reads):
    from tests.oracle_binding import Oracle
    o = Oracle()
    out = [o.process(r) for r in reads]
    o.close()
    return out


def test_config4_mixed_lengths_every_read(eng):
    """BASELINE config 4 shape (units of 50..200 bases, mixed read lengths), 10 000 reads, EVERY read against the oracle
    (run in a process pool); one wavefront per read (the batch is larger than the range-parallel threshold)."""
    from concurrent.futures import ProcessPoolExecutor
    reads = [c for _, c in synth.make_reads("c4", 10000, 44)]
    got = eng.process(reads)
    chunks = [reads[i:i + 250] for i in range(0, len(reads), 250)]
    with ProcessPoolExecutor(max_workers=8) as pool:
        want = [w for ch in pool.map(_oracle_chunk, chunks) for w in ch]
    bad = [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]
    assert not bad, f"{len(bad)} of {len(reads)} reads differ, first: " + _diff_msg(bad[0], want[bad[0]], [tuple(r) for r in got[bad[0]]])


# ---- promoted from the builder-side sweeps (tests/dev/gpu_fuzz.py) into the driver-run suite ------------------------
def _pool_oracle(args):
    manhattan, reads = args
    from tests.oracle_binding import Oracle
    o = Oracle(manhattan=manhattan)
    out = [o.process(r) for r in reads]
    o.close()
    return out


def _oracle_all(reads, manhattan=True, workers=8):
    from concurrent.futures import ProcessPoolExecutor
    order = np.argsort([-len(r) for r in reads])
    parts = [[int(i) for i in order[j::workers * 4]] for j in range(workers * 4)]
    parts = [p for p in parts if p]
    want = [None] * len(reads)
    with ProcessPoolExecutor(max_workers=workers) as pool:
        for idx, res in zip(parts, pool.map(_pool_oracle, [(manhattan, [reads[i] for i in p]) for p in parts])):
            for i, w in zip(idx, res):
                want[i] = w
    return want


def _period_limit_reads(seed, n):
    """units of 440..530 bases around MAX_PERIOD 500 (mTR.h:35): walks that close only above the limit (consensus.c:542-581 `period < 500`),
    revised units that reach string[500] (consensus.c:1012), candidates dropped by `period < MAX_PERIOD`"""
    rng = np.random.RandomState(seed)
    reads = []
    for t in range(n):
        u = int(rng.randint(440, 531)) if t >= 6 else (440, 498, 499, 500, 501, 530)[t]
        c = int(rng.randint(6, 12))
        prof = [(0, 0, 0), (0.5, 1, 1), synth.NANOPORE][t % 3]
        reads.append(synth.make_read(rng, u, c, int(rng.randint(0, 300)), int(rng.randint(0, 300)), profile=prof)[0])
    return reads


_PERIOD_CACHE = {}


@pytest.mark.parametrize("manhattan", [True, False])
@pytest.mark.parametrize("mode", ["per_read", "staged", "staged_quads", "staged_two_pass_quads"])
def test_units_at_the_period_limit(monkeypatch, mode, manhattan):
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    reads = _period_limit_reads(440 + int(manhattan), 18)
    if manhattan not in _PERIOD_CACHE:
        _PERIOD_CACHE[manhattan] = _oracle_all(reads, manhattan)
    want = _PERIOD_CACHE[manhattan]
    e = mtr_amd.Engine(manhattan=manhattan)
    got = e.process(reads)
    e.close()
    bad = [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]
    assert not bad, f"{len(bad)} of {len(reads)} reads differ\n" + _diff_msg(bad[0], want[bad[0]], [tuple(r) for r in got[bad[0]]])
    assert sum(len(w) for w in want) > 0
    assert any(r[3] >= 400 for w in want for r in w), "no reported unit near the limit: the set does not test what it is for"


def _fuzz_slice(seed):
    """a slice of tests/dev/gpu_fuzz.py: every length 1..40, homopolymers, nested repeats (a unit made of a sub-repeat + a spacer), adjacent repeats
    without a spacer, skewed composition, one short unit over a long read, repeats flush with an end and cut inside a copy"""
    rng = np.random.RandomState(seed)
    R = lambda n: rng.randint(0, 4, size=n).astype(np.uint8)
    reads = []
    for L in range(1, 41):
        reads.append(R(L)); reads.append(np.full(L, L % 4, np.uint8))
    for b in range(4):
        for L in (64, 999, 1000, 1001, 3000):
            reads.append(np.full(L, b, np.uint8))
    for _ in range(24):
        sub = R(int(rng.randint(2, 6))); unit = np.concatenate([np.tile(sub, int(rng.randint(3, 8))), R(int(rng.randint(1, 12)))])
        body = np.tile(unit, int(rng.randint(6, 30)))
        body = np.where(rng.rand(len(body)) < 0.03, R(len(body)), body).astype(np.uint8)
        reads.append(np.concatenate([R(int(rng.randint(0, 200))), body, R(int(rng.randint(0, 200)))]))
    for _ in range(24):
        parts = [synth.make_read(rng, int(rng.choice([2, 3, 4, 5, 7, 11, 24, 60, 130])), int(rng.randint(6, 40)), 0, 0, profile=(1, 2, 2))[0]
                 for _k in range(int(rng.randint(2, 4)))]
        reads.append(np.concatenate(parts))
    for _ in range(16):
        p = rng.dirichlet([0.3, 0.3, 0.3, 0.3]); L = int(rng.randint(50, 4000))
        reads.append(rng.choice(4, size=L, p=p).astype(np.uint8))
    for _ in range(4):
        u = R(int(rng.randint(1, 9))); L = int(rng.randint(3000, 12000))
        body = np.tile(u, L // len(u) + 1)[:L]
        reads.append(np.where(rng.rand(L) < rng.choice([0, 0.01, 0.05]), R(L), body).astype(np.uint8))
    for _ in range(24):
        u = int(rng.choice([2, 3, 5, 8, 13, 21, 34, 55, 89, 144])); body, _ = synth.make_read(rng, u, int(rng.randint(6, 30)), 0, 0, profile=(1, 3, 2))
        body = body[int(rng.randint(0, u)): len(body) - int(rng.randint(0, u))]
        reads.append(np.concatenate([body, R(int(rng.randint(0, 500)))]) if rng.randint(0, 2) else np.concatenate([R(int(rng.randint(0, 500))), body]))
    return reads


@pytest.mark.parametrize("manhattan", [True, False])
def test_fuzz_slice_nested_adjacent_and_flush_repeats(monkeypatch, manhattan):
    reads = _fuzz_slice(7 + int(manhattan))
    want = _oracle_all(reads, manhattan)
    for mode in ("per_read", "staged", "staged_quads", "staged_two_pass_quads"):
        for k in ("MTR_STAGED", "MTR_QUAD_MIN", "MTR_TWO_PASS", "MTR_TEST_STAGED_CAPS", "MTR_TEST_STAGED_FLAGS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in MODES[mode].items():
            monkeypatch.setenv(k, v)
        e = mtr_amd.Engine(manhattan=manhattan)
        got = e.process(reads) if mode != "staged" else [r for b in range(0, len(reads), 100) for r in e.process(reads[b:b + 100])]     # (batches of 100: the range finder of few reads)
        e.close()
        bad = [i for i in range(len(reads)) if [tuple(r) for r in got[i]] != want[i]]
        assert not bad, f"[{mode}] {len(bad)} of {len(reads)} reads differ, first (L = {len(reads[bad[0]])})\n" + _diff_msg(bad[0], want[bad[0]], [tuple(r) for r in got[bad[0]]])
    assert sum(len(w) for w in want) > 100


# ---- the WrapDPsize failure, raised by the KERNELS (wrap_around_DP.c:95-99 / :259-262 exit; handle_one_read.c:89-91 clears the record) ------
WRAP_LIMIT = 40000          # cells; the built-in 2e8 is not known to be reachable by a read of at most 833 333 bases, so both sides are lowered


def _oracle_fails(codes, tmp_path, env):
    """does reference semantics exit inside this read?  The oracle exits like the reference, so it runs as a child process."""
    from tests.oracle_binding import ORACLE_DIR
    fa = tmp_path / "one.fa"
    synth.write_fasta(str(fa), [("r", codes)])
    p = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), str(fa)], capture_output=True, env=env)
    if p.returncode != 0:
        assert b"WrapDPsize" in p.stderr, p.stderr[-300:]
    return p.returncode != 0


@pytest.mark.parametrize("mode", ["per_read", "staged", "staged_quads", "staged_two_pass_quads"])
def test_dp_too_large_is_raised_by_the_kernels(monkeypatch, tmp_path, mode):
    """MTR_TEST_WRAP_DP_SIZE lowers WrapDPsize in the kernels (device_util.hip.inc: mtr_dev_wrap_dp_size) and in the oracle alike.  The batch: reads
    whose DPs all stay below the limit, then reads with windows x units beyond it.  The library must return MTR_ERR_DP_TOO_LARGE, name the FIRST read
    (input order) in which the reference would have exited, and still hand over the records of the reads before it - the reference has printed them."""
    monkeypatch.setenv("MTR_TEST_WRAP_DP_SIZE", str(WRAP_LIMIT))
    for k, v in MODES[mode].items():
        monkeypatch.setenv(k, v)
    env = dict(os.environ)
    rng = np.random.RandomState(60)
    small = [rng.randint(0, 4, size=n).astype(np.uint8) for n in (700, 1500)] + [np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 60)]
    small += [synth.make_read(rng, 12, 14, 100, 100)[0], synth.make_read(rng, 30, 9, 50, 300)[0]]
    big = [c for _, c in synth.make_reads("headline2k", 6, 61)]                     # unit 100 x 10 copies: windows of ~1 000 rows x 100 columns
    reads = small + big[:3] + small[:2] + big[3:]
    fails = [_oracle_fails(c, tmp_path, env) for c in reads]
    assert not any(fails[:len(small)]) and any(fails), fails
    first = fails.index(True)
    from tests.oracle_binding import Oracle
    orc = Oracle()                                              # (created under the lowered limit: handle_one_read.c:89-91's clear is part of the records)
    want = [orc.process(c) for c in reads[:first]]
    orc.close()
    e = mtr_amd.Engine()
    with pytest.raises(mtr_amd.MtrError, match="MTR_ERR_DP_TOO_LARGE"):
        e.process(reads)
    assert e.first_failed_read() == first, (e.first_failed_read(), first, fails)
    data, counts = e.fetch_packed(first)
    assert list(counts) == [len(w) for w in want]
    recs = (mtr_amd.CRecord * max(int(counts.sum()), 1))()
    import ctypes as C
    buf = C.create_string_buffer(data, len(data)) if data else C.create_string_buffer(1)
    assert e.lib.mtr_unpack_records(buf, len(data), int(counts.sum()), recs) == 0
    got = e._unpack(recs, counts, first)
    for i in range(first):
        assert [tuple(r) for r in got[i]] == want[i], _diff_msg(i, want[i], [tuple(r) for r in got[i]])
    assert sum(len(w) for w in want) > 0
    # the same batch without its failing reads runs clean on the same context (the failure is latched per run, not per context)
    ok_reads = [c for c, f in zip(reads, fails) if not f]
    orc = Oracle()
    got = e.process(ok_reads)
    for i, c in enumerate(ok_reads):
        assert [tuple(r) for r in got[i]] == orc.process(c)
    orc.close()
    e.close()


def test_dp_too_large_on_the_command_line(monkeypatch, tmp_path):
    """mTR (the C host) on the same input: the reads before the failing one are printed, then the reference's message, then a failing exit status -
    byte for byte what the oracle's command line prints before IT exits (handle_one_file.c:281-287 prints read by read)."""
    from tests.oracle_binding import ORACLE_DIR
    monkeypatch.setenv("MTR_TEST_WRAP_DP_SIZE", str(WRAP_LIMIT))
    rng = np.random.RandomState(62)
    reads = [synth.make_read(rng, 12, 14, 100, 100)[0], synth.make_read(rng, 30, 9, 50, 300)[0], np.tile(np.array([3, 3, 0, 2, 2, 2], np.uint8), 60)]
    reads += [c for _, c in synth.make_reads("headline2k", 3, 63)] + [synth.make_read(rng, 5, 30, 10, 10)[0]]
    fa = tmp_path / "reads.fa"
    synth.write_fasta(str(fa), [(f"read{i}", c) for i, c in enumerate(reads)])
    o = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), str(fa)], capture_output=True)
    assert o.returncode != 0 and b"WrapDPsize" in o.stderr and o.stdout.count(b"\n") >= 3
    subprocess.run(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtr_amd", "host"), "mTR"], check=True)
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mtr_amd", "host", "mTR")
    p = subprocess.run([exe, str(fa)], capture_output=True)
    assert p.returncode != 0 and b"WrapDPsize" in p.stderr, p.stderr[-300:]
    assert p.stdout == o.stdout


# ---- file-order mode: the reference's behaviour on a multi-read file ---------------------------------------------------
def _file_order_case():
    """widely mixed lengths + reads ending inside their repeat right after a longer read (the one-past DP row)"""
    reads = [c for _, c in synth.make_mixed_file(150, 21)]
    rng = np.random.RandomState(5)
    for n in (3000, 9, 1200, 31, 4, 700):
        reads.append(rng.randint(0, 4, size=n).astype(np.uint8))
        reads.append(np.concatenate([rng.randint(0, 4, size=max(0, n // 3)).astype(np.uint8), np.tile(rng.randint(0, 4, size=7).astype(np.uint8), 40)]))
    return reads


@pytest.mark.parametrize("manhattan,split", [(True, "per_read"), (False, "per_read"), (True, "staged"), (False, "staged"), (True, "staged_quads"), (True, "staged_two_pass")])
def test_file_order_mode_matches_reference_behaviour_on_a_file(monkeypatch, manhattan, split):
    """mtr_upload_batch_in_file: the records equal the oracle's file-order mode (pinned to the reference run on whole
    files, tests/test_oracle_golden.py) whatever the batch boundaries, in both kernel modes; and they differ from the
    isolated records on this input, so the test can fail."""
    from tests.oracle_binding import Oracle
    for k, v in MODES[split].items():
        monkeypatch.setenv(k, v)
    reads = _file_order_case()
    o = Oracle(manhattan)
    o.set_file_order(True)
    want = [o.process(c) for c in reads]
    o.close()
    e = mtr_amd.Engine(manhattan=manhattan)
    fs = mtr_amd.FileState()
    got = []
    for lo, hi in ((0, 1), (1, 8), (8, 97), (97, len(reads))):
        got += e.process_in_file(reads[lo:hi], fs)
    for i in range(len(reads)):
        assert [tuple(r) for r in got[i]] == want[i], _diff_msg(i, want[i], [tuple(r) for r in got[i]])
    iso = e.process(reads)
    assert sum(1 for i in range(len(reads)) if [tuple(r) for r in iso[i]] != want[i]) >= 3
    # a shard that starts in the middle of the file: the state of the reads before it comes from mtr_file_state_skip
    fs2 = mtr_amd.FileState()
    fs2.skip(reads[:60])
    part = e.process_in_file(reads[60:110], fs2)
    for i in range(60, 110):
        assert [tuple(r) for r in part[i - 60]] == want[i], _diff_msg(i, want[i], [tuple(r) for r in part[i - 60]])
    fs.close(); fs2.close(); e.close()


def test_file_order_ranges_match_oracle():
    """K1 alone in file-order mode against the oracle's DI ranges (bit patterns of the DI values included)"""
    from tests.oracle_binding import Oracle
    reads = _file_order_case()[:80]
    for manhattan in (True, False):
        o = Oracle(manhattan)
        o.set_file_order(True)
        e = mtr_amd.Engine(manhattan=manhattan)
        fs = mtr_amd.FileState()
        e.upload(reads, fs)
        got = e.test_ranges()
        for i, c in enumerate(reads):
            want = o.ranges(c)
            assert got[i] == want, f"read {i} (L={len(c)}): {len(want)} ranges expected, {len(got[i])} produced"
        fs.close(); e.close(); o.close()


@pytest.mark.parametrize("name", ["mixed_lengths", "stale_org_base"])
@pytest.mark.parametrize("mode", ["default", "p"])
def test_file_order_mode_golden(eng, eng_p, name, mode):
    """against the capture of the unmodified reference run on the whole file (tests/golden/file_order/): candidate ranges
    (G1, DI bit patterns included) and inserted records (G4) of every read.  stale_org_base.fa holds reads whose last
    DP row reads orgInputString[L], i.e. a base of the previous longer read."""
    import gzip, json
    fo = os.path.join(gu.GOLDEN, "file_order")
    reads = [c for _, c in gu.read_fasta(os.path.join(fo, name + ".fa"))]
    per_read = []
    with gzip.open(os.path.join(fo, f"{name}.{mode}.cap.jsonl.gz"), "rt") as fh:
        for line in fh:
            ev = json.loads(line)
            if ev["t"] == "G1":
                per_read.append({"G1": ev, "G4": []})
            elif ev["t"] == "G4":
                per_read[-1]["G4"].append(ev)
    assert len(per_read) == len(reads)
    e = eng if mode == "default" else eng_p
    fs = mtr_amd.FileState()
    e.upload(reads, fs)
    ranges = e.test_ranges()
    e.run()
    got = e.fetch()
    fs.close()
    for i in range(len(reads)):
        assert ranges[i] == gu.g1_usable(per_read[i]["G1"]), f"read {i}: ranges differ"
        want = [gu.g4_tuple(ev) for ev in per_read[i]["G4"]]
        assert [tuple(r) for r in got[i]] == want, _diff_msg(i, want, [tuple(r) for r in got[i]])
