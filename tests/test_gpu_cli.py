"""The drop-in command line on the GPU box: mtr_amd/host/mTR [-a] [-p] <fasta> must print byte for byte what the
reference printed (tests/golden/*.stdout, isolated semantics) — FASTA in, report/alignment lines out."""
import glob
import json
import os
import subprocess

import pytest

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mtr_amd", "host")
FLAGS = {"default": [], "p": ["-p"], "a": ["-a"]}


@pytest.fixture(scope="module")
def cli():
    from mtr_amd import build as b
    b.build()
    subprocess.run(["make", "-s", "-C", HOST, "mTR"], check=True)
    return os.path.join(HOST, "mTR")


def _cases():
    out = []
    for p in sorted(glob.glob(os.path.join(gu.GOLDEN, "*.stdout"))):
        name, mode, _ = os.path.basename(p).rsplit(".", 2)
        out.append((name, mode))
    return out


@pytest.mark.parametrize("name,mode", _cases())
def test_cli_stdout_matches_reference(cli, name, mode):
    p = subprocess.run([cli, *FLAGS[mode], gu.input_path(name)], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[:500]
    want = open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()
    assert p.stdout == want


def test_cli_timing_block_and_errors(cli, tmp_path):
    p = subprocess.run([cli, "-c", gu.input_path("3_5")], capture_output=True)
    assert p.returncode == 0 and p.stderr.decode().startswith("Computation time\n") and "Count of queries" in p.stderr.decode()
    bad = tmp_path / "bad.fa"
    bad.write_text(">x\nACGTNACGT\n")
    q = subprocess.run([cli, str(bad)], capture_output=True)
    assert q.returncode != 0 and b"Invalid character: N" in q.stderr
    r = subprocess.run([cli, "-m", "1.5", gu.input_path("3_5")], capture_output=True)
    assert r.returncode != 0 and b"must range from 0 to 1" in r.stderr


@pytest.mark.parametrize("flags", [[], ["-p"], ["-a"]])
def test_cli_file_order_mode(cli, flags):
    """-B = the reference's behaviour on a multi-read file (include/mtr_hip.h, file-order mode).  Checked against the
    oracle's -B mode, which is pinned to the reference run on whole files at every capture point and, like this driver,
    breaks chaining ties by insertion order (the reference: by heap address)."""
    from tests.oracle_binding import ORACLE_DIR
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    for name in ("mixed_lengths", "stale_org_base"):
        fa = os.path.join(gu.GOLDEN, "file_order", name + ".fa")
        want = subprocess.run([os.path.join(ORACLE_DIR, "mtr_oracle_cli"), "-B", *flags, fa], capture_output=True, check=True).stdout
        p = subprocess.run([cli, "-B", *flags, fa], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[:500]
        assert p.stdout == want, name
        if not flags:                                              # the isolated answers differ on these files
            iso = subprocess.run([cli, fa], capture_output=True)
            assert iso.returncode == 0 and iso.stdout != want


def _report_lines_by_read(stdout: bytes):
    """{read ID: [(rep_start, rep_end, repeat_len, period, copies, matches, mismatches, insertions, deletions, unit)]} of the report lines
    (13 tab-separated fields, chaining.cpp:127-143); the blocks -a prints between them are not report lines"""
    out = {}
    for ln in stdout.decode().split("\n"):
        f = ln.split("\t")
        if len(f) != 13 or not f[1].isdigit():
            continue
        out.setdefault(f[0], []).append((int(f[2]) - 1, int(f[3]) - 1, int(f[4]), int(f[5]), int(f[6]), int(f[7]), int(f[9]), int(f[10]), int(f[11]), f[12]))
    return out


def test_reference_front_end_with_the_binding_of_integration_md():
    """oracle/_ref/mTR_ref_gpu = the UNMODIFIED reference objects (main, reader, chaining, printers) with handle_one_file()
    replaced by the binding of INTEGRATION.md (oracle/ref_gpu_binding.c), i.e. the drop-in the C-ABI is for.  Its stdout is compared
    with the reference's own, read by read.  The reference's chaining breaks ties between chains of EQUAL score by the heap addresses
    of its Alignment objects (chaining.cpp:201), which differ between the two programs; so a read's report may differ from the golden
    one only like this: every line it prints is one of the records the reference inserted for that read (golden G4: the records are
    exact), and the chain it prints has the same total score (sum of matches, chaining.cpp:262-330) as the chain the reference printed -
    a tie.  Everything else is a regression."""
    exe = os.path.join(ROOT, "oracle", "_ref", "mTR_ref_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/mTR_ref_gpu was not built (needs /root/reference at build time)")
    cases = [(n, m) for n in ("3_5", "10_50", "2_5_10_20_set", "worm_chrI") for m in ("default", "p", "a")] + [("synth_c2", "default"), ("synth_c4", "default"), ("synth_2k", "default")]
    # which cases hold a read whose chain depends on the order of its alignments (tests/golden/tied_chains.py decides that on the CPU from the golden G4 records:
    # the sweep under ten orders of a read's records): only those may differ from the golden stdout, and only by a tie; every other case is exact
    with open(os.path.join(gu.GOLDEN, "tied_chain_cases.json")) as fh:
        tied_cases = set(json.load(fh)["reads_with_an_order_dependent_chain"])
    exact = tied_reads = 0
    for name, mode in cases:
        p = subprocess.run([exe, *FLAGS[mode], gu.input_path(name)], capture_output=True)
        assert p.returncode == 0, p.stderr.decode()[:500]
        ref = open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()
        if p.stdout == ref:
            exact += 1
            continue
        assert f"{name}.{'p' if mode == 'p' else 'default'}" in tied_cases, (name, mode, "no read of this case has an order-dependent chain: its stdout must be the reference's byte for byte")
        reads = gu.read_fasta(gu.input_path(name))
        cap = gu.capture_by_read(name, "p" if mode == "p" else "default")
        inserted = {}
        for (rid, _), per_read in zip(reads, cap):
            inserted.setdefault(rid, set()).update((t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[13]) for t in map(gu.g4_tuple, per_read["G4"]))
        want, got = _report_lines_by_read(ref), _report_lines_by_read(p.stdout)
        assert set(got) <= set(inserted) and set(want) == set(got), (name, mode, "reads reported", sorted(set(want) ^ set(got))[:5])
        for rid in want:
            if want[rid] == got[rid]:
                continue
            tied_reads += 1
            stray = [ln for ln in got[rid] if ln not in inserted[rid]]
            assert not stray, (name, mode, rid, "a printed repeat is not a record the reference inserted", stray[:2])
            assert sum(ln[5] for ln in got[rid]) == sum(ln[5] for ln in want[rid]), (name, mode, rid, "the printed chain is not a tie of the reference's")
    n_must = sum(1 for n, m in cases if f"{n}.{'p' if m == 'p' else 'default'}" not in tied_cases)
    assert exact >= n_must and n_must >= 8, f"only {exact} of {len(cases)} identical ({n_must} have no order-dependent chain)"


def test_cli_reports_the_engine_library_it_bound(cli):
    """mTR -c: after the reference's timing block, the resolved path of the library that computed the records - the product library,
    not something a leaked $MTR_LIB pointed at (the replay engine of the CPU tests would print the same report from recorded records)."""
    env = {k: v for k, v in os.environ.items() if k != "MTR_LIB"}
    p = subprocess.run([cli, "-c", gu.input_path("3_5")], capture_output=True, env=env)
    assert p.returncode == 0, p.stderr.decode()[:500]
    err = p.stderr.decode()
    assert "Computation time" in err
    line = [ln for ln in err.splitlines() if ln.endswith("engine library")]
    assert len(line) == 1 and os.path.realpath(line[0].split("\t")[0]) == os.path.realpath(os.path.join(ROOT, "mtr_amd", "libmtr_hip.so")), err
