"""The accuracy scorers of tools/ (SURVEY.md §8f-3): rotation test and wrap-around match ratio, on cases with known answers."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import accuracy  # noqa: E402


def test_rotation():
    lib = accuracy.load_scorer()
    assert lib.us_is_rotation(b"ACGT", b"GTAC") == 1
    assert lib.us_is_rotation(b"ACGT", b"ACGT") == 1
    assert lib.us_is_rotation(b"ACGT", b"ACTG") == 0
    assert lib.us_is_rotation(b"ACG", b"ACGT") == 0
    assert lib.us_is_rotation(b"AAAC", b"CAAA") == 1


def test_match_ratio():
    lib = accuracy.load_scorer()
    assert lib.us_match_ratio(b"ACGTTGCA", b"ACGTTGCA") == 1.0
    assert lib.us_match_ratio(b"ACGTTGCA", b"TTGCAACG") == 1.0                 # a rotation aligns perfectly round the cycle
    assert lib.us_match_ratio(b"ACGTTGCAACGTTGCA", b"ACGTTGCA") == 1.0         # two copies against the cycle
    r = lib.us_match_ratio(b"ACGTTGCA", b"ACGTAGCA")                          # one substitution: 7 of 8 columns match
    assert abs(r - 7 / 8) < 1e-12
    r = lib.us_match_ratio(b"ACGTTGCA", b"ACGTGCA")                           # one deletion: 7 matches on 8 columns
    assert abs(r - 7 / 8) < 1e-12
    assert lib.us_match_ratio(b"", b"ACG") == 0.0


def test_score_report_lines():
    lib = accuracy.load_scorer()
    truth = ["ACGTTGCA", "AC"]
    report = "0\t100\t1\t80\t80\t8\t10\t78\t0.97\t1\t1\t0\tTTGCAACG\n1\t50\t1\t20\t20\t2\t10\t20\t1.0\t0\t0\t0\tAG\n"
    s = accuracy.score(lib, report, truth)
    assert s["exact_rotation_reads"] == 1 and s["report_lines"] == 2 and s["ratio>=1"] == 1
