"""Host side of the boundary on CPU: FASTA reader, chaining and printers of mtr_amd/host (plain C) against the
reference's golden stdout, fed with the reference's own recorded records (G4) instead of the GPU library."""
import os
import subprocess
import tempfile

import pytest

from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "mtr_amd", "host")


@pytest.fixture(scope="module")
def selftest():
    subprocess.run(["make", "-s", "-C", HOST, "host_selftest"], check=True)
    return os.path.join(HOST, "host_selftest")


def _cases():
    out = []
    for name, mode in gu.cases():
        out.append((name, mode, False))
        if mode == "default" and os.path.exists(os.path.join(gu.GOLDEN, f"{name}.a.stdout")):
            out.append((name, "a", True))
    return out


@pytest.mark.parametrize("name,mode,align", _cases())
def test_chain_and_print_match_reference_stdout(selftest, name, mode, align):
    cap = gu.capture_by_read(name, "default" if mode == "a" else mode)
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as tf:
        for i, per_read in enumerate(cap):
            for ev in per_read["G4"]:
                t = gu.g4_tuple(ev)
                tf.write(" ".join([str(i)] + [str(x) for x in t[:13]] + [t[13]]) + "\n")
        path = tf.name
    try:
        cmd = [selftest, gu.input_path(name), path] + (["-a"] if align else [])
        p = subprocess.run(cmd, capture_output=True, check=True)
    finally:
        os.unlink(path)
    want = open(os.path.join(gu.GOLDEN, f"{name}.{mode}.stdout"), "rb").read()
    assert p.stdout == want


def test_fasta_reader_rejects_non_acgt(selftest, tmp_path):
    fa = tmp_path / "bad.fa"
    fa.write_text(">r\nACGTN\n")
    rec = tmp_path / "r.txt"
    rec.write_text("")
    p = subprocess.run([selftest, str(fa), str(rec)], capture_output=True)
    assert p.returncode != 0 and b"Invalid character: N" in p.stderr


def test_cli_fails_loudly_without_gpu():
    """no CPU fallback: the driver must refuse to run when there is no HIP device (this container has none)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    subprocess.run(["make", "-s", "-C", HOST, "mTR"], check=True)
    p = subprocess.run([os.path.join(HOST, "mTR"), gu.input_path("3_5")], capture_output=True)
    assert p.returncode != 0 and b"no usable HIP device" in p.stderr and p.stdout == b""
