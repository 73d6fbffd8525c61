import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_cli():
    """Path of oracle/mtr_oracle_cli, (re)built from oracle/*.c — test infrastructure only."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    return os.path.join(ORACLE_DIR, "mtr_oracle_cli")


@pytest.fixture(scope="session")
def oracle_lib():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)
    from tests import oracle_binding

    return oracle_binding.load()
