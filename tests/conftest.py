import os
import subprocess
import sys

import pytest

try:        # PyTorch ships its own HIP runtime (same soname as /opt/rocm's, which libmtr_hip.so links): whichever is loaded first serves both, and
    import torch  # noqa: F401  torch finds no GPU through the other one - so torch first, whatever subset of the test files is run
except Exception:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_DIR = os.path.join(ROOT, "oracle")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Evidence cheapest-first: the in-process parity files, then the stage-level ones, then the command line, and the
# multi-process tests (launcher, torchrun, RCCL) last - a plumbing failure must never stand between `pytest -x` and the parity tests.
_FILE_ORDER = ["test_gpu_parity.py", "test_gpu_stages.py", "test_gpu_cli.py", "test_gpu_multi.py"]
_LAST_IN_FILE = ("launcher", "rccl", "bench_strong", "c_host")          # inside test_gpu_multi.py: in-process tests before the ones that start ranks


def pytest_collection_modifyitems(session, config, items):
    def key(ix_item):
        ix, item = ix_item
        fname = os.path.basename(str(item.fspath))
        rank = _FILE_ORDER.index(fname) if fname in _FILE_ORDER else -1      # files not named keep their place in front (the CPU suite)
        late = int(any(w in item.name for w in _LAST_IN_FILE)) if fname == "test_gpu_multi.py" else 0
        return (rank, late, ix)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


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
