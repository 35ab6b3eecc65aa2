import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

SHARDED_READS = 200003
SHARDED_WORLD = 2
_sharded = {"directory": None, "log": ""}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_sessionstart(session):
    """A `-m gpu` session on a box with a GPU: the two rank sharded run (tests/tools/sharded_ranks.py) goes first, as
    a child process started before this process touches the GPU; tests/test_gpu_sharded.py reads what it wrote."""
    expression = session.config.getoption("-m") or ""
    if "gpu" not in expression or "not gpu" in expression:
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:             # counting devices does not initialise the GPU
            return
    except Exception:
        return
    directory = tempfile.mkdtemp(prefix="phq_sharded_")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "sharded_ranks.py"), directory, str(SHARDED_READS), str(SHARDED_WORLD)],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    _sharded["directory"] = directory
    _sharded["log"] = done.stdout.decode(errors="replace")[-4000:]


@pytest.fixture(scope="session")
def sharded_run():
    return dict(_sharded, reads=SHARDED_READS, world=SHARDED_WORLD)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
