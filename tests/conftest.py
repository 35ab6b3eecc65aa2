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
_sharded = {"directory": None, "log": "", "returncode": None}
BENCH_RANKS_COMMAND = ["--gpus", "2", "--share-device", "--reads", "2000000", "--steps", "3", "--warmup", "1", "--no-extra-legs", "--no-cpu-baseline"]
_bench_ranks = {"log": "", "stdout": "", "returncode": None, "command": BENCH_RANKS_COMMAND}


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
    _sharded["directory"] = directory
    # a hang or a failure of a child fails the tests that read its output (with its log), never the whole session
    try:
        done = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "sharded_ranks.py"), directory, str(SHARDED_READS), str(SHARDED_WORLD)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
        _sharded["returncode"] = done.returncode
        _sharded["log"] = done.stdout.decode(errors="replace")[-4000:]
    except subprocess.TimeoutExpired as error:
        _sharded["log"] = "timed out after 900 s\n" + (error.stdout or b"").decode(errors="replace")[-4000:]
    except OSError as error:
        _sharded["log"] = "could not start: %s" % error
    # bench.py's own N > 1 branch (it starts its ranks itself: python -m torch.distributed.run), two ranks on this one GPU
    try:
        done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_RANKS_COMMAND,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
        _bench_ranks["returncode"] = done.returncode
        _bench_ranks["stdout"] = done.stdout.decode(errors="replace")       # what the driver reads: the contract line is its LAST line
        _bench_ranks["log"] = done.stderr.decode(errors="replace")[-8000:] + "\n" + _bench_ranks["stdout"][-8000:]
    except subprocess.TimeoutExpired as error:
        _bench_ranks["log"] = "timed out after 900 s\n" + (error.stdout or b"").decode(errors="replace")[-4000:]
    except OSError as error:
        _bench_ranks["log"] = "could not start: %s" % error


def pytest_sessionfinish(session, exitstatus):
    import shutil
    if _sharded["directory"]:
        shutil.rmtree(_sharded["directory"], ignore_errors=True)
        _sharded["directory"] = None


@pytest.fixture(scope="session")
def sharded_run():
    return dict(_sharded, reads=SHARDED_READS, world=SHARDED_WORLD)


@pytest.fixture(scope="session")
def bench_ranks_run():
    return dict(_bench_ranks)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
