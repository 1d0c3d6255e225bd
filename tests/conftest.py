import os
import sys

# torch's default is one intra-op thread per core pair -- 128 on the GPU box -- in THIS process and in every child the suite starts (two-rank runs, the
# subprocess suites of tests/_background.py): ten such pools on one box spend their time spinning at barriers (r06, measured: driver scenarios 7 -> 80 s when
# the children ran beside them).  The CPU work here is the oracle on batches of 2 - 4: 16 threads is where bench.py's cpu_baseline leg runs it too.
_THREADS = str(min(16, os.cpu_count() or 16))
os.environ.setdefault("OMP_NUM_THREADS", _THREADS)
os.environ.setdefault("MKL_NUM_THREADS", _THREADS)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_here():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """the tests that only wait for a child suite (tests/_background.py) run last: their children get the whole run to finish in"""
    from tests import _background
    items.sort(key=lambda it: getattr(it, "originalname", it.name) in _background.SPECS)          # (stable: everything else keeps its order)


def pytest_collection_finish(session):
    from tests import _background
    if os.environ.get("CLIMB_AMD_BACKGROUND_CHILD") or os.environ.get("PYTEST_XDIST_WORKER") or session.config.option.collectonly:          # (xdist workers each collect everything: their tests start their own child)
        return
    names = [getattr(it, "originalname", it.name) for it in session.items]
    todo = [n for n in names if n in _background.SPECS]
    if todo and len(names) > len(todo) and _gpu_here():          # (selected alone, a test starts its child itself)
        for n in todo:
            _background.start(n)


def pytest_sessionfinish(session, exitstatus):
    from tests import _background
    _background.stop_all()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _drop_big_files_after_test(request):
    """The driver scenarios write full checkpoints (470 MB each, several per task and rank) under `tmp_path`; pytest keeps the last three
    base temps, i.e. 35+ GB of /tmp after three runs of the CPU suite -- enough to fill the disk and fail `torch.save` in a later run.
    Nothing reads them once the test has made its assertions: files above 4 MB are removed, the small ones (results, reports) stay."""
    yield
    tmp = request.node.funcargs.get("tmp_path") if hasattr(request.node, "funcargs") else None
    if tmp is None:
        return
    for dp, _, fs in os.walk(str(tmp)):
        for f in fs:
            p = os.path.join(dp, f)
            try:
                if not os.path.islink(p) and os.path.getsize(p) > (4 << 20):
                    os.remove(p)
            except OSError:
                pass
