import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
