import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Same in-tree MIOpen user find-db / kernel cache as bench.py: the PyTorch wheel ships no gfx950 kernel
# database, so without it every convolution shape is JIT-compiled on first use (minutes for the
# ResNet-50 @224 / @448 tests on a fresh box).  Only affects the time to the first launch.
_MIOPEN_DIR = os.path.join(ROOT, ".miopen")
os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(_MIOPEN_DIR, "db"))
os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_MIOPEN_DIR, "cache"))
for _d in (os.environ["MIOPEN_USER_DB_PATH"], os.environ["MIOPEN_CUSTOM_CACHE_DIR"]):
    os.makedirs(_d, exist_ok=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must never silently pass without a GPU."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _settle_device_state(request):
    """GPU tests: after each test, finish the device's work, drop what the fused backbone's side channels still hold and
    collect garbage NOW -- hipGraph objects and parked tensors are then destroyed between tests, not by a collection that
    happens to run inside a later test's stream capture (seen twice as `Fatal Python error: Aborted` while collecting)."""
    yield
    if "gpu" not in request.keywords:
        return
    import gc

    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
        try:
            from peclr_amd import bn2d

            bn2d.end_backward()
        except Exception:  # pragma: no cover -- the package failing to import is some test's own finding
            pass
        gc.collect()
        torch.cuda.synchronize()
