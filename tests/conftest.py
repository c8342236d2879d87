import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must not silently pass: gpu tests fail loudly there
    # (the product has no CPU path).  Without -m, gpu tests are skipped when no GPU exists.
    markexpr = config.getoption("-m") or ""
    if _has_gpu() or "gpu" in markexpr:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
