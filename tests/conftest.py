import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_usable():
    """A built HIP library and a GPU device node -- decided WITHOUT a HIP call: the collecting process must not initialise the
    GPU before tests/test_00_gpu_multirank.py has started its child processes."""
    try:
        from phylomap_amd import _lib
        return os.path.exists(_lib.LIB_PATH) and os.path.exists("/dev/kfd")
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _gpu_usable():
        return
    skip = pytest.mark.skip(reason="needs a HIP device and the built library (run on the GPU box with -m gpu)")
    for it in gpu_items:
        it.add_marker(skip)
