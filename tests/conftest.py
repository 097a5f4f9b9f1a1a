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
    """A built HIP library that sees a device (no torch import: counting devices must stay cheap on the CPU box)."""
    try:
        from phylomap_amd import _lib
        return _lib.load().phm_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _gpu_usable():
        return
    skip = pytest.mark.skip(reason="needs a HIP device and the built library (run on the GPU box with -m gpu)")
    for it in gpu_items:
        it.add_marker(skip)
