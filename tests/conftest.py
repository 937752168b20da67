import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _gpu_present():
    try:
        from vllm_rs_amd import _lib
        return _lib.load().vra_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if any("gpu" in it.keywords for it in items) and not _gpu_present():
        skip = pytest.mark.skip(reason="no GPU visible in this container")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)
