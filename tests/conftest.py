import os
import sys

import pytest

# the oracle is OpenMP code; on a 256-core GPU host the default team size makes tiny problems crawl
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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


@pytest.fixture(autouse=True, scope="session")
def _oracle_follows_the_engines_norm_order():
    """GPU sessions: the oracle restates the order of the engine's 1..4-row fused-norm launches (oracle/model.py ENGINE_RULE)"""
    if _gpu_present():
        from oracle import model as om
        from vllm_rs_amd import _lib
        om.ENGINE_RULE = _lib.load()
    yield
