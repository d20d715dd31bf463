import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _whole_host_lists(request, monkeypatch):
    """the host-pointer entry points cut long query lists into pipelined chunks (csrc/hostpath.hip); the tests of
    these modules are about how ONE launch descends (split descents, dense layers in a launch of their own), so their
    lists run whole"""
    if request.module.__name__.split(".")[-1] in ("test_gpu_tiny", "test_gpu_locality", "test_gpu_fullsize") and \
            "PHNSW_HOST_CHUNKS" not in os.environ:
        monkeypatch.setenv("PHNSW_HOST_CHUNKS", "4000000000,1024,4096")
    yield
