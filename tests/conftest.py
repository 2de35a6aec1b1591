import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import apsamd

        return apsamd.lib.aps_device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def aps():
    import apsamd

    return apsamd


@pytest.fixture(scope="session")
def gpu(aps):
    """The product package with a usable device; gpu tests must not silently pass without one."""
    n = aps.lib.aps_device_count()
    assert n > 0, "a -m gpu test was started without a gfx950 device"
    return aps


@pytest.fixture(scope="module", autouse=True)
def _hand_back_device_memory_after_each_module(request):
    """The -m gpu suite runs in ONE process: after the full-size modules (256 x 4K, 500 views) its workspace pools and torch's
    cache hold most of the 288 GB, and a test that starts a child process on the same GPU finds none left (round 5: RCCL's
    bring-up allocation failed in one full-suite run in three).  Caches go back to the driver after every module."""
    yield
    if "apsamd" not in sys.modules:
        return
    try:
        import torch

        if not torch.cuda.is_available():
            return
        from importlib import import_module

        import_module(sys.modules["apsamd"].__name__ + ".pipeline").release_device_memory()
    except Exception:  # housekeeping must never fail a test
        pass
