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
