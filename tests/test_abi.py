"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/aps.h declares,
and refuses to compute without a gfx950 device (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "aps.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(aps_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(aps):
    lib = ctypes.CDLL(aps._capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in aps.h but not exported"


def test_binding_table_matches_header(aps):
    assert sorted(aps._capi.EXPORTED_SYMBOLS) == declared_symbols()


def test_version_and_error_string(aps):
    assert aps.lib.aps_version() == 100
    assert isinstance(aps.lib.aps_last_error(), bytes)


def test_no_silent_cpu_fallback(aps):
    """Without a device a compute call must fail with APS_E_DEVICE, never return numbers."""
    if aps.lib.aps_device_count() > 0:
        pytest.skip("device present")
    from importlib import import_module

    fm = import_module(aps.__name__ + ".featureMatching")
    a = np.random.default_rng(0).random((8, 128), dtype=np.float32)
    with pytest.raises(aps.ApsError) as e:
        fm.matchFeaturesScratch(a, a)
    assert e.value.code == aps._capi.APS_E_DEVICE


def test_product_does_not_import_oracle(aps):
    """The product package must not reference oracle/ anywhere."""
    pkg_dir = os.path.dirname(aps.__file__)
    for base, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(base, f), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "libaps_oracle" not in src, f
