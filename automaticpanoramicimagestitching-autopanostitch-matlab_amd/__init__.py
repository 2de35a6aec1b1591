"""aps-hip: the MI355X-native (gfx950) hot path of AutoPanoStitch behind the reference's operator surface.

Layout: ``csrc/`` holds the hand-written HIP kernels and the C ABI (include/aps.h); the Python modules
mirror the reference's MATLAB operators one to one (featureMatching, imageMatching, renderPanorama,
blending, imageProcessing) on top of that ABI.  Importing this package loads lib/libaps_hip.so and
fails loudly if it is missing — there is no CPU, PyTorch or oracle fallback.
"""
from . import _capi  # noqa: F401  (loads the shared library; ImportError if absent)
from ._capi import ApsError, lib  # noqa: F401

__all__ = ["_capi", "ApsError", "lib"]
