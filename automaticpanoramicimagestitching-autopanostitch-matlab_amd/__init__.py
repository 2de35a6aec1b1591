"""aps-hip: the MI355X-native (gfx950) hot path of AutoPanoStitch behind the reference's operator surface.

Layout: ``csrc/`` holds the hand-written HIP kernels and the C ABI (include/aps.h); the Python modules
mirror the reference's MATLAB operators one to one (featureMatching, imageMatching, renderPanorama,
blending, imageProcessing) on top of that ABI.  Importing this package loads lib/libaps_hip.so and
fails loudly if it is missing — there is no CPU, PyTorch or oracle fallback.
"""
import os as _os

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a
# queue serialise.  The pipeline drives eight SIFT streams plus the caller's copy streams: with 4 queues host uploads that
# run beside the feature extraction cost it 17 ms per 64 x 4K step, with 8 they cost 3 (2 and 16 queues are slower;
# DESIGN.md section 5).  Effective only if set before the first HIP call of the process; a user's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _capi  # noqa: E402,F401  (loads the shared library; ImportError if absent)
from ._capi import ApsError, lib  # noqa: E402,F401

__all__ = ["_capi", "ApsError", "lib"]
