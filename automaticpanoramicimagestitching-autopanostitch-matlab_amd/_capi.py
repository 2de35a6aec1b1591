"""ctypes binding of libaps_hip.so (the C ABI declared in include/aps.h).

This module is plumbing only: it loads the in-tree shared library, declares the argument types of
every exported symbol and turns non-zero status codes into ``ApsError``.  There is deliberately no
fallback of any kind: if the library is missing the import fails, and if no gfx950 device is visible
every compute call raises ``ApsError(APS_E_DEVICE)``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APS_LIB_PATH") or os.path.join(_HERE, "lib", "libaps_hip.so")   # APS_LIB_PATH: the `make debug` library for the probes

APS_OK, APS_E_ARG, APS_E_DIM, APS_E_TYPE, APS_E_OOM, APS_E_DEVICE, APS_E_INTERNAL, APS_E_CAP = (
    0, -1, -2, -3, -4, -5, -6, -7)
APS_COLMAJOR, APS_ROWMAJOR = 0, 1
APS_ROBUST_RANSAC, APS_ROBUST_MLESAC = 0, 1
APS_RESIZE_BILINEAR, APS_RESIZE_BICUBIC = 0, 1
APS_WARP_NEAREST, APS_WARP_BILINEAR, APS_WARP_BICUBIC = 0, 1, 2
APS_TFORM_PROJECTIVE, APS_TFORM_AFFINE, APS_TFORM_SIMILARITY, APS_TFORM_RIGID, APS_TFORM_TRANSLATION = 0, 1, 2, 3, 4
APS_PROJ_CYLINDRICAL, APS_PROJ_SPHERICAL, APS_PROJ_PLANAR, APS_PROJ_STEREOGRAPHIC = 0, 1, 2, 3
APS_BLEND_NONE, APS_BLEND_LINEAR, APS_BLEND_MULTIBAND = 0, 1, 2
APS_NONE_LAST, APS_NONE_FIRST, APS_NONE_MAXANGLE = 0, 1, 2
APS_IMG_U8_HWC, APS_IMG_U8_MATLAB = 0, 1

_CODE_NAMES = {
    APS_E_ARG: "APS_E_ARG", APS_E_DIM: "APS_E_DIM", APS_E_TYPE: "APS_E_TYPE", APS_E_OOM: "APS_E_OOM",
    APS_E_DEVICE: "APS_E_DEVICE", APS_E_INTERNAL: "APS_E_INTERNAL", APS_E_CAP: "APS_E_CAP",
}


class ApsError(RuntimeError):
    """A non-zero status from libaps_hip.so; ``.code`` is the APS_E_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(f"{_CODE_NAMES.get(code, code)}: {message}")
        self.code = code
        self.message = message


class aps_match_opts(C.Structure):
    _fields_ = [("max_ratio", C.c_double), ("match_threshold", C.c_double), ("unique", C.c_int),
                ("normalize", C.c_int)]


class aps_ransac_opts(C.Structure):
    _fields_ = [("max_distance", C.c_double), ("confidence", C.c_double), ("max_iter", C.c_int),
                ("tform_type", C.c_int), ("method", C.c_int)]


class aps_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int), ("width", C.c_int), ("channels", C.c_int),
                ("layout", C.c_int), ("K", C.c_double * 9), ("R", C.c_double * 9),
                ("gain", C.c_float * 3)]


class aps_canvas(C.Structure):
    _fields_ = [("mode", C.c_int), ("height", C.c_int), ("width", C.c_int), ("f_pan", C.c_double),
                ("origin0", C.c_double), ("origin1", C.c_double), ("R_ref", C.c_double * 9)]


class aps_render_opts(C.Structure):
    _fields_ = [("tile_h", C.c_int), ("tile_w", C.c_int), ("angle_power", C.c_float),
                ("blending", C.c_int), ("pyr_levels", C.c_int), ("pyr_sigma", C.c_float),
                ("none_policy", C.c_int), ("canvas_white", C.c_int)]


class aps_sift_params(C.Structure):
    _fields_ = [("sigma", C.c_double), ("n_layers", C.c_int), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("max_features", C.c_int)]


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/aps.h one to one
_SIGNATURES = {
    "aps_version": [],
    "aps_last_error": [],
    "aps_device_count": [],
    "aps_set_device": [_i],
    "aps_set_thread_device": [_i],
    "aps_set_thread_stream_priority": [_i],
    "aps_get_device": [],
    "aps_set_stream": [_vp],
    "aps_synchronize": [],
    "aps_release_workspace": [],
    "aps_timer_begin": [],
    "aps_timer_end": [C.POINTER(_f)],
    "aps_profile_enable": [_i],
    "aps_profile_reset": [],
    "aps_profile_get": [C.c_char_p, C.POINTER(_d), C.POINTER(_i)],
    "aps_profile_names": [C.c_char_p, _i],
    "aps_profile_series": [C.c_char_p, C.POINTER(_d), _i, C.POINTER(_i)],
    "aps_match_2nn_ssd": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _vp, _vp, _vp],
    "aps_match_pca2nn": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "aps_match_features": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, C.POINTER(aps_match_opts), _vp,
                           _vp, _vp, _i64, C.POINTER(_i64)],
    "aps_match_pairwise": [C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), _i, _i, _i,
                           C.POINTER(aps_match_opts), _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64)],
    "aps_match_pairs": [C.POINTER(_vp), C.POINTER(_i64), C.POINTER(_i64), _i, _i, _i, _vp, _vp, _i64,
                        C.POINTER(aps_match_opts), _vp, _vp, _vp, _vp, _i64, C.POINTER(_i64)],
    "aps_knn_global": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64],
    "aps_knn_global_screened": [_vp, _i64, _i64, _i, _i, _vp, _i, _f, _i, _vp, _vp, _i64],
    "aps_knn_global_screen_stats": [C.POINTER(_i64), C.POINTER(_i64)],
    "aps_global_filter": [_vp, _vp, _i64, _i, _i64, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _i64,
                          C.POINTER(_i64)],
    "aps_hamming_2nn": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _vp, _vp, _vp],
    "aps_ransac_score": [_vp, _i, _vp, _vp, _i64, _i64, _d, _i, _vp, _vp, _vp],
    "aps_ransac_homography": [_vp, _vp, _i64, _i64, _vp, _i, C.POINTER(aps_ransac_opts), _vp, _vp,
                              C.POINTER(_i), C.POINTER(_i)],
    "aps_ransac_draw_samples": [_vp, _vp, _i, _i, C.c_uint64, _vp],
    "aps_ransac_homography_batch": [_vp, _vp, _i64, _vp, _i, _vp, _i, C.POINTER(aps_ransac_opts), _vp,
                                    _vp, _vp, _vp],
    "aps_render": [C.POINTER(aps_image), _i, C.POINTER(aps_canvas), C.POINTER(aps_render_opts), _i,
                   _vp, _vp],
    "aps_render_tiles": [C.POINTER(aps_image), _i, C.POINTER(aps_canvas), C.POINTER(aps_render_opts), _i,
                         _i, _i, _vp, _vp],
    "aps_render_tile_range": [C.POINTER(aps_image), _i, C.POINTER(aps_canvas), C.POINTER(aps_render_opts), _i,
                              _i, _i, _vp, _vp],
    "aps_warp_tile": [C.POINTER(aps_image), C.POINTER(aps_canvas), _i, _i, _i, _i, _f, _vp, _vp, _vp,
                      _vp],
    "aps_gain_overlap_stats": [C.POINTER(aps_image), _i, C.POINTER(aps_canvas), _i, _vp, _vp, _vp],
    "aps_gain_overlap_stats_warped": [_vp, _vp, _i, _i64, _i64, _i, _i, _i, _vp, _vp, _vp],
    "aps_imresize_u8": [_vp, _i, _i, _i, _i, _i, _i, _d, _d, _i, _vp],
    "aps_crop_rect": [_vp, _i64, _i64, _i, _i, _d, _vp, _vp],
    "aps_crop_nonzero_bbox": [_vp, _i64, _i64, _i, _i, _vp, _vp],
    "aps_ransac_draws_exhausted": [],
    "aps_gather_match_points": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64],
    "aps_match_screen_stats": [_vp, _vp],
    "aps_match_screen_exact_jobs": [_vp, _vp],
    "aps_match_screen_kernel_regs": [C.c_int, C.c_int, _vp, _vp],
    "aps_match_set_stats": [_vp, _i64, _i64, C.c_int, C.c_int, _vp],
    "aps_global_normalize": [_vp, _i64, _i64, _i, _i, _vp],
    "aps_knn_hamming": [_vp, _i64, _i64, _vp, _i64, _i64, _i, _i, _i, _vp, _vp, _i64],
    "aps_ba_pair_blocks": [_vp, _vp, _i64, _vp, _i, _vp, _d, _i, _vp],
    "aps_multiband_blend": [_vp, _vp, _i, _i, _i, _i, _f, _vp],
    "aps_linear_blend": [_vp, _vp, _i, _i, _i, _vp],
    "aps_image_warp_h_u8": [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, C.c_uint8, _vp],
    "aps_image_warp_h_f32": [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, _f, _vp],
    "aps_image_warp_u8": [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, C.c_uint8, _i, _vp],
    "aps_image_warp_f32": [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, _f, _i, _vp],
    "aps_synth_view": [_vp, _vp, _i, _i, C.c_uint, _f, _f, _vp],
    "aps_sift_extract": [_vp, _i, _i, _i, _i, C.POINTER(aps_sift_params), _vp, _i, _i64, _vp, _i64,
                         _vp, _i64, C.POINTER(_i64)],
}
_RESTYPES = {"aps_last_error": C.c_char_p}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). This package has no CPU or PyTorch fallback.")
    # One HIP runtime per process: the torch wheel bundles its own libamdhip64/libhsa-runtime64 with the
    # same SONAMEs as /opt/rocm's.  If ours were loaded first, torch would later bring up a second HSA
    # runtime and see no GPUs ("No HIP GPUs are available").  Importing torch first makes the dynamic
    # loader resolve libaps_hip.so's dependency to the copy torch already mapped.  (A MATLAB/mex host
    # has no torch and simply uses the system ROCm runtime.)
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover - torch is plumbing, not a requirement of the C ABI
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == the .so does not export what aps.h declares
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    return lib


lib = _load()


def check(status: int) -> None:
    if status != APS_OK:
        raise ApsError(status, (lib.aps_last_error() or b"").decode("utf-8", "replace"))


# ---- pointer helpers -------------------------------------------------------------------------------
def is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


def ptr(x) -> int:
    """Address of a numpy array (host) or torch tensor (host or device); None -> NULL."""
    if x is None:
        return None
    if is_torch(x):
        return x.data_ptr()
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    raise TypeError(f"expected numpy array or torch tensor, got {type(x)}")


def use_torch_stream() -> None:
    """Make the library enqueue on torch's current HIP stream (so torch ops and ours are ordered)."""
    import torch

    check(lib.aps_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def profile_enable(on=True) -> None:
    """True/1: every launch site; 2: only the per-batch sites (matching, RANSAC, coverage ...); False/0: off."""
    check(lib.aps_profile_enable(2 if on == 2 else (1 if on else 0)))


def profile_reset() -> None:
    check(lib.aps_profile_reset())


def profile_get(name: str):
    """(total_ms, launches) of the named kernel since the last reset."""
    t, n = C.c_double(0), C.c_int(0)
    check(lib.aps_profile_get(name.encode(), C.byref(t), C.byref(n)))
    return t.value, n.value


def profile_series(name: str, cap: int = 4096):
    """The recorded launches of the named kernel since the last reset, one duration (ms) each, in launch order."""
    buf, n = (C.c_double * cap)(), C.c_int(0)
    check(lib.aps_profile_series(name.encode(), buf, cap, C.byref(n)))
    return [buf[k] for k in range(min(n.value, cap))]


def profile_all():
    buf = C.create_string_buffer(4096)
    check(lib.aps_profile_names(buf, 4096))
    names = [s for s in buf.value.decode().split(";") if s]
    return {n: profile_get(n) for n in names}
