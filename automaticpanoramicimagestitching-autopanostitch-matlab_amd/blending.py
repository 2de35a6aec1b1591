"""Host-side mirror of PP/blending/ (multiBandBlending.m, linearBlending.m)."""
from __future__ import annotations

import numpy as np

from ._capi import check, lib, ptr


def _stack(Ci, Wi):
    K = len(Ci)
    if K < 1 or K != len(Wi):
        raise ValueError("Ci and Wi must be non-empty cell arrays of equal length")
    h, w = np.asarray(Ci[0]).shape[:2]
    C = np.zeros((K, h, w, 3), np.float32)
    W = np.zeros((K, h, w), np.float32)
    for k in range(K):
        c = np.asarray(Ci[k], np.float32)
        if c.ndim == 2:
            c = np.repeat(c[..., None], 3, axis=2)  # C0 == 1 -> repmat (multiBandBlending.m:93-95)
        ww = np.asarray(Wi[k], np.float32)
        if ww.ndim == 3:
            ww = ww[..., 0]
        if c.shape[:2] != (h, w) or ww.shape != (h, w):
            raise ValueError(f"Ci{{{k + 1}}} size differs from Ci{{1}}.")
        C[k], W[k] = c, ww
    return C, W, K, h, w


def multiBandBlending(Ci, Wi, levels, onGPU=True, sigma=1.0):
    """F = multiBandBlending(Ci, Wi, levels, onGPU, sigma) (multiBandBlending.m:1-178); F is h x w x 3 single
    in [0,1] (h x w when the inputs are single channel)."""
    if int(levels) != levels or levels < 1:
        raise ValueError("levels must be a positive integer")
    if sigma <= 0:
        raise ValueError("sigma must be positive")
    gray = np.asarray(Ci[0]).ndim == 2
    C, W, K, h, w = _stack(Ci, Wi)
    F = np.zeros((h, w, 3), np.float32)
    check(lib.aps_multiband_blend(ptr(C), ptr(W), K, h, w, int(levels), float(sigma), ptr(F)))
    return F[..., 0] if gray else F


def linearBlending(warpedImages, warpedWeights):
    """imageBlended = linearBlending(warpedImages, warpedWeights) (linearBlending.m:1-117); integer inputs are
    rounded and saturated back to their class (:104-112)."""
    if len(warpedImages) == 0:
        return None
    cls = np.asarray(warpedImages[0]).dtype
    gray = np.asarray(warpedImages[0]).ndim == 2
    C, W, K, h, w = _stack(warpedImages, warpedWeights)
    F = np.zeros((h, w, 3), np.float32)
    check(lib.aps_linear_blend(ptr(C), ptr(W), K, h, w, ptr(F)))
    if np.issubdtype(cls, np.integer):
        info = np.iinfo(cls)
        Fd = np.clip(F.astype(np.float64), info.min, info.max)
        F = (np.sign(Fd) * np.floor(np.abs(Fd) + 0.5)).astype(cls)
    else:
        F = F.astype(cls)
    return F[..., 0] if gray else F
