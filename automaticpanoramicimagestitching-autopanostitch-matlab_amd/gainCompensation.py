"""Gain compensation (PP/gainCompensation/gainCompensationRKf.m, gainCompensationH.m) -- SURVEY 8(f) rank 1, the component
that sits between RANSAC/cameras and the render on every run: the O(pixels x N) overlap statistics run on the device
(gainCompensationRKf: with the render's own ray -> project -> bilinear sampler; gainCompensationH, the planar-scan form:
strided sums over the already warped canvases), the N x N solve per channel stays on the host (north star)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr
from .renderPanorama import make_canvas_struct, make_image_structs


def gain_overlap_stats(images, cameras, geo, stride=5):
    """processOneTile summed over all tiles (gainCompensationRKf.m:126-149,239-367): for every stride-th canvas point
    (1-based, :106-107) and every image pair i < j covering it: Nij += 1, sumCi += Ci, sumCj += Cj (raw 0..255
    bilinear samples).  geo: the canvas dict of renderPanorama.canvas_geometry.
    Returns (Nij [N,N], sumCi [N,N,3], sumCj [N,N,3]) float64, upper triangle."""
    n = len(images)
    arr, keep = make_image_structs(images, cameras)
    cv = make_canvas_struct(geo)
    Nij = np.zeros((n, n), np.float64, order="F")
    sCi = np.zeros((n, n, 3), np.float64, order="F")
    sCj = np.zeros((n, n, 3), np.float64, order="F")
    check(lib.aps_gain_overlap_stats(arr, n, C.byref(cv), int(stride), ptr(Nij), ptr(sCi), ptr(sCj)))
    del keep
    return np.ascontiguousarray(Nij), np.ascontiguousarray(sCi), np.ascontiguousarray(sCj)


def solve_gains(Nij, sumCi, sumCj, opts=None, refIdx=0):
    """The host part of gainCompensationRKf (:151-238): Brown-Lowe normal equations per channel from the pair
    means, prior sigma_g around 1, optional anchor on refIdx (0-based here); gains clamped to [0.25, 4]."""
    o = {"minOverlapSamples": 50, "sigmaN": 10.0, "sigmag": 0.1, "lambdaDiag": 1e-8, "anchorRef": False}
    o.update(opts or {})
    N = Nij.shape[0]
    gains = np.ones((N, 3), np.float32)
    minOv = max(1, int(o["minOverlapSamples"]))
    sN2, sg2 = float(o["sigmaN"]) ** 2, float(o["sigmag"]) ** 2
    edges = [(i, j) for i in range(N - 1) for j in range(i + 1, N) if Nij[i, j] >= minOv]
    if not edges:
        return gains
    A = np.zeros((3, N, N), np.float64)
    b = np.zeros(N, np.float64)
    for i, j in edges:
        K = float(Nij[i, j])
        Ii, Ij = sumCi[i, j] / K, sumCj[i, j] / K
        wN, wG = K / sN2, K / sg2
        for ch in range(3):
            A[ch, i, i] += wN * (Ii[ch] * Ii[ch]) + wG
            A[ch, j, j] += wN * (Ij[ch] * Ij[ch]) + wG
            A[ch, i, j] += -wN * (Ii[ch] * Ij[ch])
            A[ch, j, i] += -wN * (Ii[ch] * Ij[ch])
        b[i] += wG
        b[j] += wG
    for ch in range(3):
        A[ch] += float(o["lambdaDiag"]) * np.eye(N)
    if o["anchorRef"]:
        pin = max(0, min(N - 1, int(refIdx)))
        for ch in range(3):
            A[ch, pin, :] = 0
            A[ch, :, pin] = 0
            A[ch, pin, pin] = 1e6
        b[pin] = 1e6
    for ch in range(3):
        x = np.linalg.solve(A[ch], b)
        gains[:, ch] = np.clip(x, 0.25, 4.0).astype(np.float32)
    return gains


def gainCompensationRKf(images, cameras, mode, refIdx, opts, geo):
    """gains = gainCompensationRKf(images, cameras, mode, refIdx, opts, H, W, u0, v0, th0, h0, ph0, srcW)
    (gainCompensationRKf.m:1-238); the canvas scalars travel in `geo` (renderPanorama.canvas_geometry) and the
    tent maps are rebuilt on the device from the image sizes.  refIdx is 0-based.  Returns N x 3 float32."""
    o = dict(opts or {})
    stride = max(1, int(o.get("overlapStride", 5)))
    g = dict(geo)
    g["mode"] = mode
    Nij, sCi, sCj = gain_overlap_stats(images, cameras, g, stride)
    return solve_gains(Nij, sCi, sCj, o, refIdx)


def gain_overlap_stats_warped(Iw, Ww, ds=4):
    """The accumulation of gainCompensationH.m:45-52,78-149 on the device (aps_gain_overlap_stats_warped): Iw[k] H x W x 3
    (or H x W x 1 / H x W) float32 canvases, Ww[k] H x W float32 weight maps - numpy arrays or resident CUDA tensors, all
    of one size.  Every ds-th row and column; valid = Ww > 0 & finite; returns (Nij [N,N], sumCi [N,N,3], sumCj [N,N,3])
    float64, upper triangle."""
    n = len(Iw)
    if n == 0 or len(Ww) != n:
        raise ValueError("Iw and Ww must be non-empty lists of equal length")
    is_t = _capi.is_torch(Iw[0])
    if is_t:
        import torch

        Iw = [a.contiguous().float() for a in Iw]
        Ww = [a.contiguous().float() for a in Ww]
        if Iw[0].is_cuda:
            torch.cuda.current_stream().synchronize()  # torch produced them; the library reads on its own stream
    else:
        Iw = [np.ascontiguousarray(np.asarray(a, np.float32)) for a in Iw]
        Ww = [np.ascontiguousarray(np.asarray(a, np.float32)) for a in Ww]
    h, w = (int(v) for v in Iw[0].shape[:2])
    ch = int(Iw[0].shape[2]) if Iw[0].ndim == 3 else 1
    for a, b in zip(Iw, Ww):
        if tuple(a.shape[:2]) != (h, w) or tuple(b.shape) != (h, w) or (a.ndim == 3 and int(a.shape[2]) != ch):
            raise ValueError("every warped image and weight map must have the canvas size of the first")
    pi = (C.c_void_p * n)(*[ptr(a) for a in Iw])
    pw = (C.c_void_p * n)(*[ptr(a) for a in Ww])
    Nij = np.zeros((n, n), np.float64, order="F")
    sCi = np.zeros((n, n, 3), np.float64, order="F")
    sCj = np.zeros((n, n, 3), np.float64, order="F")
    check(lib.aps_gain_overlap_stats_warped(C.addressof(pi), C.addressof(pw), n, h, w, ch, _capi.APS_ROWMAJOR, int(ds), ptr(Nij),
                                            ptr(sCi), ptr(sCj)))
    return np.ascontiguousarray(Nij), np.ascontiguousarray(sCi), np.ascontiguousarray(sCj)


def solve_gains_H(Nij, sumCi, sumCj, opts=None):
    """The host part of gainCompensationH (:152-223).  It differs from gainCompensationRKf's solve: the prior weight is
    1 / sigmag^2 per PARTICIPATING image, added once (not scaled by the overlap size), b(i) = that weight, and the anchor
    (`anchorRef`, 1-based `refIdx` as in the reference) overwrites row and column `pin`.  Defaults :27-33."""
    o = {"minOverlapSamples": 100, "sigmaN": 10.0, "sigmag": 10.0, "lambdaDiag": 1e-8, "anchorRef": False, "refIdx": 1}
    o.update({k: v for k, v in (opts or {}).items() if k in o})
    N = Nij.shape[0]
    gains = np.ones((N, 3), np.float32)
    if N <= 1:
        return gains
    edges = [(i, j) for j in range(N) for i in range(j) if Nij[i, j] >= o["minOverlapSamples"]]  # find() order: column-major
    if not edges:
        return gains
    sN2, sg2 = float(o["sigmaN"]) ** 2, float(o["sigmag"]) ** 2
    A = np.zeros((3, N, N), np.float64)
    b = np.zeros(N, np.float64)
    for i, j in edges:
        K = float(Nij[i, j])
        Ii, Ij = sumCi[i, j] / K, sumCj[i, j] / K
        wN = K / sN2
        for ch in range(3):
            A[ch, i, i] += wN * (Ii[ch] ** 2)
            A[ch, j, j] += wN * (Ij[ch] ** 2)
            A[ch, i, j] += -wN * (Ii[ch] * Ij[ch])
            A[ch, j, i] += -wN * (Ii[ch] * Ij[ch])
    wG = 1.0 / sg2
    for i in sorted({v for e in edges for v in e}):
        A[:, i, i] += wG
        b[i] = wG
    for ch in range(3):
        A[ch] += float(o["lambdaDiag"]) * np.eye(N)
    if o["anchorRef"]:
        pin = max(1, min(N, int(o["refIdx"]))) - 1
        for ch in range(3):
            A[ch, pin, :] = 0
            A[ch, :, pin] = 0
            A[ch, pin, pin] = 1e6
        b[pin] = 1e6
    for ch in range(3):
        x = np.linalg.solve(A[ch], b)
        gains[:, ch] = np.clip(x, 0.25, 4.0).astype(np.float32)
    return gains


def gainCompensationH(Iw, Ww, opts=None):
    """gains = gainCompensationH(Iw, Ww, opts) (gainCompensationH.m:1-223): N x 3 float32 gains in [0.25, 4] from images
    already warped to a common canvas.  The overlap accumulation runs on the device, the N x N systems on the host."""
    o = dict(opts or {})
    if len(Iw) <= 1:
        return np.ones((len(Iw), 3), np.float32)
    Nij, sCi, sCj = gain_overlap_stats_warped(Iw, Ww, max(1, int(o.get("overlapDownsample", 4))))
    return solve_gains_H(Nij, sCi, sCj, o)
