"""Host-side mirror of PP/imageProcessing/imageWarp.m and its imref2d helpers."""
from __future__ import annotations

import numpy as np

from ._capi import check, lib, ptr


def imref2dScratch(imageSize, xWorldLimits=None, yWorldLimits=None):
    """imref2dScratch.m:50-76 — a struct clone of imref2d."""
    h, w = int(imageSize[0]), int(imageSize[1])
    xl = (0.5, w + 0.5) if xWorldLimits is None else tuple(map(float, xWorldLimits))
    yl = (0.5, h + 0.5) if yWorldLimits is None else tuple(map(float, yWorldLimits))
    return {"ImageSize": (h, w), "XWorldLimits": xl, "YWorldLimits": yl,
            "PixelExtentInWorldX": (xl[1] - xl[0]) / w, "PixelExtentInWorldY": (yl[1] - yl[0]) / h}


def transformPointsForwardScratch(tform, xy):
    """transformPointsForwardScratch.m:57-77: [x y 1] -> H*[x;y;1], |w| < 1e-12 -> NaN."""
    H = np.asarray(tform, np.float64)
    xy = np.asarray(xy, np.float64)
    q = np.c_[xy, np.ones(len(xy))] @ H.T
    w = q[:, 2].copy()
    w[np.abs(w) < 1e-12] = np.nan
    return q[:, :2] / w[:, None]


def outputLimitsScratch(tform, xLimitsIn, yLimitsIn):
    """outputLimitsScratch.m:71-111: bounding box of the transformed input rectangle (corners + edge midpoints)."""
    x0, x1 = map(float, xLimitsIn)
    y0, y1 = map(float, yLimitsIn)
    xm, ym = 0.5 * (x0 + x1), 0.5 * (y0 + y1)
    pts = np.array([[x0, y0], [xm, y0], [x1, y0], [x0, ym], [xm, ym], [x1, ym], [x0, y1], [xm, y1], [x1, y1]])
    out = transformPointsForwardScratch(tform, pts)
    return (np.nanmin(out[:, 0]), np.nanmax(out[:, 0])), (np.nanmin(out[:, 1]), np.nanmax(out[:, 1]))


def imageWarp(image, tform, outputView, method="bilinear", fillValue=0):
    """warped = imageWarp(image, tform, outputView, options) (imageWarp.m:1-273), 'bilinear' on the device."""
    if str(method).lower() != "bilinear":
        raise NotImplementedError("only the reference's default 'bilinear' method is built on the device")
    img = np.asarray(image)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[..., None]
    H = np.ascontiguousarray(np.asarray(tform, np.float64).T)  # column-major 3x3
    oh, ow = outputView["ImageSize"]
    x0, y0 = outputView["XWorldLimits"][0], outputView["YWorldLimits"][0]
    sx, sy = outputView["PixelExtentInWorldX"], outputView["PixelExtentInWorldY"]
    c = img.shape[2]
    if img.dtype == np.uint8:
        src = np.ascontiguousarray(img)
        out = np.zeros((oh, ow, c), np.uint8)
        check(lib.aps_image_warp_h_u8(ptr(src), img.shape[0], img.shape[1], c, ptr(H), oh, ow, x0, y0, sx, sy,
                                      int(fillValue), ptr(out)))
    else:
        src = np.ascontiguousarray(img, np.float32)
        out = np.zeros((oh, ow, c), np.float32)
        check(lib.aps_image_warp_h_f32(ptr(src), img.shape[0], img.shape[1], c, ptr(H), oh, ow, x0, y0, sx, sy,
                                       float(fillValue), ptr(out)))
        out = out.astype(img.dtype) if img.dtype != np.float32 else out
    return out[..., 0] if squeeze else out
