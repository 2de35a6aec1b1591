"""Host-side mirror of PP/renderPanorama/renderPanorama.m: option defaults, bounds and canvas sizing stay
on the host in float64 exactly as the reference computes them; the tile loop (rays, sampling, fusion,
multiband blending, paint) runs on the device through aps_render.

Cameras are dicts with 'K' and 'R' (3x3, world->camera), optionally 'noRotation' and 'H2refined'
(cameras struct, initializeCameraMatrices.m:114-122).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _capi
from ._capi import check, lib, ptr

_MODES = {"cylindrical": _capi.APS_PROJ_CYLINDRICAL, "spherical": _capi.APS_PROJ_SPHERICAL,
          "equirectangular": _capi.APS_PROJ_SPHERICAL, "planar": _capi.APS_PROJ_PLANAR,
          "perspective": _capi.APS_PROJ_PLANAR, "stereographic": _capi.APS_PROJ_STEREOGRAPHIC}
_BLEND = {"none": _capi.APS_BLEND_NONE, "linear": _capi.APS_BLEND_LINEAR, "multiband": _capi.APS_BLEND_MULTIBAND}
_POLICY = {"last": _capi.APS_NONE_LAST, "first": _capi.APS_NONE_FIRST, "maxangle": _capi.APS_NONE_MAXANGLE}


def default_opts(opts, cameras, refIdx):
    """renderPanorama.m:41-71."""
    o = dict(opts or {})
    o.setdefault("fPan", float(np.asarray(cameras[refIdx]["K"])[0, 0]))
    for k, v in (("resScale", 1.0), ("anglePower", 1), ("cropBorder", True), ("margin", 0.01),
                 ("tile", None), ("maxMegapixel", 50), ("robustPct", (1, 99)), ("uvAbsCap", 8.0),
                 ("pixelPad", 24), ("autoRef", True), ("canvasColor", "black"),
                 ("gainCompensation", True), ("blending", "multiband"), ("pyrLevels", 3),
                 ("pyrSigma", 1.0), ("composeNonePolicy", "last")):
        o.setdefault(k, v)
    return o


_GRID_CACHE = {}


def _grid_points(H, W, nx, ny, border):
    """The sample pixels of the bounds functions for one image size: 48 x 32 interior grid in MATLAB's U(:) order and,
    for the planar / stereographic bounds, 4 x `border` edge samples (renderPanorama.m:1524-1530,1612-1625)."""
    key = (int(H), int(W), nx, ny, border)
    xy1 = _GRID_CACHE.get(key)
    if xy1 is None:
        xs = np.linspace(1, W, nx)
        ys = np.linspace(1, H, ny)
        U, V = np.meshgrid(xs, ys)
        u = U.T.reshape(-1)  # MATLAB U(:) walks column-major
        v = V.T.reshape(-1)
        if border:
            xb = np.linspace(1, W, border)
            yb = np.linspace(1, H, border)
            u = np.concatenate([u, xb, xb, np.ones(border), W * np.ones(border)])
            v = np.concatenate([v, np.ones(border), H * np.ones(border), yb, yb])
        xy1 = np.stack([u, v, np.ones_like(u)])
        if len(_GRID_CACHE) > 64:
            _GRID_CACHE.clear()
        _GRID_CACHE[key] = xy1
    return xy1


_RAYC_CACHE = {}


def _solve_K(Ks, B):
    """K \\ B for a stack of intrinsic matrices (n x 3 x 3) and right-hand sides (n x 3 x m).  MATLAB's mldivide takes the
    triangular solver for an upper-triangular K (every pinhole K is): back substitution, which is also what the oracle
    does (oracle/bounds_oracle.c::solve_K) and a tenth of the cost of one LAPACK factorisation per camera.  Any other K
    goes through the general solver."""
    Ks = np.asarray(Ks, np.float64)
    if np.all(Ks[:, 1, 0] == 0) and np.all(Ks[:, 2, 0] == 0) and np.all(Ks[:, 2, 1] == 0):
        k = lambda r, c: Ks[:, r, c][:, None]  # noqa: E731
        x3 = B[:, 2] / k(2, 2)
        x2 = (B[:, 1] - k(1, 2) * x3) / k(1, 1)
        x1 = (B[:, 0] - k(0, 1) * x2 - k(0, 2) * x3) / k(0, 0)
        return np.stack([x1, x2, x3], axis=1)
    return np.linalg.solve(Ks, B)


def _grid_rays(cam, H, W, nx=48, ny=32, border=0):
    pts = _grid_points(H, W, nx, ny, border)
    rayC = _solve_K(np.asarray(cam["K"], np.float64)[None], pts[None])[0]
    return np.asarray(cam["R"], np.float64).T @ rayC


def _all_grid_rays(cams, imgSize, border=0, stacked=False):
    """_grid_rays of every camera, batched per image size (one LAPACK call and one matrix product for all cameras of
    a size; the per-camera results are the same numbers as the one-at-a-time form).  stacked: the per-size stacks
    (n_cameras_of_that_size x 3 x points) instead of one array per camera, for callers that only reduce over everything."""
    out = [None] * len(cams)
    stacks = []
    groups = {}
    for i in range(len(cams)):
        groups.setdefault((int(imgSize[i][0]), int(imgSize[i][1])), []).append(i)
    for (H, W), ids in groups.items():
        xy1 = _grid_points(H, W, 48, 32, border)
        Ks = np.stack([np.asarray(cams[i]["K"], np.float64) for i in ids])
        Rt = np.stack([np.asarray(cams[i]["R"], np.float64).T for i in ids])
        # the camera-frame rays depend on the intrinsics and the image size only: kept from call to call (a video or a
        # re-render with refined rotations solves nothing again)
        key = (Ks.tobytes(), H, W, border)
        rayC = _RAYC_CACHE.get(key)
        if rayC is None:
            if len(_RAYC_CACHE) >= 8:
                _RAYC_CACHE.clear()
            rayC = _RAYC_CACHE[key] = _solve_K(Ks, np.broadcast_to(xy1, (len(ids),) + xy1.shape))
        rays = Rt @ rayC
        stacks.append(rays)
        for q, i in enumerate(ids):
            out[i] = rays[q]
    return stacks if stacked else out


def sphericalBounds(cams, imgSize):
    """renderPanorama.m:1544-1579."""
    tmin = pmin = math.inf
    tmax = pmax = -math.inf

    def ext(r):
        x, y, z = r[:, 0], r[:, 1], r[:, 2]
        th = np.arctan2(x, z)
        ph = np.arctan2(y, np.hypot(x, z))
        return th.min(), th.max(), ph.min(), ph.max()

    # (Round 5: the same evaluation cut into camera chunks on four host threads is no faster - 0.88 against 0.73 ms for 64
    # cameras on the GPU box's host - so it stays one piece.)
    for rays in _all_grid_rays(cams, imgSize, stacked=True):  # (all cameras of a size at once: the same elementwise values)
        a, b, c, d = ext(rays)
        tmin, tmax = min(tmin, a), max(tmax, b)
        pmin, pmax = min(pmin, c), max(pmax, d)
    return tmin, tmax, pmin, pmax


def cylindricalBounds(cams, imgSize):
    """renderPanorama.m:1507-1542."""
    tmin = hmin = math.inf
    tmax = hmax = -math.inf
    for rays in _all_grid_rays(cams, imgSize, stacked=True):
        x, y, z = rays[:, 0], rays[:, 1], rays[:, 2]
        th = np.arctan2(x, z)
        hh = y / np.hypot(x, z)
        tmin, tmax = min(tmin, th.min()), max(tmax, th.max())
        hmin, hmax = min(hmin, hh.min()), max(hmax, hh.max())
    return tmin, tmax, hmin, hmax


def _prctile(x, p):
    """MATLAB prctile: linear interpolation with sample i at 100*(i-0.5)/n percent."""
    x = np.sort(np.asarray(x, np.float64))
    n = x.size
    pos = p / 100.0 * n + 0.5  # 1-based fractional rank
    pos = min(max(pos, 1.0), float(n))
    lo = int(math.floor(pos))
    hi = min(lo + 1, n)
    return x[lo - 1] + (pos - lo) * (x[hi - 1] - x[lo - 1])


def planarBounds(cams, imgSize, Rref, robustPct, uvAbsCap):
    """renderPanorama.m:1581-1665."""
    umin = vmin = math.inf
    umax = vmax = -math.inf
    for rays in _all_grid_rays(cams, imgSize, border=512):
        rayR = np.asarray(Rref, np.float64) @ rays
        zr = rayR[2]
        m = zr > 1e-4
        if not m.any():
            continue
        ur, vr = rayR[0, m] / zr[m], rayR[1, m] / zr[m]
        if np.isfinite(uvAbsCap) and uvAbsCap > 0:
            ur = np.clip(ur, -uvAbsCap, uvAbsCap)
            vr = np.clip(vr, -uvAbsCap, uvAbsCap)
        umin, umax = min(umin, _prctile(ur, robustPct[0])), max(umax, _prctile(ur, robustPct[1]))
        vmin, vmax = min(vmin, _prctile(vr, robustPct[0])), max(vmax, _prctile(vr, robustPct[1]))
    if not (np.isfinite(umin) and np.isfinite(umax)) or umin >= umax:
        umin, umax = -1.0, 1.0
    if not (np.isfinite(vmin) and np.isfinite(vmax)) or vmin >= vmax:
        vmin, vmax = -1.0, 1.0
    return umin, umax, vmin, vmax


def stereographicBounds(cams, imgSize, Rref, robustPct, absCap):
    """renderPanorama.m:1667-1754."""
    amin = bmin = math.inf
    amax = bmax = -math.inf
    for rays in _all_grid_rays(cams, imgSize, border=512):
        rayR = np.asarray(Rref, np.float64) @ rays
        nr = np.sqrt((rayR ** 2).sum(0))
        xr, yr, zr = rayR / nr
        den = 1 + zr
        valid = den > 1e-6
        if not valid.any():
            continue
        a, b = xr[valid] / den[valid], yr[valid] / den[valid]
        if np.isfinite(absCap) and absCap > 0:
            a = np.clip(a, -absCap, absCap)
            b = np.clip(b, -absCap, absCap)
        amin, amax = min(amin, _prctile(a, robustPct[0])), max(amax, _prctile(a, robustPct[1]))
        bmin, bmax = min(bmin, _prctile(b, robustPct[0])), max(bmax, _prctile(b, robustPct[1]))
    if not (np.isfinite(amin) and np.isfinite(amax)) or amin >= amax:
        amin, amax = -1.0, 1.0
    if not (np.isfinite(bmin) and np.isfinite(bmax)) or bmin >= bmax:
        bmin, bmax = -1.0, 1.0
    return amin, amax, bmin, bmax


def canvas_geometry(cameras, imgSize, mode, refIdx, opts):
    """Bounds + canvas size (renderPanorama.m:84-232).  Returns dict(mode, H, W, fPan, o0, o1, Rref, refIdx)."""
    mode = str(mode).lower()
    if mode not in _MODES:
        raise ValueError("mode must be cylindrical, spherical, or planar/perspective")
    o = opts
    f = float(o["fPan"])
    n = len(cameras)
    if mode in ("planar", "perspective", "stereographic") and o["autoRef"]:  # :84-122
        best, best_idx = math.inf, refIdx
        for ii in range(n):
            Rr = cameras[ii]["R"]
            if mode == "stereographic":
                a0, a1, b0, b1 = stereographicBounds(cameras, imgSize, Rr, o["robustPct"], o["uvAbsCap"])
                ext = max(abs(a0), abs(a1), abs(b0), abs(b1)) * (1 + 2 * o["margin"]) + o["pixelPad"] / f
                Wi = max(1, math.ceil(2 * f * ext * o["resScale"]))
                area = float(Wi) * Wi
            else:
                u0, u1, v0, v1 = planarBounds(cameras, imgSize, Rr, o["robustPct"], o["uvAbsCap"])
                du, dv = u1 - u0, v1 - v0
                u0, u1 = u0 - o["margin"] * du - o["pixelPad"] / f, u1 + o["margin"] * du + o["pixelPad"] / f
                v0, v1 = v0 - o["margin"] * dv - o["pixelPad"] / f, v1 + o["margin"] * dv + o["pixelPad"] / f
                area = float(max(1, math.ceil(f * (u1 - u0) * o["resScale"]))) * max(1, math.ceil(f * (v1 - v0) * o["resScale"]))
            if area < best:
                best, best_idx = area, ii
        refIdx = best_idx
    Rref = np.asarray(cameras[refIdx]["R"], np.float64)
    rs = o["resScale"]
    if mode == "cylindrical":
        a0, a1, b0, b1 = cylindricalBounds(cameras, imgSize)
    elif mode in ("spherical", "equirectangular"):
        a0, a1, b0, b1 = sphericalBounds(cameras, imgSize)
    elif mode in ("planar", "perspective"):
        a0, a1, b0, b1 = planarBounds(cameras, imgSize, Rref, o["robustPct"], o["uvAbsCap"])
    else:
        a0, a1, b0, b1 = stereographicBounds(cameras, imgSize, Rref, o["robustPct"], o["uvAbsCap"])
        ext = max(abs(a0), abs(a1), abs(b0), abs(b1))
        a0, a1, b0, b1 = -ext, ext, -ext, ext
    da, db = a1 - a0, b1 - b0
    a0, a1 = a0 - o["margin"] * da, a1 + o["margin"] * da
    b0, b1 = b0 - o["margin"] * db, b1 + o["margin"] * db
    if mode in ("planar", "perspective", "stereographic"):
        a0, a1 = a0 - o["pixelPad"] / f, a1 + o["pixelPad"] / f
        b0, b1 = b0 - o["pixelPad"] / f, b1 + o["pixelPad"] / f
    W = max(1, math.ceil(f * (a1 - a0) * rs))
    H = max(1, math.ceil(f * (b1 - b0) * rs))
    if mode in ("planar", "perspective", "stereographic"):  # global pixel cap (:170-177)
        maxPixel = round(o["maxMegapixel"] * 1e6)
        if float(H) * float(W) > maxPixel:
            s = math.sqrt(maxPixel / (float(H) * float(W)))
            rs = rs * s
            W = max(1, math.ceil(f * (a1 - a0) * rs))
            H = max(1, math.ceil(f * (b1 - b0) * rs))
    return {"mode": mode, "H": int(H), "W": int(W), "fPan": f, "o0": float(a0), "o1": float(b0),
            "Rref": Rref, "refIdx": refIdx}


def effective_tile(opts, geo):
    """The tile side renderPanorama uses for a canvas: opts['tile'] when given, else 2048 clamped to the canvas."""
    if opts.get("tile") is not None:
        return tuple(int(v) for v in opts["tile"])
    H, W = int(geo["H"]), int(geo["W"])
    side = max(512, min(2048, H, W)) if min(H, W) >= 512 else min(H, W)
    return (side, side)


def cropNonzeroBbox(panorama, canvasColor="black"):
    """[panoCropped, rect, didCrop] = cropNonzeroBbox(panorama, canvasColor) (renderPanorama.m:1459-1504): bounding box of
    rgb2gray(panorama) > 0 (or < 255 on a white canvas), 6 px pad; the box is a device reduction (aps_crop_nonzero_bbox),
    the crop itself a slice (a view of a resident panorama, no copy).  rect = (r1, r2, c1, c2), 1-based inclusive."""
    H, W = int(panorama.shape[0]), int(panorama.shape[1])
    if _capi.is_torch(panorama):
        src = panorama if panorama.is_contiguous() else panorama.contiguous()
    else:
        src = np.ascontiguousarray(panorama, np.uint8)
    if src.ndim == 2:
        src = src[..., None].repeat(3, axis=2) if not _capi.is_torch(src) else src[..., None].expand(H, W, 3).contiguous()
    rect = np.zeros(4, np.int64)
    did = C.c_int(0)
    check(lib.aps_crop_nonzero_bbox(ptr(src), H, W, _capi.APS_IMG_U8_HWC, 1 if str(canvasColor).lower() == "white" else 0,
                                    ptr(rect), C.byref(did)))
    r1, r2, c1, c2 = (int(v) for v in rect)
    if not did.value:
        return panorama, (1, H, 1, W), False
    return panorama[r1 - 1:r2, c1 - 1:c2], (r1, r2, c1, c2), True


def make_image_structs(images, cameras, gains=None):
    n = len(images)
    arr = (_capi.aps_image * n)()
    keep = []
    for i, img in enumerate(images):
        if _capi.is_torch(img):
            t = img.contiguous()
            h, w = t.shape[0], t.shape[1]
            c = 1 if t.dim() == 2 else t.shape[2]
        else:
            t = np.ascontiguousarray(img, np.uint8)
            h, w = t.shape[0], t.shape[1]
            c = 1 if t.ndim == 2 else t.shape[2]
        keep.append(t)
        a = arr[i]
        a.data = ptr(t)
        a.height, a.width, a.channels, a.layout = int(h), int(w), int(c), _capi.APS_IMG_U8_HWC
        K = np.asarray(cameras[i]["K"], np.float64)
        R = np.asarray(cameras[i]["R"], np.float64)
        for e in range(9):
            a.K[e] = K[e % 3, e // 3]
            a.R[e] = R[e % 3, e // 3]
        g = (1.0, 1.0, 1.0) if gains is None else gains[i]
        for ch in range(3):
            a.gain[ch] = float(g[ch])
    return arr, keep


def make_canvas_struct(geo):
    cv = _capi.aps_canvas()
    cv.mode = _MODES[geo["mode"]]
    cv.height, cv.width = geo["H"], geo["W"]
    cv.f_pan, cv.origin0, cv.origin1 = geo["fPan"], geo["o0"], geo["o1"]
    R = np.asarray(geo["Rref"], np.float64)
    for e in range(9):
        cv.R_ref[e] = R[e % 3, e // 3]
    return cv


def make_render_opts(opts):
    ro = _capi.aps_render_opts()
    tile = opts["tile"]
    ro.tile_h, ro.tile_w = int(tile[0]), int(tile[1])
    ro.angle_power = float(opts["anglePower"])
    ro.blending = _BLEND[str(opts["blending"]).lower()]
    ro.pyr_levels = int(opts["pyrLevels"])
    ro.pyr_sigma = float(opts["pyrSigma"])
    ro.none_policy = _POLICY[str(opts["composeNonePolicy"]).lower()]
    ro.canvas_white = 1 if str(opts["canvasColor"]).lower() == "white" else 0
    return ro


def warpWeights(images):
    """srcWeights = warpWeights(images, numImages) (renderPanorama.m:1282-1312): the separable tent map
    wy * wx of every image, single.  (The tiled render path never materialises it: the device samples the
    two 1-D tables; the planar-scan path below warps the map itself, as the reference does.)"""
    out = []
    for im in images:
        h, w = np.asarray(im).shape[:2]

        def tent(n):
            t = np.zeros(n, np.float32)
            a = (n + 1) // 2
            t[:a] = np.linspace(0, 1, a, dtype=np.float64).astype(np.float32) if a > 1 else 1.0
            b = n - n // 2
            t[n // 2:] = np.linspace(1, 0, b, dtype=np.float64).astype(np.float32) if b > 1 else 0.0
            return t

        out.append(np.outer(tent(h), tent(w)).astype(np.float32))
    return out


def _matlab_round(x):
    return float(np.sign(x) * np.floor(abs(x) + 0.5))


def pureNonRotationalImagesToCanvas(images, tforms, outputView, srcWeights, opts=None):
    """[Iw, Ww, xBoxes, yBoxes, centers] = pureNonRotationalImagesToCanvas(...) (renderPanorama.m:701-822):
    every image and its weight map warped to the common canvas with the device imageWarp (bilinear, fill 0,
    valid only where all four taps are inside, imageWarp.m:125-168); weights clamped to [0,1]."""
    from .imageProcessing import imageWarp, transformPointsForwardScratch

    Iw, Ww, xB, yB = [], [], [], []
    centers = np.zeros((len(images), 2))
    Hc, Wc = outputView["ImageSize"]
    for k, im in enumerate(images):
        Ik = np.asarray(im)
        r, c = Ik.shape[:2]
        corners = np.array([[1, 1], [c, 1], [c, r], [1, r], [1, 1]], np.float64)  # (x, y) of :751
        q = transformPointsForwardScratch(tforms[k], corners)
        xB.append(q[:, 0] - outputView["XWorldLimits"][0])
        yB.append(q[:, 1] - outputView["YWorldLimits"][0])
        centers[k] = (xB[-1].mean(), yB[-1].mean())
        Ik = Ik.astype(np.float32) / 255.0 if Ik.dtype == np.uint8 else Ik.astype(np.float32)
        Iwk = imageWarp(Ik, tforms[k], outputView)[:Hc, :Wc]
        Ws = np.ones((r, c), np.float32) if not srcWeights else np.asarray(srcWeights[k], np.float32)
        Wk = imageWarp(Ws, tforms[k], outputView)[:Hc, :Wc]
        Iw.append(Iwk if Iwk.ndim == 3 else Iwk[..., None])
        Ww.append(np.clip(Wk, 0.0, 1.0).astype(np.float32))
    return Iw, Ww, xB, yB, centers


def pureNonRotationalPanoramas(images, cameras, numImages, opts, gains=None):
    """[panorama, rgbAnnotation] = pureNonRotationalPanoramas(images, cameras, numImages, opts)
    (renderPanorama.m:519-699): planar-scan compositing.  Canvas = bounding box of all H2refined corner
    maps with MATLAB-rounded size (:547-575); every image warped to the FULL canvas; then whole-canvas
    'none' (winner-take-all by weight, first maximum) / 'linear' / 'multiband'; void pixels painted; uint8.
    Gains: pass `gains` (N x 3), or set opts['gainCompensation'] to have gainCompensationH run on the warped canvases
    (:584-591: overlap statistics on the device, solve on the host); otherwise ones."""
    from .blending import linearBlending, multiBandBlending
    from .imageProcessing import imref2dScratch, outputLimitsScratch

    o = {"blending": "multiband", "pyrLevels": 3, "pyrSigma": 1.0, "canvasColor": "black",
         "sigmaN": 10.0, "sigmag": 0.1}  # (renderPanorama.m:56-58: the defaults opts carries into gainCompensationH)
    o.update(opts or {})
    tforms = [np.asarray(cam["H2refined"], np.float64) for cam in cameras[:numImages]]
    lims = [outputLimitsScratch(T, (1, np.asarray(im).shape[1]), (1, np.asarray(im).shape[0]))
            for T, im in zip(tforms, images)]
    xMin, xMax = min(l[0][0] for l in lims), max(l[0][1] for l in lims)
    yMin, yMax = min(l[1][0] for l in lims), max(l[1][1] for l in lims)
    width, height = int(_matlab_round(xMax - xMin)), int(_matlab_round(yMax - yMin))
    if width < 1 or height < 1:
        raise ValueError("degenerate planar canvas")
    view = imref2dScratch((height, width), (xMin, xMax), (yMin, yMax))
    Iw, Ww, _, _, _ = pureNonRotationalImagesToCanvas(images, tforms, view, warpWeights(images), o)
    if gains is None and o.get("gainCompensation"):
        # renderPanorama.m:584-591: gains from the warped canvases (statistics on the device, N x N solve on the host).  As
        # in the ray renderer below only when the caller asks for it explicitly; otherwise ones (or the caller's own).
        from .gainCompensation import gainCompensationH

        gains = gainCompensationH(Iw, Ww, o)
    if gains is not None:
        Iw = [I * np.asarray(g, np.float32).reshape(1, 1, -1) for I, g in zip(Iw, gains)]
    mode = str(o["blending"]).lower()
    if mode == "none":
        Wst = np.stack(Ww, 2)
        idx = np.argmax(Wst, 2)  # first maximum, like MATLAB's max
        pano = np.take_along_axis(np.stack(Iw, 3), idx[:, :, None, None], 3)[..., 0]
    elif mode == "linear":
        pano = linearBlending(Iw, Ww)
    elif mode == "multiband":
        pano = multiBandBlending(Iw, Ww, int(o["pyrLevels"]), True, float(o["pyrSigma"]))
    else:
        raise ValueError("Wrong blening mode.")
    pano = np.array(pano, np.float32, copy=True)
    void = ~(np.stack(Ww, 2) > 0).any(2)
    pano[void] = 1.0 if str(o["canvasColor"]).lower() == "white" else 0.0
    v = 255.0 * pano.astype(np.float64)
    out = np.sign(v) * np.floor(np.abs(v) + 0.5)  # MATLAB round
    return np.clip(out, 0, 255).astype(np.uint8), None


def renderPanorama(input, images, imgSize, cameras, mode, refIdx, opts=None, gains=None,
                   return_covered=False, device_out=False, tile_subset=None, geo=None, zero_rest=True):
    """[panorama, rgbAnnotation] = renderPanorama(input, images, imgSize, cameras, mode, refIdx, opts)
    (renderPanorama.m:1-500).  refIdx is 0-based here.  Differences that are deliberate:
      * opts['tile'] must be explicit; the reference derives it from free GPU/CPU memory (:269-298),
        which makes its multiband output machine dependent.  Default here: [2048 2048] clamped to the
        canvas — the value the reference's auto-tiler reaches whenever memory is plentiful.
      * gains: pass `gains` (N x 3), or set opts['gainCompensation'] explicitly to have gainCompensationRKf run
        (statistics on the device, solve on the host); otherwise ones.
      * annotations (insertShape/insertText) are not produced: rgbAnnotation is always None."""
    if cameras and (cameras[0].get("noRotation", 0) == 1 or input.get("forcePlanarScan", False)):
        # renderPanorama.m:78-90: planar scans bypass the tiled ray renderer
        pano, ann = pureNonRotationalPanoramas(images, cameras, len(images), opts or {}, gains)
        return (pano, ann, None, None) if return_covered else (pano, ann)
    o = default_opts(opts, cameras, refIdx)
    imgSize = [tuple(int(v) for v in s) for s in imgSize]
    if geo is None:  # (a caller that has sized the canvas already - the sharded driver - passes it in)
        geo = canvas_geometry(cameras, imgSize, mode, refIdx, o)
    o["tile"] = effective_tile(o, geo)
    if gains is None and opts and opts.get("gainCompensation"):
        # renderPanorama.m:303-330: overlap statistics on the device, N x N solve on the host.  Only when the
        # caller asks for it explicitly; otherwise gains are ones (or the caller's own).
        from .gainCompensation import gainCompensationRKf

        gains = gainCompensationRKf(images, cameras, mode, refIdx, opts, geo)
    arr, keep = make_image_structs(images, cameras, gains)
    cv = make_canvas_struct(geo)
    ro = make_render_opts(o)
    H, W = geo["H"], geo["W"]
    if device_out:
        import torch

        pano = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        cov = torch.empty((H, W), dtype=torch.uint8, device="cuda")
    else:
        pano = np.zeros((H, W, 3), np.uint8)
        cov = np.zeros((H, W), np.uint8)
    # ONE call renders all tiles (render_batch.hip: multiband level-major, 'linear' / 'none' as one fused launch); the
    # per-tile path (APS_RENDER_LEGACY=1) still gains from a few host threads driving interleaved tile subsets
    import os as _os

    batched = not _os.environ.get("APS_RENDER_LEGACY")  # (since round 4 'linear' and 'none' are one fused launch as well)
    workers = _render_workers() if (device_out and not batched) else 1
    if device_out and batched and _os.environ.get("APS_RENDER_BATCH_WORKERS"):
        # Experiment (round 4): the batched call split over host threads / streams (tiles t % workers == k each), so that one
        # share's vector-bound warp runs beside another's latency-bound pyramid kernels.  The render alone: 18.2 -> 17.2 ms with
        # two shares for the 66 tiles of the 64 x 4K scene (both prepare all images), 18.4-19.5 with three; inside the pipeline
        # step nothing is left of it (19.2 against 19.4 ms), so one call stays the default.  Same bytes either way.
        n_tiles = -(-H // int(o["tile"][0])) * -(-W // int(o["tile"][1]))
        workers = max(1, min(int(_os.environ["APS_RENDER_BATCH_WORKERS"]), n_tiles))
    if tile_subset is None and workers <= 1:
        check(lib.aps_render(arr, len(images), C.byref(cv), C.byref(ro), _capi.APS_IMG_U8_HWC, ptr(pano), ptr(cov)))
    elif tile_subset is not None and tile_subset[0] == "range":
        # ("range", begin, end): the contiguous tiles begin <= t < end - a rank's band of the canvas (parallel.tile_ranges)
        import torch

        # zero_rest=False (a resident shard whose tiles alone are sent on: parallel.gather_tiles_to_root): the ~1 GB fill of
        # the canvas outside the run is skipped - those pixels are never read
        if device_out and zero_rest:
            pano.zero_()
            cov.zero_()
            torch.cuda.current_stream().synchronize()  # torch's fills must land before the library's stream paints tiles
        check(lib.aps_render_tile_range(arr, len(images), C.byref(cv), C.byref(ro), _capi.APS_IMG_U8_HWC,
                                        int(tile_subset[1]), int(tile_subset[2]), ptr(pano), ptr(cov)))
    else:  # (first, step): only tiles t with t % step == first — an interleaved shard of the tile loop
        first, step = (0, 1) if tile_subset is None else (int(tile_subset[0]), int(tile_subset[1]))
        if device_out and tile_subset is not None:
            import torch

            pano.zero_()
            cov.zero_()
            torch.cuda.current_stream().synchronize()  # torch's fills must land before the library's stream paints tiles
        if workers <= 1:
            check(lib.aps_render_tiles(arr, len(images), C.byref(cv), C.byref(ro), _capi.APS_IMG_U8_HWC,
                                       first, step, ptr(pano), ptr(cov)))
        else:
            # The tile loop of renderPanorama.m:342-406 is a parfor candidate: tiles are independent and paint
            # disjoint canvas rectangles.  A few host threads each drive their own stream over an interleaved
            # share of this rank's tiles (t % (step*workers) == first + step*k), so the many small pyramid
            # launches of one tile overlap with another tile's; every tile's arithmetic is unchanged.
            here = lib.aps_get_device()

            def work(k):
                check(lib.aps_set_thread_device(here))  # not the process-wide default another thread may have changed
                check(lib.aps_render_tiles(arr, len(images), C.byref(cv), C.byref(ro), _capi.APS_IMG_U8_HWC,
                                           first + step * k, step * workers, ptr(pano), ptr(cov)))
                check(lib.aps_synchronize())  # this thread's stream

            list(_render_pool(workers).map(work, range(workers)))
    del keep
    if o["cropBorder"] and tile_subset is None:  # renderPanorama.m:430-432 (a tile shard is cropped after it is combined)
        if device_out:
            check(lib.aps_synchronize())  # the panorama was painted on the library's stream of this thread
        pano, _, _ = cropNonzeroBbox(pano, o["canvasColor"])
    if return_covered:
        return pano, None, cov, geo
    return pano, None


_RENDER_POOL = None


def _render_workers():
    import os

    return max(1, int(os.environ.get("APS_RENDER_WORKERS", "2")))


def _render_pool(workers):
    global _RENDER_POOL
    from concurrent.futures import ThreadPoolExecutor

    if _RENDER_POOL is None or _RENDER_POOL._max_workers < workers:
        _RENDER_POOL = ThreadPoolExecutor(max_workers=workers)
    return _RENDER_POOL


def warp_tile(image, camera, geo, r0, c0, ht, wt, anglePower=2.0, gain=(1.0, 1.0, 1.0)):
    """sampleOneTile for one tile/image (renderPanorama.m:1063-1146): returns S, M, Wang, Wf."""
    arr, keep = make_image_structs([image], [camera], [gain])
    cv = make_canvas_struct(geo)
    S = np.zeros((ht, wt, 3), np.float32)
    M = np.zeros((ht, wt), np.uint8)
    Wa = np.zeros((ht, wt), np.float32)
    Wf = np.zeros((ht, wt), np.float32)
    check(lib.aps_warp_tile(arr, C.byref(cv), r0, c0, ht, wt, float(anglePower), ptr(S), ptr(M), ptr(Wa), ptr(Wf)))
    return S, M.astype(bool), Wa, Wf
