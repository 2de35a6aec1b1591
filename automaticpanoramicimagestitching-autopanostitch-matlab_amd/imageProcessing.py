"""Host-side mirror of PP/imageProcessing/imageWarp.m and its imref2d helpers."""
from __future__ import annotations

import numpy as np

from . import _capi
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
    """warped = imageWarp(image, tform, outputView, options) (imageWarp.m:1-273) on the device: options.method
    'nearest' (:109-123), 'bilinear' (:125-168, the default) or 'bicubic' (:170-264); uint8 images stay uint8, every
    other class is interpolated as single."""
    m = {"nearest": _capi.APS_WARP_NEAREST, "bilinear": _capi.APS_WARP_BILINEAR, "bicubic": _capi.APS_WARP_BICUBIC}.get(str(method).lower())
    if m is None:
        raise ValueError(f"unknown interpolation method '{method}'")  # (the reference's switch silently leaves the fill value)
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
        check(lib.aps_image_warp_u8(ptr(src), img.shape[0], img.shape[1], c, ptr(H), oh, ow, x0, y0, sx, sy,
                                    int(fillValue), m, ptr(out)))
    else:
        src = np.ascontiguousarray(img, np.float32)
        out = np.zeros((oh, ow, c), np.float32)
        check(lib.aps_image_warp_f32(ptr(src), img.shape[0], img.shape[1], c, ptr(H), oh, ow, x0, y0, sx, sy,
                                     float(fillValue), m, ptr(out)))
        out = out.astype(img.dtype) if img.dtype != np.float32 else out
    return out[..., 0] if squeeze else out


def imresize(I, scale_or_size, method="bicubic"):
    """J = imresize(I, s, method) / imresize(I, [oh ow], method) for uint8 images on the device (toolbox semantics
    as restated in oracle/render_oracle.c).  The scalar form gives ceil(s * size) with scale s in both dimensions.
    A torch CUDA tensor stays resident: the result is a CUDA tensor and nothing visits the host."""
    import math

    dev = _capi.is_torch(I) and I.is_cuda
    if dev:
        import torch

        a = I.contiguous()
        if a.dtype != torch.uint8:
            raise TypeError("uint8 images only")
        sq = a.dim() == 2
        if sq:
            a = a[..., None]
        h, w, c = (int(v) for v in a.shape)
    else:
        a = np.ascontiguousarray(I)
        if a.dtype != np.uint8:
            raise TypeError("uint8 images only")
        sq = a.ndim == 2
        if sq:
            a = a[..., None]
        h, w, c = a.shape
    if np.isscalar(scale_or_size):
        s = float(scale_or_size)
        oh, ow, sr, sc = int(math.ceil(h * s)), int(math.ceil(w * s)), s, s
    else:
        oh, ow = int(scale_or_size[0]), int(scale_or_size[1])
        sr, sc = oh / h, ow / w
    m = {"bicubic": _capi.APS_RESIZE_BICUBIC, "bilinear": _capi.APS_RESIZE_BILINEAR}[str(method).lower()]
    if dev:
        out = torch.empty((oh, ow, c), dtype=torch.uint8, device=a.device)
        torch.cuda.current_stream().synchronize()  # the input came from torch's stream; the library runs on its own
    else:
        out = np.zeros((oh, ow, c), np.uint8)
    check(lib.aps_imresize_u8(ptr(a), h, w, c, _capi.APS_IMG_U8_HWC, oh, ow, sr, sc, m, ptr(out)))
    return out[..., 0] if sq else out


def _center_crop(J, Ht, Wt):
    h, w = J.shape[:2]
    r0, c0 = (h - Ht) // 2, (w - Wt) // 2
    return J[r0:r0 + Ht, c0:c0 + Wt]


def resizeImagesToLimits(imageFiles, heightLimit, widthLimit, mode="fit"):
    """imageFilesResized = resizeImagesToLimits(imageFiles, heightLimit, widthLimit, mode)
    (resizeImagesToLimits.m:1-160): 'fit' (isotropic shrink to the box, then all images to the common largest size),
    'pad' (fit, then replicate-pad to the box) or 'fillcrop' (cover, then centre crop).  Resizing runs on the device."""
    mode = str(mode).lower()
    if mode not in ("fit", "pad", "fillcrop"):
        raise ValueError(f"unknown mode '{mode}'")
    stage1 = []
    for I in imageFiles:
        I = np.asarray(I)
        if I.size == 0:
            stage1.append(I)
            continue
        h, w = I.shape[:2]
        if mode in ("fit", "pad"):
            s = min(heightLimit / h, widthLimit / w)
            if not np.isfinite(s) or s <= 0:
                s = 1
            J = imresize(I, s, "bicubic") if s < 1 else I
            if mode == "pad":
                if J.shape[0] > heightLimit or J.shape[1] > widthLimit:
                    J = _center_crop(J, min(heightLimit, J.shape[0]), min(widthLimit, J.shape[1]))
                ph, pw = heightLimit - J.shape[0], widthLimit - J.shape[1]
                pad = ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)) + (((0, 0),) if J.ndim == 3 else ())
                J = np.pad(J, pad, mode="edge")
        else:
            s = max(heightLimit / h, widthLimit / w)
            if not np.isfinite(s) or s <= 0:
                s = 1
            J = _center_crop(imresize(I, s, "bicubic"), heightLimit, widthLimit)
        stage1.append(J)
    if mode != "fit":
        return stage1
    sizes = {(J.shape[0], J.shape[1]) for J in stage1 if J.size}
    if len(sizes) <= 1:
        return stage1
    Hmax, Wmax = max(s[0] for s in sizes), max(s[1] for s in sizes)
    return [J if J.size == 0 else imresize(J, (Hmax, Wmax), "bicubic") for J in stage1]


def cropRectangle(stitchedImage, canvasColor="black", blackRange=0, whiteRange=250):
    """The crop indices of panoramaCropper.m:73-157 on the device: (offsetx, offsety, cropW, cropH) 1-based as in the
    reference, and whether rows offsety..offsety+cropH / columns offsetx..offsetx+cropW lie inside the image."""
    white = str(canvasColor).lower() == "white"
    if _capi.is_torch(stitchedImage):
        img = stitchedImage.contiguous()
        if str(img.dtype) != "torch.uint8" or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("stitchedImage must be M-by-N-by-3 uint8")
        h, w = int(img.shape[0]), int(img.shape[1])
    else:
        img = np.ascontiguousarray(stitchedImage)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("stitchedImage must be M-by-N-by-3 uint8")
        h, w = img.shape[:2]
    rect = np.zeros(4, np.int32)
    valid = np.zeros(1, np.int32)
    check(lib.aps_crop_rect(ptr(img), h, w, _capi.APS_IMG_U8_HWC, int(white), float(whiteRange if white else blackRange),
                            ptr(rect), ptr(valid)))
    return tuple(int(v) for v in rect), bool(valid[0])


def panoramaCropper(input, stitchedImage):
    """croppedImage = panoramaCropper(input, stitchedImage) (panoramaCropper.m:1-178): same field checks, same crop
    (offsety:offsety+cropH, offsetx:offsetx+cropW, 1-based inclusive), the input unchanged with a warning when the
    rectangle leaves the image ("Image has background holes")."""
    import warnings

    req = ["canvasColor", "blackRange", "whiteRange", "showCropBoundingBox", "displayPanoramas"]
    missing = [f for f in req if f not in input]
    if missing:
        raise ValueError("panoramaCropper:MissingField: Missing required input fields: " + ", ".join(missing))
    color = str(input["canvasColor"]).lower()
    if color not in ("black", "white"):
        raise ValueError('panoramaCropper:InvalidCanvasColor: input.canvasColor must be "black" or "white".')
    for f in ("blackRange", "whiteRange"):
        v = input[f]
        if not (np.isscalar(v) and np.isfinite(v) and 0 <= v <= 255):
            raise ValueError(f"panoramaCropper:Invalid{f[0].upper() + f[1:]}: input.{f} must be a numeric scalar in [0,255].")
    for f in ("showCropBoundingBox", "displayPanoramas"):
        if not isinstance(input[f], (bool, np.bool_)):
            raise ValueError(f"panoramaCropper:InvalidFlag: input.{f} must be a logical scalar.")
    (ox, oy, cw, ch), ok = cropRectangle(stitchedImage, color, input["blackRange"], input["whiteRange"])
    if not ok:
        warnings.warn("Cannot crop the image. Image has background holes.")
        return stitchedImage
    return stitchedImage[oy - 1:oy + ch, ox - 1:ox + cw]


PANO_PROJECTIONS = ("planar", "cylindrical", "spherical", "equirectangular", "stereographic")


def panorama_file_names(input, panoStore, myImg, datasetName):
    """The file names cropNsavePanorama.m:136-208 writes, in its order: for every panorama ii (1-based) and every projection
    field present and non-empty: '<proj>_<transformationType>_<myImg>_<ii>_<dataset>.png'; then, with input.cropPanorama,
    '<proj>_cropped_...'; then, with showPanoramaImgsNums and showCropBoundingBox, '<proj>_annotated_...'.
    Returns [(file name, panorama index (0-based), projection, slot)], slot 0 = base, 1 = annotated, 2 = cropped
    (the cell slots {1}, {2}, {3} of the reference)."""
    tt = str(input["transformationType"])
    name = str(datasetName[myImg - 1])
    out = []
    for ii, rec in enumerate(panoStore):
        have = [p for p in PANO_PROJECTIONS if rec.get(p)]
        for tag, slot, on in (("", 0, True), ("_cropped", 2, int(input.get("cropPanorama", 0)) == 1),
                              ("_annotated", 1, bool(input.get("showPanoramaImgsNums")) and bool(input.get("showCropBoundingBox")))):
            if on:
                out += [(f"{p}{tag}_{tt}_{myImg}_{ii + 1}_{name}.png", ii, p, slot) for p in have]
    return out


def cropNsavePanorama(input, panoStore, myImg, datasetName):
    """panoStore = cropNsavePanorama(input, panoStore, myImg, datasetName) (cropNsavePanorama.m:1-215): panoStore is a list of
    dicts, one per panorama, with a projection name -> [base RGB, annotated RGB or None, cropped RGB] (the reference's 1 x M
    struct array of cells).  With input.cropPanorama every present projection's base panorama goes through panoramaCropper
    (crop rectangle on the device) into slot 3; with input.imageWrite the PNG files are written to input.imageSaveFolder under
    the reference's names (host I/O: PIL).  Same argument checks and messages as :51-66.  myImg is 1-based."""
    import os

    if not (isinstance(myImg, (int, np.integer)) and 1 <= myImg <= len(datasetName)):
        raise ValueError(f"cropNsavePanorama:InvalidDatasetIndex: myImg ({myImg}) must index into datasetName (numel={len(datasetName)}).")
    if not isinstance(datasetName[myImg - 1], str):
        raise ValueError("cropNsavePanorama:InvalidDatasetNameType: datasetName{myImg} must be char or string.")
    write = bool(input.get("imageWrite"))
    if write and "transformationType" not in input:
        raise ValueError("cropNsavePanorama:MissingTransformationType: input.transformationType is required when input.imageWrite is true.")
    if write and not os.path.isdir(input["imageSaveFolder"]):
        os.makedirs(input["imageSaveFolder"])
    if int(input.get("cropPanorama", 0)) == 1:
        for rec in panoStore:
            for p in PANO_PROJECTIONS:
                if rec.get(p):
                    cell = list(rec[p]) + [None] * (3 - len(rec[p]))
                    cell[2] = panoramaCropper(input, cell[0])
                    rec[p] = cell
    if write:
        from PIL import Image

        for fname, ii, p, slot in panorama_file_names(input, panoStore, myImg, datasetName):
            img = panoStore[ii][p][slot]
            if _capi.is_torch(img):
                img = img.cpu().numpy()
            Image.fromarray(np.ascontiguousarray(img)).save(os.path.join(input["imageSaveFolder"], fname))
    return panoStore


# ---- loadImages.m:57-68,103-215: the two per-image steps in front of the resize ----------------------------------------
def applyOrientation(img, orientation):
    """The EXIF orientation switch of imreadAutoRotate (loadImages.m:194-212) on an H x W [x C] array: numpy, or a torch
    tensor (host or resident: pure data movement, the result stays where the input is).  rot90(A, k) turns
    counter-clockwise like MATLAB's; unknown codes leave the image unchanged, like the reference's switch."""
    is_t = _capi.is_torch(img)
    if is_t:
        import torch

        flip = lambda a, ax: torch.flip(a, (ax,))  # noqa: E731
        rot = lambda a, k: torch.rot90(a, k, (0, 1))  # noqa: E731
    else:
        flip = lambda a, ax: np.flip(a, ax)  # noqa: E731
        rot = lambda a, k: np.rot90(a, k, (0, 1))  # noqa: E731
    o = int(orientation) if orientation is not None else 1
    if o == 2:
        out = flip(img, 1)
    elif o == 3:
        out = rot(img, 2)
    elif o == 4:
        out = flip(img, 0)
    elif o == 5:
        out = rot(flip(img, 1), -1)
    elif o == 6:
        out = rot(img, -1)
    elif o == 7:
        out = rot(flip(img, 1), 1)
    elif o == 8:
        out = rot(img, 1)
    else:
        return img
    return out.contiguous() if is_t else np.ascontiguousarray(out)


def convertToRGB(img):
    """loadImages.m:103-125: a single-channel image replicated to H x W x 3, a three-channel one unchanged."""
    if img.ndim == 2:
        img = img[:, :, None]
    if img.shape[2] == 1:
        return img.repeat(1, 1, 3).contiguous() if _capi.is_torch(img) else np.ascontiguousarray(np.repeat(img, 3, axis=2))
    return img


def imreadAutoRotate(filename, device=None):
    """img = imreadAutoRotate(filename) (loadImages.m:127-215): imread + the EXIF orientation (tag 274) undone.  Returns a
    uint8 numpy array, or with device='cuda' a resident torch tensor (the turn is then done on the device)."""
    from PIL import Image

    with Image.open(filename) as im_:
        orientation = None
        try:
            orientation = im_.getexif().get(274)
        except Exception:  # noqa: BLE001 - like the reference's try/catch around imfinfo: unreadable metadata = no turn
            orientation = None
        if im_.mode not in ("L", "RGB", "RGBA"):
            im_ = im_.convert("RGB")
        arr = np.asarray(im_)
    if device is not None:
        import torch

        arr = torch.from_numpy(np.ascontiguousarray(arr)).to(device)
    return applyOrientation(arr, orientation)
