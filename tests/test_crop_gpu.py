"""GPU parity: aps_crop_rect (csrc/crop.hip) against oracle/crop_oracle.c — integer results, identical or wrong."""
from importlib import import_module

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ip(gpu):
    return import_module(gpu.__name__ + ".imageProcessing")


def _img(mask, rng=None):
    a = np.zeros(mask.shape + (3,), np.uint8)
    if rng is None:
        a[mask] = 200
    else:
        a[mask] = rng.integers(1, 256, (int(mask.sum()), 3), dtype=np.uint8)
    return a


def _blobs(rng, h, w, n=6, holes=4):
    m = np.zeros((h, w), bool)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
        ry, rx = rng.uniform(0.1, 0.35) * h, rng.uniform(0.1, 0.35) * w
        m |= ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
    for _ in range(holes):
        cy, cx = rng.uniform(0.2, 0.8) * h, rng.uniform(0.2, 0.8) * w
        r = rng.uniform(0.01, 0.06) * min(h, w)
        m &= ~((yy - cy) ** 2 + (xx - cx) ** 2 < r * r)
    return m


def _check(ip, img, white=False, rng_=0):
    want_rect, want_ok, _ = oracle.crop_rect(img, white, rng_)
    got_rect, got_ok = ip.cropRectangle(img, "white" if white else "black", rng_, rng_)
    assert got_rect == want_rect and got_ok == want_ok, (img.shape, got_rect, want_rect)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (7, 1), (2, 2), (3, 64), (5, 65), (40, 63), (33, 128), (70, 129),
                                   (257, 300), (600, 1025), (1100, 2049)])
def test_random_shapes_equal_oracle(ip, shape):
    rng = np.random.default_rng(shape[0] * 7919 + shape[1])
    h, w = shape
    for trial in range(4):
        if trial == 0:
            m = rng.random((h, w)) < 0.7  # salt and pepper: many holes, many ties of height 1..
        elif trial == 1:
            m = np.ones((h, w), bool)
            m[: h // 5] = False
        else:
            m = _blobs(rng, h, w) if min(h, w) >= 8 else rng.random((h, w)) < 0.8
        _check(ip, _img(m, rng))


def test_spiral_background_needs_many_fill_rounds(ip):
    """A background corridor that winds inwards: the row/column passes advance one arm per round, and only the
    fixed point equals the 4-connected flood fill.  The spiral's centre is background reachable from the border, so
    nothing may be filled."""
    n = 61
    m = np.ones((n, n), bool)
    r0, c0, r1, c1 = 0, 0, n - 1, n - 1
    m[0, :] = False
    # carve a square spiral corridor of width 1 with walls of width 1
    r, c, dr, dc = 0, 0, 0, 1
    seg = n - 1
    steps = 0
    while seg > 0:
        for _ in range(2 if steps else 3):
            for _ in range(seg):
                r, c = r + dr, c + dc
                m[r, c] = False
            dr, dc = dc, -dr
            steps += 1
        seg -= 2
    img = _img(m)
    inside = oracle.crop_inside(img)
    assert not inside[r, c]  # the oracle reaches the end of the corridor
    _check(ip, img)
    m2 = m.copy()
    m2[0, 1] = True
    m2[1, 0] = True  # ...and with the entrance walled off the corridor becomes one big hole
    m2[0, 0] = True
    _check(ip, _img(m2))


def test_thresholds_white_canvas_and_gray_weights(ip):
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (90, 150, 3), dtype=np.uint8)
    for white, t in [(False, 0), (False, 37), (False, 128), (True, 250), (True, 200), (False, 255), (True, 0)]:
        _check(ip, a, white, t)
    g = np.zeros((40, 70, 3), np.uint8)
    g[5:30, 8:60] = (rng.integers(0, 3, (25, 52, 3)) * 127).astype(np.uint8)  # values around typical thresholds
    for t in (0, 14, 15, 29, 37, 75, 76, 126, 127):
        _check(ip, g, False, t)


def test_resident_input_and_the_cropper_wrapper(ip):
    import torch

    rng = np.random.default_rng(9)
    m = _blobs(rng, 300, 500)
    img = _img(m, rng)
    want_rect, want_ok, _ = oracle.crop_rect(img)
    got = ip.cropRectangle(torch.from_numpy(img).cuda())
    torch.cuda.synchronize()
    assert got == (want_rect, want_ok)
    inp = {"canvasColor": "black", "blackRange": 0, "whiteRange": 250, "showCropBoundingBox": False, "displayPanoramas": False}
    out = ip.panoramaCropper(inp, img)
    ox, oy, cw, ch = want_rect
    assert want_ok and np.array_equal(out, img[oy - 1:oy + ch, ox - 1:ox + cw]) and out.shape[:2] == (ch + 1, cw + 1)
    full = np.full((50, 60, 3), 77, np.uint8)  # content everywhere: the reference's range overshoots -> input returned
    with pytest.warns(UserWarning):
        assert ip.panoramaCropper(inp, full) is full
    with pytest.raises(ValueError):
        ip.panoramaCropper({"canvasColor": "black"}, img)
    with pytest.raises(ValueError):
        ip.panoramaCropper(dict(inp, canvasColor="green"), img)


def test_argument_errors(ip, gpu):
    with pytest.raises(ValueError):
        ip.cropRectangle(np.zeros((4, 4), np.uint8))
    rect = np.zeros(4, np.int32)
    valid = np.zeros(1, np.int32)
    a = np.zeros((4, 4, 3), np.uint8)
    capi = gpu._capi
    assert capi.lib.aps_crop_rect(capi.ptr(a), 4, 4, capi.APS_IMG_U8_HWC, 0, 300.0, capi.ptr(rect), capi.ptr(valid)) != 0
    assert capi.lib.aps_crop_rect(capi.ptr(a), 0, 4, capi.APS_IMG_U8_HWC, 0, 0.0, capi.ptr(rect), capi.ptr(valid)) != 0
    assert capi.lib.aps_crop_rect(capi.ptr(a), 4, 70000, capi.APS_IMG_U8_HWC, 0, 0.0, capi.ptr(rect), capi.ptr(valid)) != 0


def test_matlab_planar_layout(ip, gpu):
    rng = np.random.default_rng(11)
    img = _img(_blobs(rng, 120, 200), rng)
    planar = np.asfortranarray(img)  # h x w x 3 column-major = MATLAB's layout
    rect = np.zeros(4, np.int32)
    valid = np.zeros(1, np.int32)
    capi = gpu._capi
    buf = np.ascontiguousarray(planar.ravel(order="F"))
    capi.check(capi.lib.aps_crop_rect(capi.ptr(buf), 120, 200, capi.APS_IMG_U8_MATLAB, 0, 0.0, capi.ptr(rect), capi.ptr(valid)))
    want_rect, want_ok, _ = oracle.crop_rect(img)
    assert tuple(int(v) for v in rect) == want_rect and bool(valid[0]) == want_ok


def test_crop_nonzero_bbox_equals_oracle(gpu):
    """cropNonzeroBbox (renderPanorama.m:1459-1504) as a device reduction: rectangle and didCrop equal the oracle's on
    host arrays, resident tensors, both canvas colours, grey-rounding edge cases, empty and full images."""
    import torch

    rp = import_module(gpu.__name__ + ".renderPanorama")
    rng = np.random.default_rng(12)
    cases = []
    img = np.zeros((301, 517, 3), np.uint8)
    img[40:222, 100:400] = rng.integers(0, 256, (182, 300, 3))
    img[10, 20] = (0, 0, 4)    # gray rounds to 0: not foreground
    img[280, 500] = (0, 0, 5)  # gray rounds to 1: foreground
    cases.append((img, "black"))
    white = np.full((130, 259, 3), 255, np.uint8)
    white[50:60, 70:200] = rng.integers(0, 250, (10, 130, 3))
    white[100, 250] = (255, 255, 252)  # gray 255: canvas
    cases.append((white, "white"))
    cases.append((np.zeros((9, 70, 3), np.uint8), "black"))
    cases.append((np.full((5, 3, 3), 7, np.uint8), "black"))
    for im_, color in cases:
        want_rect, want_did = oracle.crop_nonzero_bbox(im_, color == "white")
        out, rect, did = rp.cropNonzeroBbox(im_, color)
        assert tuple(rect) == want_rect and did == want_did
        r1, r2, c1, c2 = want_rect
        assert np.array_equal(out, im_[r1 - 1:r2, c1 - 1:c2])
        t = torch.from_numpy(im_).cuda()
        out_t, rect_t, did_t = rp.cropNonzeroBbox(t, color)
        assert tuple(rect_t) == want_rect and did_t == want_did and torch.equal(out_t.cpu(), torch.from_numpy(out))


def test_crop_and_save_panorama_writes_the_reference_files(ip, tmp_path):
    """cropNsavePanorama.m:68-208: slot 3 = panoramaCropper's crop of the base panorama, PNG files under the reference's
    names in input.imageSaveFolder; the files decode to the arrays that were handed in."""
    from PIL import Image

    rng = np.random.default_rng(3)
    m = np.zeros((90, 140), bool)
    m[10:80, 15:120] = True
    m[10:30, 15:40] = False  # a notch: the largest inscribed rectangle is not the bounding box
    pano = _img(m, rng)
    inp = {"canvasColor": "black", "blackRange": 0, "whiteRange": 250, "showCropBoundingBox": False, "displayPanoramas": False,
           "cropPanorama": 1, "imageWrite": True, "imageSaveFolder": str(tmp_path / "out"), "transformationType": "projective",
           "showPanoramaImgsNums": False}
    store = ip.cropNsavePanorama(inp, [{"spherical": [pano, None, None]}], 1, ["set"])
    crop = store[0]["spherical"][2]
    want = oracle.crop_rect(pano)[0]  # (offsetx, offsety, cropW, cropH), 1-based
    assert crop.shape == (want[3] + 1, want[2] + 1, 3) and (crop.sum(2) > 0).mean() > 0.9
    base = np.asarray(Image.open(tmp_path / "out" / "spherical_projective_1_1_set.png"))
    cut = np.asarray(Image.open(tmp_path / "out" / "spherical_cropped_projective_1_1_set.png"))
    assert np.array_equal(base, pano) and np.array_equal(cut, crop)
    assert sorted(p.name for p in (tmp_path / "out").iterdir()) == ["spherical_cropped_projective_1_1_set.png", "spherical_projective_1_1_set.png"]
