"""GPU parity for warp / fuse / blend against the oracle.

Tolerance (stated per north star "warped/blended pixels within a stated fp32 tol"):
  * building blocks that involve no transcendental (multiband, linear, imresize chains, imageWarp) must be
    BIT-EXACT: both sides evaluate the same f32 fma chains in the same order;
  * anything downstream of the ray generator differs only through sinf/cosf (device libm vs glibc, <= 2 ulp):
    sampled colours/weights within 2e-4 absolute on [0,1] data wherever both masks agree, masks may differ on
    at most 0.1 % of pixels (rays that land within rounding of an image border), final uint8 panoramas within
    1 grey level on >= 99.8 % of pixels and, for the blended modes, within 2 grey levels on EVERY pixel both sides cover.
"""
import math
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_render_oracle import cam

pytestmark = pytest.mark.gpu

ATOL_F32 = 2e-4


@pytest.fixture(scope="module")
def rp(gpu):
    return import_module(gpu.__name__ + ".renderPanorama")


@pytest.fixture(scope="module")
def bl(gpu):
    return import_module(gpu.__name__ + ".blending")


@pytest.fixture(scope="module")
def ip(gpu):
    return import_module(gpu.__name__ + ".imageProcessing")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("K,h,w,levels", [(1, 32, 48, 3), (2, 33, 47, 4), (3, 64, 64, 5), (4, 5, 9, 5),
                                          (2, 1, 7, 3), (3, 130, 257, 6)])
def test_multiband_bit_exact(bl, K, h, w, levels):
    rng = np.random.default_rng(K * 1000 + h)
    C = rng.random((K, h, w, 3), dtype=np.float32)
    W = rng.random((K, h, w), dtype=np.float32)
    W[:, : h // 3, : w // 2] = 0  # uncovered corner
    W[0, :, w // 2:] = 0
    F = bl.multiBandBlending(list(C), list(W), levels, True, 1.0)
    O = oracle.multiband_blend(C, W, levels, 1.0)
    assert np.array_equal(bits(F), bits(O))


@pytest.mark.parametrize("K,h,w,levels,sigma", [(2, 64, 96, 4, 1.0), (3, 128, 64, 5, 1.0), (2, 96, 160, 3, 1.5),
                                                 (1, 32, 32, 5, 0.8), (3, 72, 136, 3, 1.0)])
def test_fused_blur_downsample_levels_are_bit_exact(bl, monkeypatch, K, h, w, levels, sigma):
    """Levels whose next size is exactly half take the fused blur+downsample kernel; it must give the oracle's bits
    and the bits of the unfused kernels (APS_RENDER_NO_FUSE=1), including mixed pyramids (72 -> 36 -> 18 -> 9)."""
    rng = np.random.default_rng(K * 100 + h + w)
    C = rng.random((K, h, w, 3), dtype=np.float32)
    W = rng.random((K, h, w), dtype=np.float32)
    W[0, : h // 2] = 0
    W[-1, :, w // 3:] = 0
    monkeypatch.delenv("APS_RENDER_NO_FUSE", raising=False)
    F = bl.multiBandBlending(list(C), list(W), levels, True, sigma)
    monkeypatch.setenv("APS_RENDER_NO_FUSE", "1")
    U = bl.multiBandBlending(list(C), list(W), levels, True, sigma)
    O = oracle.multiband_blend(C, W, levels, sigma)
    assert np.array_equal(bits(F), bits(O)) and np.array_equal(bits(U), bits(O))


def test_multiband_other_sigma_and_gray(bl):
    rng = np.random.default_rng(1)
    C = rng.random((2, 40, 56), dtype=np.float32)
    W = rng.random((2, 40, 56), dtype=np.float32)
    F = bl.multiBandBlending(list(C), list(W), 3, True, 1.7)
    O = oracle.multiband_blend(np.repeat(C[..., None], 3, 3), W, 3, 1.7)[..., 0]
    assert F.shape == (40, 56) and np.array_equal(bits(F), bits(O))
    with pytest.raises(ValueError):
        bl.multiBandBlending(list(C), list(W), 0)
    with pytest.raises(ValueError):
        bl.multiBandBlending(list(C), list(W[:1]), 3)


def test_linear_blend_bit_exact_and_integer_cast(bl):
    rng = np.random.default_rng(2)
    C = rng.random((3, 21, 34, 3), dtype=np.float32)
    W = rng.random((3, 21, 34), dtype=np.float32)
    W[:, 0, :] = 0
    F = bl.linearBlending(list(C), list(W))
    assert np.array_equal(bits(F), bits(oracle.linear_blend(C, W)))
    Cu = (C * 255).astype(np.uint8)
    Fu = bl.linearBlending(list(Cu), list(W))
    Ou = oracle.linear_blend(Cu.astype(np.float32), W)
    assert Fu.dtype == np.uint8 and np.array_equal(Fu, np.floor(np.clip(Ou, 0, 255) + 0.5).astype(np.uint8))


@pytest.mark.parametrize("mode", ["spherical", "cylindrical", "planar", "stereographic"])
def test_warp_tile_within_tolerance(rp, mode):
    rng = np.random.default_rng(3)
    W, H, f = 160, 120, 200.0
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    c = cam(f, W, H, yaw=0.07, pitch=-0.05)
    Rref = cam(f, W, H, yaw=0.02)["R"]
    geo = {"mode": mode, "H": 150, "W": 220, "fPan": f, "o0": -0.5, "o1": -0.4, "Rref": Rref}
    S, M, Wa, Wf = rp.warp_tile(img, c, geo, 10, 20, 128, 190, 2.0, gain=(1.1, 0.9, 1.0))
    oS, oM, oWa, oWf = oracle.warp_tile(img, c, geo, 10, 20, 128, 190, 2.0, gain=(1.1, 0.9, 1.0))
    assert oM.sum() > 3000
    assert (M != oM).mean() <= 1e-3
    both = M & oM
    # colours of a random image have per-pixel gradients of O(1): compare where the source is smooth enough
    np.testing.assert_allclose(Wa[both], oWa[both], atol=ATOL_F32)
    np.testing.assert_allclose(Wf[both], oWf[both], atol=ATOL_F32)
    assert np.abs(S[both] - oS[both]).max() <= 5e-3          # |grad| <= 1/px x sub-1e-3 px coordinate noise
    assert np.abs(S[both] - oS[both]).mean() <= ATOL_F32
    assert np.all(S[~M] == 0) and np.all(Wa[~M] == 0) and np.all(Wf[~M] == 0)


def _scene(rng, n=3, W=160, H=120, f=220.0, smooth=True):
    imgs, cams = [], []
    for i in range(n):
        base = rng.random((H // 8 + 2, W // 8 + 2, 3))
        img = np.kron(base, np.ones((8, 8, 1)))[:H, :W] if smooth else rng.random((H, W, 3))
        imgs.append((img * 255).astype(np.uint8))
        cams.append(cam(f, W, H, yaw=(i - (n - 1) / 2) * 0.35, pitch=0.03 * (-1) ** i))
    return imgs, cams


@pytest.mark.parametrize("blending,policy", [("multiband", "last"), ("linear", "last"), ("none", "last"),
                                             ("none", "first"), ("none", "maxangle")])
def test_render_matches_oracle(rp, blending, policy):
    rng = np.random.default_rng(4)
    imgs, cams = _scene(rng)
    sizes = [(120, 160, 3)] * 3
    opts = {"anglePower": 2, "blending": blending, "pyrLevels": 3, "pyrSigma": 1.0, "tile": (64, 96),
            "cropBorder": False, "composeNonePolicy": policy}
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 1, opts, return_covered=True)
    op, oc = oracle.render(imgs, cams, geo, (64, 96), 2.0, blending, 3, 1.0, policy)
    assert pano.shape == op.shape and oc.sum() > 10000
    assert (cov != oc).mean() <= 1e-3
    both = (cov == 1) & (oc == 1)
    diff = np.abs(pano.astype(int) - op.astype(int))[both]
    assert (diff <= 1).mean() >= 0.998
    if blending != "none":
        # blended pixels: a border pixel that flips in one layer enters with a vanishing tent weight, so the bound holds
        # EVERYWHERE both sides cover ('none' picks one image per pixel: a flip swaps the source image there)
        assert diff.max() <= 2, diff.max()
    assert np.all(pano[cov == 0] == 0)


def test_render_five_bands_gains_white_canvas_and_partial_tiles(rp):
    rng = np.random.default_rng(5)
    imgs, cams = _scene(rng, n=4, W=200, H=150, f=260.0)
    sizes = [(150, 200, 3)] * 4
    gains = [(1.0, 1.0, 1.0), (0.9, 1.1, 1.0), (1.2, 0.8, 1.0), (1.0, 1.0, 0.7)]
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (100, 130),
            "cropBorder": False, "canvasColor": "white"}
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, "cylindrical", 0, opts, gains=gains,
                                          return_covered=True)
    op, oc = oracle.render(imgs, cams, geo, (100, 130), 2.0, "multiband", 5, 1.0, "last", True, gains)
    assert (cov != oc).mean() <= 1e-3
    both = (cov == 1) & (oc == 1)
    d5 = np.abs(pano.astype(int) - op.astype(int))[both]
    assert (d5 <= 1).mean() >= 0.998 and d5.max() <= 2
    assert np.all(pano[cov == 0] == 255)


@pytest.mark.parametrize("blending,mode", [("multiband", "spherical"), ("linear", "spherical"),
                                           ("multiband", "cylindrical"), ("multiband", "planar")])
def test_footprint_culling_changes_no_bit(rp, monkeypatch, blending, mode):
    """Layers are processed on their footprint rectangles only; the result must equal full-tile processing
    (APS_RENDER_NO_CULL=1) in every byte, including tiles where footprints are slivers, and coverage too."""
    rng = np.random.default_rng(11)
    imgs, cams = _scene(rng, n=6, W=220, H=140, f=300.0)
    sizes = [(140, 220, 3)] * 6
    opts = {"anglePower": 2, "blending": blending, "pyrLevels": 5, "pyrSigma": 1.0, "tile": (72, 104),
            "cropBorder": False}
    monkeypatch.delenv("APS_RENDER_NO_CULL", raising=False)
    a, _, ca, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, return_covered=True)
    monkeypatch.setenv("APS_RENDER_NO_CULL", "1")
    b, _, cb, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, return_covered=True)
    assert a.shape == b.shape and ca.sum() > 10000
    assert np.array_equal(ca, cb)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("blending", ["multiband", "linear", "none"])
def test_planar_scan_compositing_matches_oracle_pieces(rp, ip, blending):
    """pureNonRotationalPanoramas (renderPanorama.m:519-699): canvas from the H2refined corner maps, every image
    and its tent map warped to the full canvas, whole-canvas blend.  Every step is bit-exact against the oracle's
    pieces, so the final uint8 canvas must be identical."""
    rng = np.random.default_rng(21)
    imgs = [rng.integers(0, 256, (60, 90, 3), dtype=np.uint8) for _ in range(3)]
    Hs = [np.eye(3), np.array([[1.0, 0.01, 55.0], [-0.01, 1.0, 4.0], [1e-5, 0, 1.0]]),
          np.array([[0.98, 0.0, 108.5], [0.02, 1.01, -6.0], [0, 2e-5, 1.0]])]
    cams = [{"H2refined": H, "noRotation": 1} for H in Hs]
    opts = {"blending": blending, "pyrLevels": 3, "pyrSigma": 1.0, "canvasColor": "white"}
    pano, ann = rp.renderPanorama({"forcePlanarScan": True}, imgs, [(60, 90, 3)] * 3, cams, "planar", 0, opts)
    # restatement from oracle pieces
    lims = [ip.outputLimitsScratch(H, (1, 90), (1, 60)) for H in Hs]
    xMin, xMax = min(l[0][0] for l in lims), max(l[0][1] for l in lims)
    yMin, yMax = min(l[1][0] for l in lims), max(l[1][1] for l in lims)
    width, height = int(np.floor(xMax - xMin + 0.5)), int(np.floor(yMax - yMin + 0.5))
    assert pano.shape == (height, width, 3) and ann is None
    sx, sy = (xMax - xMin) / width, (yMax - yMin) / height
    Iw, Ww = [], []
    for im, H in zip(imgs, Hs):
        Iw.append(oracle.image_warp_h(im.astype(np.float32) / 255.0, H, height, width, xMin, yMin, sx, sy, 0.0))
        tent = np.outer(oracle.tent(60), oracle.tent(90)).astype(np.float32)
        Ww.append(np.clip(oracle.image_warp_h(tent, H, height, width, xMin, yMin, sx, sy, 0.0), 0, 1))
    assert all(np.array_equal(a, b) for a, b in zip(rp.warpWeights(imgs), [np.outer(oracle.tent(60), oracle.tent(90)).astype(np.float32)] * 3))
    C, W = np.stack(Iw), np.stack(Ww)
    if blending == "multiband":
        F = oracle.multiband_blend(C, W, 3, 1.0)
    elif blending == "linear":
        F = oracle.linear_blend(C, W)
    else:
        F = np.take_along_axis(np.moveaxis(C, 0, 3), np.argmax(W, 0)[:, :, None, None], 3)[..., 0]
    F = np.array(F, np.float32)
    F[~(W > 0).any(0)] = 1.0
    ref = np.clip(np.floor(255.0 * F.astype(np.float64) + 0.5), 0, 255).astype(np.uint8)
    assert (W > 0).any(0).mean() > 0.5
    assert np.array_equal(pano, ref)


@pytest.mark.parametrize("world", [2, 3])
def test_tile_shards_compose_to_the_full_render(rp, world):
    """The multi-GPU render: rank r paints tiles t with t % world == r into a zeroed canvas and the canvases are
    combined with max (parallel.py step 6).  The union must equal the one-process render in every byte, for
    host buffers and for resident (torch) outputs alike."""
    import torch

    rng = np.random.default_rng(31)
    imgs, cams = _scene(rng, n=5, W=200, H=130, f=280.0)
    sizes = [(130, 200, 3)] * 5
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 4, "pyrSigma": 1.0, "tile": (64, 80), "cropBorder": False}
    full, _, cov, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, return_covered=True)
    acc = np.zeros_like(full)
    acc_t = None
    for r in range(world):
        part, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, tile_subset=(r, world))
        acc = np.maximum(acc, part)
        dimgs = [torch.from_numpy(i).cuda() for i in imgs]
        torch.cuda.synchronize()
        pt, _ = rp.renderPanorama({}, dimgs, sizes, cams, "spherical", 2, opts, tile_subset=(r, world), device_out=True)
        acc_t = pt if acc_t is None else torch.maximum(acc_t, pt)
    assert cov.sum() > 10000 and np.array_equal(acc, full)
    assert np.array_equal(acc_t.cpu().numpy(), full)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("blending", ["multiband", "linear"])
def test_contiguous_tile_ranges_compose_to_the_full_render(gpu, rp, world, blending):
    """Round 5: a rank renders a contiguous, area-balanced run of the tile list (parallel.tile_ranges ->
    aps_render_tile_range) instead of the tiles t % world.  The runs must cover the canvas and reproduce the one-process
    render in every byte, host and resident."""
    import torch

    par = import_module(gpu.__name__ + ".parallel")
    rng = np.random.default_rng(33)
    imgs, cams = _scene(rng, n=5, W=200, H=130, f=280.0)
    sizes = [(130, 200, 3)] * 5
    opts = {"anglePower": 2, "blending": blending, "pyrLevels": 4, "pyrSigma": 1.0, "tile": (64, 80), "cropBorder": False}
    full, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, return_covered=True)
    ranges = par.tile_ranges(int(geo["H"]), int(geo["W"]), (64, 80), world)
    acc = np.zeros_like(full)
    acc_t = None
    dimgs = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    for r in range(world):
        part, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, tile_subset=("range",) + ranges[r])
        assert not (acc.astype(bool) & part.astype(bool)).any()      # disjoint
        acc = np.maximum(acc, part)
        pt, _ = rp.renderPanorama({}, dimgs, sizes, cams, "spherical", 2, opts, tile_subset=("range",) + ranges[r], device_out=True)
        acc_t = pt if acc_t is None else torch.maximum(acc_t, pt)
    assert cov.sum() > 10000 and np.array_equal(acc, full)
    assert np.array_equal(acc_t.cpu().numpy(), full)


@pytest.mark.parametrize("workers", [1, 2, 3])
@pytest.mark.parametrize("subset", [None, (1, 2)])
def test_threaded_tile_loop_changes_no_byte(rp, monkeypatch, workers, subset):
    """Resident renders drive the tile loop from a few host threads (own stream each, interleaved tile shares,
    APS_RENDER_WORKERS): the canvas must equal the single-stream render in every byte, for the full tile set and
    for a multi-GPU shard of it."""
    import torch

    rng = np.random.default_rng(32)
    imgs, cams = _scene(rng, n=5, W=200, H=130, f=280.0)
    sizes = [(130, 200, 3)] * 5
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 4, "pyrSigma": 1.0, "tile": (64, 80), "cropBorder": False}
    ref, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, tile_subset=subset)
    dimgs = [torch.from_numpy(i).cuda() for i in imgs]
    torch.cuda.synchronize()
    monkeypatch.setenv("APS_RENDER_WORKERS", str(workers))
    for _ in range(2):  # twice: the second call reuses the pool and the per-thread workspaces
        got, _ = rp.renderPanorama({}, dimgs, sizes, cams, "spherical", 2, opts, tile_subset=subset, device_out=True)
        assert ref.any() and np.array_equal(got.cpu().numpy(), ref)


def test_canvas_geometry_and_crop(rp):
    rng = np.random.default_rng(6)
    imgs, cams = _scene(rng)
    sizes = [(120, 160, 3)] * 3
    o = rp.default_opts({"anglePower": 2}, cams, 1)
    geo = rp.canvas_geometry(cams, sizes, "spherical", 1, o)
    # three cameras 0.35 rad apart, hfov = 2*atan(80/220) = 0.70 rad -> ~1.40 rad span * 1.02 * f
    assert abs(geo["W"] - math.ceil(220 * (0.70 + 0.70) * 1.02)) <= 6
    pano, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 1, {"anglePower": 2, "tile": (128, 128)})
    assert pano.shape[0] <= geo["H"] and pano.shape[1] <= geo["W"] and pano.any()
    with pytest.raises(ValueError):
        rp.canvas_geometry(cams, sizes, "mercator", 1, o)


def test_image_warp_bit_exact(ip):
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    H = np.array([[1.01, 0.02, 7.5], [-0.015, 0.99, -3.25], [1e-5, -2e-5, 1.0]])
    view = ip.imref2dScratch((100, 140), (-5.5, 134.5), (-4.5, 95.5))
    out = ip.imageWarp(img, H, view, "bilinear", 9)
    ref = oracle.image_warp_h(img, H, 100, 140, -5.5, -4.5, 1.0, 1.0, 9)
    assert np.array_equal(out, ref)
    f32 = rng.random((90, 120), dtype=np.float32)
    outf = ip.imageWarp(f32, H, view)
    reff = oracle.image_warp_h(f32, H, 100, 140, -5.5, -4.5, 1.0, 1.0, 0.0)
    assert np.array_equal(bits(outf), bits(reff))
    (xl, yl) = ip.outputLimitsScratch(H, (0.5, 120.5), (0.5, 90.5))
    assert xl[0] < 10 and xl[1] > 120 and yl[0] < 0


@pytest.mark.parametrize("method", ["nearest", "bilinear", "bicubic"])
def test_image_warp_methods_bit_exact(ip, method):
    """options.method of imageWarp.m: 'nearest' (:109-123), 'bilinear' (:125-168), 'bicubic' (:170-264), uint8 and single."""
    rng = np.random.default_rng(17)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    H = np.array([[0.97, 0.05, 6.5], [-0.04, 1.02, -3.25], [2e-5, -1e-5, 1.0]])
    view = ip.imref2dScratch((100, 140), (-5.5, 134.5), (-4.5, 95.5))
    out = ip.imageWarp(img, H, view, method, 9)
    ref = oracle.image_warp_h(img, H, 100, 140, -5.5, -4.5, 1.0, 1.0, 9, method=method)
    assert out.dtype == np.uint8 and np.array_equal(out, ref) and (out != 9).mean() > 0.5
    f32 = rng.random((90, 120), dtype=np.float32)
    outf = ip.imageWarp(f32, H, view, method, 0.25)
    reff = oracle.image_warp_h(f32, H, 100, 140, -5.5, -4.5, 1.0, 1.0, 0.25, method=method)
    assert np.array_equal(bits(outf), bits(reff))
    with pytest.raises(ValueError):
        ip.imageWarp(img, H, view, "lanczos")


# ---- gain-compensation overlap statistics ---------------------------------------------------
@pytest.mark.parametrize("mode,stride", [("spherical", 3), ("cylindrical", 5), ("planar", 2), ("stereographic", 1)])
def test_gain_overlap_stats_match_oracle(gpu, rp, mode, stride):
    gc = import_module(gpu.__name__ + ".gainCompensation")
    rng = np.random.default_rng(41)
    imgs, cams = _scene(rng, n=5, W=180, H=120, f=260.0)
    gains_true = np.array([1.0, 0.8, 1.25, 0.9, 1.1])
    imgs = [np.clip(im.astype(np.float32) * g, 0, 255).astype(np.uint8) for im, g in zip(imgs, gains_true)]
    sizes = [(120, 180, 3)] * 5
    o = rp.default_opts({"anglePower": 2}, cams, 2)
    geo = rp.canvas_geometry(cams, sizes, mode, 2, o)
    Nij, sCi, sCj = gc.gain_overlap_stats(imgs, cams, geo, stride)
    oN, oI, oJ = oracle.gain_overlap_stats(imgs, cams, geo, stride)
    # counts: the rays differ only through sinf/cosf (<= 2 ulp), so a sample within rounding of an image border may flip
    assert Nij.shape == (5, 5) and np.all(np.tril(Nij) == 0) and oN.sum() > 500
    assert np.abs(Nij - oN).sum() <= 2e-3 * oN.sum()
    big = oN >= 50
    assert big.sum() >= 3
    assert np.allclose((sCi / np.maximum(Nij, 1)[..., None])[big], (oI / np.maximum(oN, 1)[..., None])[big], rtol=2e-3)
    assert np.allclose((sCj / np.maximum(Nij, 1)[..., None])[big], (oJ / np.maximum(oN, 1)[..., None])[big], rtol=2e-3)
    g = gc.solve_gains(Nij, sCi, sCj, {"minOverlapSamples": 50})
    go = gc.solve_gains(oN, oI, oJ, {"minOverlapSamples": 50})
    assert g.shape == (5, 3) and np.allclose(g, go, rtol=2e-3) and np.all((g >= 0.25) & (g <= 4.0))


def test_gain_compensation_recovers_planted_gains(gpu, rp):
    """Views of one scene scaled by known factors: the solved gains must undo them (up to the common scale the
    sigma_g prior fixes near 1) and the compensated render must be more uniform than the raw one."""
    gc = import_module(gpu.__name__ + ".gainCompensation")
    rng = np.random.default_rng(43)
    imgs, cams = _scene(rng, n=4, W=200, H=140, f=300.0)
    smooth = [np.clip(60 + 0.5 * im.astype(np.float32), 0, 255) for im in imgs]
    planted = np.array([1.0, 0.7, 1.3, 0.85])
    imgs2 = [np.clip(s * p, 0, 255).astype(np.uint8) for s, p in zip(smooth, planted)]
    sizes = [(140, 200, 3)] * 4
    o = rp.default_opts({"anglePower": 2}, cams, 1)
    geo = rp.canvas_geometry(cams, sizes, "spherical", 1, o)
    g = gc.gainCompensationRKf(imgs2, cams, "spherical", 1, {"overlapStride": 2, "minOverlapSamples": 30}, geo)
    prod = g[:, 0] * planted
    assert prod.std() / prod.mean() < 0.08  # g_i * planted_i is (nearly) one constant
    assert np.allclose(g[:, 0], g[:, 1], rtol=0.05)


# ---- preprocessing: imresize / resizeImagesToLimits ------------------------------------------
@pytest.mark.parametrize("shape,arg,method", [((90, 130, 3), 0.37, "bicubic"), ((64, 48, 3), (40, 100), "bicubic"),
                                              ((75, 75), 0.5, "bilinear"), ((33, 200, 3), 2.0, "bicubic"),
                                              ((216, 384, 3), 800 / 3840, "bicubic"), ((50, 70, 3), (50, 35), "bilinear")])
def test_imresize_u8_bit_exact(ip, shape, arg, method):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    out = ip.imresize(img, arg, method)
    ref = oracle.imresize_u8(img, arg, method)
    assert out.shape == ref.shape and out.dtype == np.uint8 and np.array_equal(out, ref)


def test_resize_images_to_limits_modes(ip):
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (120, 200, 3), dtype=np.uint8), rng.integers(0, 256, (90, 90, 3), dtype=np.uint8),
            rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)]
    fit = ip.resizeImagesToLimits(imgs, 80, 100, "fit")
    # 'fit': s = min(80/h, 100/w) < 1 shrinks, small images stay; then everything goes to the common largest size
    assert {f.shape for f in fit} == {(80, 100, 3)} or len({f.shape for f in fit}) == 1
    s0 = min(80 / 120, 100 / 200)
    first = oracle.imresize_u8(imgs[0], s0, "bicubic")
    Hm = max(first.shape[0], oracle.imresize_u8(imgs[1], min(80 / 90, 100 / 90), "bicubic").shape[0], 40)
    Wm = max(first.shape[1], oracle.imresize_u8(imgs[1], min(80 / 90, 100 / 90), "bicubic").shape[1], 50)
    assert fit[0].shape == (Hm, Wm, 3) and np.array_equal(fit[0], oracle.imresize_u8(first, (Hm, Wm), "bicubic"))
    pad = ip.resizeImagesToLimits(imgs, 80, 100, "pad")
    assert all(p.shape == (80, 100, 3) for p in pad) and np.array_equal(pad[2][20:60, 25:75], imgs[2])
    crop = ip.resizeImagesToLimits(imgs, 80, 100, "fillcrop")
    assert all(c.shape == (80, 100, 3) for c in crop)
    with pytest.raises(ValueError):
        ip.resizeImagesToLimits(imgs, 80, 100, "stretch")


@pytest.mark.parametrize("mode,levels,tile,sigma,white", [
    ("spherical", 5, (72, 104), 1.0, False), ("cylindrical", 3, (64, 96), 1.0, True),
    ("planar", 4, (100, 130), 1.5, False), ("stereographic", 5, (61, 77), 1.0, False),
    ("spherical", 1, (64, 64), 1.0, False), ("spherical", 6, (33, 47), 0.8, False),
    ("spherical", 5, (512, 512), 1.0, False)])
def test_batched_multiband_equals_per_tile_path(rp, monkeypatch, mode, levels, tile, sigma, white):
    """render_batch.hip (all tiles level-major, warp fused into level 0, table-driven resize taps, compact
    footprint stores) against render.hip's per-tile kernels (APS_RENDER_LEGACY=1): every byte of the panorama and
    of the coverage must agree - odd tile sizes (non-half pyramid levels), partial edge tiles, tiles whose level
    count is clamped, tiles without any layer, gains, white canvas."""
    rng = np.random.default_rng(21)
    imgs, cams = _scene(rng, n=6, W=220, H=140, f=300.0)
    sizes = [(140, 220, 3)] * 6
    gains = [(1.0, 1.0, 1.0), (0.9, 1.1, 1.0), (1.2, 0.8, 1.0), (1.0, 1.0, 0.7), (1.05, 1.0, 0.95), (1, 1, 1)]
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": levels, "pyrSigma": sigma, "tile": tile,
            "cropBorder": False, "canvasColor": "white" if white else "black", "margin": 0.08}
    monkeypatch.delenv("APS_RENDER_LEGACY", raising=False)
    # the default batched warp samples with reciprocals instead of IEEE divisions (tolerance against the per-tile path
    # below); APS_WARP_EXACT=1 keeps the per-tile arithmetic in the warp, so that everything ELSE in the batched pipeline
    # is still pinned byte for byte
    fast, _, fcov, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    monkeypatch.setenv("APS_WARP_EXACT", "1")
    # APS_RENDER_CHECK_RECTS: the analytic (host) footprints of the cylindrical / spherical canvases are checked against the
    # exact coverage kernel inside the call - every exact footprint must lie inside its analytic rectangle
    monkeypatch.setenv("APS_RENDER_CHECK_RECTS", "1")
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    monkeypatch.delenv("APS_RENDER_CHECK_RECTS")
    monkeypatch.setenv("APS_RENDER_DEVICE_COVER", "1")  # and the panorama must not depend on where the footprints came from
    pano_d, _, cov_d, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    monkeypatch.delenv("APS_RENDER_DEVICE_COVER")
    assert np.array_equal(pano, pano_d) and np.array_equal(cov, cov_d)
    monkeypatch.setenv("APS_RENDER_LEGACY", "1")
    ref, _, rcov, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    assert cov.sum() > 20000 and (cov == 0).sum() > 100
    assert np.array_equal(cov, rcov)
    assert np.array_equal(pano, ref)
    # the fast warp against the exact one: the SAME coverage (rays near an image border are re-projected exactly),
    # colours within one grey level (two at most) - the tolerance the oracle comparison states
    assert np.array_equal(fcov, cov)
    both = (fcov == 1) & (cov == 1)
    dd = np.abs(fast.astype(int) - pano.astype(int))[both]
    assert dd.max() <= 2 and (dd <= 1).mean() >= 0.9995, (dd.max(), (dd <= 1).mean())
    assert (dd == 0).mean() >= 0.97, (dd == 0).mean()


def test_batched_multiband_tile_subsets_compose(rp, monkeypatch):
    """aps_render_tiles(first, step) on the batched path: the shards of two 'ranks' paint disjoint tiles whose
    union is the one-call panorama."""
    rng = np.random.default_rng(22)
    imgs, cams = _scene(rng, n=5, W=220, H=140, f=300.0)
    sizes = [(140, 220, 3)] * 5
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 4, "pyrSigma": 1.0, "tile": (64, 80),
            "cropBorder": False}
    monkeypatch.delenv("APS_RENDER_LEGACY", raising=False)
    full, _, cov, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, return_covered=True)
    parts = [rp.renderPanorama({}, imgs, sizes, cams, "spherical", 2, opts, return_covered=True, tile_subset=(k, 2))
             for k in range(2)]
    assert np.array_equal(np.maximum(parts[0][0], parts[1][0]), full)
    assert np.array_equal(np.maximum(parts[0][2], parts[1][2]), cov)
    assert not np.any((parts[0][2] == 1) & (parts[1][2] == 1))


@pytest.mark.parametrize("mode", ["spherical", "cylindrical"])
def test_analytic_footprints_contain_the_exact_ones_for_tilted_wide_and_near_pole_cameras(rp, monkeypatch, mode):
    """Host-side footprints (forward image of the image border, clipped per tile) against the exact coverage kernel:
    rolled and pitched cameras, a wide field of view, views close to a pole and one ACROSS the theta = +-pi seam (for which
    the analytic path must step aside and the coverage kernel take over) - the check inside the library raises if an exact
    footprint leaves its analytic rectangle; the panorama must equal the device-cover one in every byte."""
    rng = np.random.default_rng(41)
    W, H = 200, 150
    def rot(yaw, pitch, roll):
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1.0]])
        return (Ry @ Rx @ Rz).T
    poses = [(0.0, 0.0, 0.0, 260.0), (0.5, 0.3, 0.4, 260.0), (-0.6, -0.5, -0.7, 180.0), (1.2, 0.9, 0.2, 300.0),
             (-1.4, -1.0, 1.0, 240.0), (2.2, 0.1, -0.3, 120.0)]
    if mode == "spherical":
        poses.append((3.1, 0.2, 0.1, 260.0))  # straddles the seam: analytic path declines, device cover runs
    imgs, cams = [], []
    for (yaw, pitch, roll, f) in poses:
        base = rng.random((H // 8 + 2, W // 8 + 2, 3))
        imgs.append((np.kron(base, np.ones((8, 8, 1)))[:H, :W] * 255).astype(np.uint8))
        cams.append({"K": np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]), "R": rot(yaw, pitch, roll)})
    sizes = [(H, W, 3)] * len(imgs)
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 4, "pyrSigma": 1.0, "tile": (96, 128), "cropBorder": False}
    monkeypatch.setenv("APS_RENDER_CHECK_RECTS", "1")
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, mode, 0, opts, return_covered=True)
    monkeypatch.delenv("APS_RENDER_CHECK_RECTS")
    monkeypatch.setenv("APS_RENDER_DEVICE_COVER", "1")
    ref, _, rcov, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 0, opts, return_covered=True)
    monkeypatch.delenv("APS_RENDER_DEVICE_COVER")
    assert cov.sum() > 20000 and np.array_equal(cov, rcov) and np.array_equal(pano, ref)


@pytest.mark.parametrize("ds", [1, 4, 7])
@pytest.mark.parametrize("resident", [False, True])
def test_gain_overlap_stats_of_warped_canvases_match_oracle(gpu, ds, resident):
    """aps_gain_overlap_stats_warped (gainCompensationH.m:45-52,78-149) against the oracle on random canvases with holes,
    non-finite colours, a single-channel set and > 16 images covering one point (the kernel's overflow path); host arrays
    and resident tensors."""
    import torch

    gc = import_module(gpu.__name__ + ".gainCompensation")
    rng = np.random.default_rng(100 + ds)
    for n, ch in ((5, 3), (19, 3), (3, 1)):
        Hc, Wc = 83, 121
        Iw = [rng.random((Hc, Wc, ch)).astype(np.float32) for _ in range(n)]
        Ww = [((rng.random((Hc, Wc)) > 0.3) * rng.random((Hc, Wc))).astype(np.float32) for _ in range(n)]
        Iw[1][::5, ::3, ch - 1] = np.nan
        Iw[2][3::7, 1::4, 0] = np.inf
        if resident:
            a = [torch.from_numpy(x).cuda() for x in Iw]
            b = [torch.from_numpy(x).cuda() for x in Ww]
            torch.cuda.synchronize()
        else:
            a, b = Iw, Ww
        N, sI, sJ = gc.gain_overlap_stats_warped(a, b, ds)
        oN, oI, oJ = oracle.gain_overlap_stats_warped(Iw, Ww, ds)
        assert np.array_equal(N, oN) and oN.sum() > 50 and np.all(np.tril(N) == 0)
        assert np.allclose(sI, oI, rtol=1e-12, atol=1e-9) and np.allclose(sJ, oJ, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("mode,tile,white", [("spherical", (72, 104), False), ("cylindrical", (64, 96), True),
                                             ("planar", (100, 130), False), ("stereographic", (61, 77), False),
                                             ("spherical", (512, 512), False)])
@pytest.mark.parametrize("blending,policy", [("linear", "last"), ("none", "last"), ("none", "first"), ("none", "maxangle")])
def test_batched_linear_and_none_equal_per_tile_path(rp, monkeypatch, mode, tile, white, blending, policy):
    """rw_fuse_kernel (render_batch.hip: 'linear' and 'none' blending of every tile in one launch, samples folded straight
    into per-pixel accumulators, renderPanorama.m:864-978) against render.hip's per-tile kernels (APS_RENDER_LEGACY=1:
    one float4 layer per image and tile, linear_fuse / none_fuse, paint): every byte of the panorama and of the coverage,
    for host and device footprints, partial edge tiles, tiles without any layer, gains, both canvas colours, the three
    'none' policies - and for a multi-GPU tile shard."""
    rng = np.random.default_rng(23)
    imgs, cams = _scene(rng, n=6, W=220, H=140, f=300.0)
    sizes = [(140, 220, 3)] * 6
    gains = [(1.0, 1.0, 1.0), (0.9, 1.1, 1.0), (1.2, 0.8, 1.0), (1.0, 1.0, 0.7), (1.05, 1.0, 0.95), (1, 1, 1)]
    opts = {"anglePower": 2, "blending": blending, "composeNonePolicy": policy, "tile": tile, "cropBorder": False,
            "canvasColor": "white" if white else "black", "margin": 0.08}
    monkeypatch.delenv("APS_RENDER_LEGACY", raising=False)
    monkeypatch.setenv("APS_RENDER_CHECK_RECTS", "1")
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    monkeypatch.delenv("APS_RENDER_CHECK_RECTS")
    monkeypatch.setenv("APS_RENDER_DEVICE_COVER", "1")
    pano_d, _, cov_d, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    monkeypatch.delenv("APS_RENDER_DEVICE_COVER")
    part, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, tile_subset=(1, 2))
    monkeypatch.setenv("APS_RENDER_LEGACY", "1")
    ref, _, rcov, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, return_covered=True)
    rpart, _ = rp.renderPanorama({}, imgs, sizes, cams, mode, 2, opts, gains=gains, tile_subset=(1, 2))
    assert cov.sum() > 20000 and (cov == 0).sum() > 100
    assert np.array_equal(cov, rcov) and np.array_equal(pano, ref)
    assert np.array_equal(cov_d, rcov) and np.array_equal(pano_d, ref)
    assert np.array_equal(part, rpart) and not np.array_equal(part, ref)
    op, oc = oracle.render(imgs, cams, geo, tile, 2.0, blending, 3, 1.0, policy, white, gains)
    assert (cov != oc).mean() <= 1e-3  # (the per-tile path's stated agreement with the oracle: test_render_matches_oracle)
    dd = np.abs(op.astype(int) - pano.astype(int))[(cov == 1) & (oc == 1)]
    assert (dd <= 1).mean() >= 0.998


def test_tile_range_argument_errors_and_empty_range(gpu, rp):
    """aps_render_tile_range: a negative or reversed range is refused; an empty range paints nothing."""
    rng = np.random.default_rng(34)
    imgs, cams = _scene(rng, n=3, W=120, H=90, f=200.0)
    sizes = [(90, 120, 3)] * 3
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 3, "pyrSigma": 1.0, "tile": (64, 64), "cropBorder": False}
    none, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 1, opts, tile_subset=("range", 2, 2))
    assert not none.any()
    with pytest.raises(gpu.ApsError) as e:
        rp.renderPanorama({}, imgs, sizes, cams, "spherical", 1, opts, tile_subset=("range", 3, 1))
    assert e.value.code == gpu._capi.APS_E_ARG
