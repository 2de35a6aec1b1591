"""BASELINE.json configs[0] as a chain: two 1024 x 768 views related by a planar homography ->
SIFT -> exhaustive 2-NN + ratio -> RANSAC homography -> H2refined = (I, H) -> planar-scan compositing
(renderPanorama.m:78-81 -> pureNonRotationalPanoramas, :519-699), every stage against the oracle.
The two views see the same world through cameras that differ by a rotation only, so the image-to-image map is exactly
the homography K R1 R0' K^-1; the reference reaches this path with cameras(1).noRotation == 1."""
from importlib import import_module

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

W, H, F = 1024, 768, 1100.0
bits = lambda x: np.ascontiguousarray(x).view(np.uint8)  # noqa: E731


@pytest.fixture(scope="module")
def pair(gpu):
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    views, cams = synth.make_scene(2, 1, W, H, F, 0.55, seed=77, device="cuda", finest_px=4.0)
    torch.cuda.synchronize()
    return [v.cpu().numpy() for v in views], cams


@pytest.mark.parametrize("blending", ["multiband", "linear"])
def test_two_view_homography_stitch_matches_the_oracle_stage_by_stage(gpu, pair, blending):
    fm = import_module(gpu.__name__ + ".featureMatching")
    im = import_module(gpu.__name__ + ".imageMatching")
    rp = import_module(gpu.__name__ + ".renderPanorama")
    ip = import_module(gpu.__name__ + ".imageProcessing")
    pl = import_module(gpu.__name__ + ".pipeline")
    imgs, cams = pair
    inp = pl.default_input(bands=3)
    # getFeaturePoints on both views: bit-identical to the oracle
    feats = []
    for img in imgs:
        f, pts = fm.getFeaturePoints(inp, img)
        od, ol, _ = oracle.sift(img)
        assert f.shape == od.shape and f.shape[0] > 1000
        assert np.array_equal(bits(f), bits(od)) and np.array_equal(bits(pts), bits(ol))
        feats.append((f, pts))
    # featureMatchingPairwise (cell (1,2)) == the oracle's matchFeaturesScratch
    cells = fm.featureMatchingPairwise(inp, [f for f, _ in feats], 2)
    om, _ = oracle.match_features(feats[0][0], feats[1][0], inp["Ratiothreshold"], inp["Matchingthreshold"], True, 2)
    assert np.array_equal(cells[0][1].astype(np.int64), om.astype(np.int64)) and len(om) > 200
    # estimateTransformationRANSAC on explicit draws: tforms{1,2} maps view 2 -> view 1 (imageMatching.m:242)
    p1 = feats[0][1][om[:, 0] - 1].astype(np.float64)
    p2 = feats[1][1][om[:, 1] - 1].astype(np.float64)
    samples = im.draw_samples([len(om)], 564, seed=11)[0]
    Hm, mask, found = im.estimateTransformationRANSAC(p2, p1, "projective", inp, sample_idx=samples)
    oH, omask, ofound, _ = oracle.ransac_homography(p2, p1, samples, inp["maxDistance"], inp["inliersConfidence"], inp["maxIter"])
    assert found and ofound and np.array_equal(mask, omask) and np.array_equal(bits(Hm), bits(oH))
    assert mask.sum() > 8 + 0.3 * len(om)  # imageMatching.m:150: the pair is verified
    # the recovered map is the scene's: K R0 R1' K^-1 up to scale (pixel accuracy over the image)
    K = cams[0]["K"]
    Ht = K @ cams[0]["R"] @ cams[1]["R"].T @ np.linalg.inv(K)
    corners = np.array([[1, 1, 1], [W, 1, 1], [W, H, 1], [1, H, 1.0]]).T
    a, b = Hm @ corners, Ht @ corners
    assert np.abs(a[:2] / a[2] - b[:2] / b[2]).max() < 1.5
    # planar-scan compositing with H2refined = (I, H): identical to the composition of the oracle's pieces
    Hn = Hm / Hm[2, 2]
    pcams = [{"H2refined": np.eye(3), "noRotation": 1}, {"H2refined": Hn, "noRotation": 1}]
    opts = {"blending": blending, "pyrLevels": 3, "pyrSigma": 1.0, "canvasColor": "black"}
    pano, _ = rp.renderPanorama(inp, imgs, [(H, W, 3)] * 2, pcams, "planar", 0, opts)
    lims = [ip.outputLimitsScratch(T, (1, W), (1, H)) for T in (np.eye(3), Hn)]
    xMin, xMax = min(l[0][0] for l in lims), max(l[0][1] for l in lims)
    yMin, yMax = min(l[1][0] for l in lims), max(l[1][1] for l in lims)
    width, height = int(np.floor(xMax - xMin + 0.5)), int(np.floor(yMax - yMin + 0.5))
    assert pano.shape == (height, width, 3) and width > 1.3 * W
    sx, sy = (xMax - xMin) / width, (yMax - yMin) / height
    tent = np.outer(oracle.tent(H), oracle.tent(W)).astype(np.float32)
    Iw, Ww = [], []
    for img, T in zip(imgs, (np.eye(3), Hn)):
        Iw.append(oracle.image_warp_h(img.astype(np.float32) / 255.0, T, height, width, xMin, yMin, sx, sy, 0.0))
        Ww.append(np.clip(oracle.image_warp_h(tent, T, height, width, xMin, yMin, sx, sy, 0.0), 0, 1))
    C_, W_ = np.stack(Iw), np.stack(Ww)
    Fb = oracle.multiband_blend(C_, W_, 3, 1.0) if blending == "multiband" else oracle.linear_blend(C_, W_)
    Fb = np.array(Fb, np.float32)
    Fb[~(W_ > 0).any(0)] = 0.0
    ref = np.clip(np.floor(255.0 * Fb.astype(np.float64) + 0.5), 0, 255).astype(np.uint8)
    assert (W_ > 0).any(0).mean() > 0.6
    assert np.array_equal(pano, ref)
    # the seam is invisible where both views cover: the composite agrees with either warped view within the blend's reach
    both = (W_[0] > 0.2) & (W_[1] > 0.2)
    assert both.sum() > 50000
    assert np.abs(pano.astype(np.float32)[both] - 255.0 * C_[0][both]).mean() < 10.0


def test_planar_scan_gain_compensation_recovers_a_planted_gain(gpu, pair):
    """gainCompensationH on the configs[0] pair (renderPanorama.m:584-591 -> gainCompensationH.m): view 2 is darkened by a
    known factor, both views are warped to the planar canvas with the scene's homography, the overlap statistics come from
    the device (against the oracle: counts equal, sums to 1e-12) and the solved gains undo the factor to 1 %; the
    planar-scan renderer with opts.gainCompensation set applies exactly those gains."""
    rp = import_module(gpu.__name__ + ".renderPanorama")
    ip = import_module(gpu.__name__ + ".imageProcessing")
    gc = import_module(gpu.__name__ + ".gainCompensation")
    imgs, cams = pair
    planted = 0.8
    dark = np.clip(np.floor(imgs[1].astype(np.float32) * planted + 0.5), 0, 255).astype(np.uint8)
    views = [imgs[0], dark]
    K = cams[0]["K"]
    Ht = K @ cams[0]["R"] @ cams[1]["R"].T @ np.linalg.inv(K)
    Hn = Ht / Ht[2, 2]
    tforms = [np.eye(3), Hn]
    lims = [ip.outputLimitsScratch(T, (1, W), (1, H)) for T in tforms]
    xMin, xMax = min(l[0][0] for l in lims), max(l[0][1] for l in lims)
    yMin, yMax = min(l[1][0] for l in lims), max(l[1][1] for l in lims)
    width, height = int(np.floor(xMax - xMin + 0.5)), int(np.floor(yMax - yMin + 0.5))
    view = ip.imref2dScratch((height, width), (xMin, xMax), (yMin, yMax))
    Iw, Ww, _, _, _ = rp.pureNonRotationalImagesToCanvas(views, tforms, view, rp.warpWeights(views), {})
    for ds in (4, 3):
        N, sI, sJ = gc.gain_overlap_stats_warped(Iw, Ww, ds)
        oN, oI, oJ = oracle.gain_overlap_stats_warped(Iw, Ww, ds)
        assert np.array_equal(N, oN) and N[0, 1] > 10000 and N.sum() == N[0, 1]
        assert np.allclose(sI, oI, rtol=1e-12, atol=0) and np.allclose(sJ, oJ, rtol=1e-12, atol=0)
    g = gc.gainCompensationH(Iw, Ww, {"sigmag": 10.0})  # (gainCompensationH.m:30: its own default prior, loose)
    assert g.shape == (2, 3) and np.all((g >= 0.25) & (g <= 4.0))
    assert np.allclose(g[1] / g[0], 1.0 / planted, rtol=0.01)
    assert np.allclose(np.sqrt(g[0] * g[1] * planted), 1.0, atol=0.15)  # the pair's common scale stays near one
    # the renderer: opts.gainCompensation -> the same gains, applied before the blend
    pcams = [{"H2refined": T, "noRotation": 1} for T in tforms]
    opts = {"blending": "linear", "canvasColor": "black", "gainCompensation": 1, "sigmag": 10.0}
    pano_auto, _ = rp.renderPanorama({}, views, [(H, W, 3)] * 2, pcams, "planar", 0, opts)
    opts_off = {"blending": "linear", "canvasColor": "black"}
    pano_given, _ = rp.renderPanorama({}, views, [(H, W, 3)] * 2, pcams, "planar", 0, opts_off, gains=g)
    pano_none, _ = rp.renderPanorama({}, views, [(H, W, 3)] * 2, pcams, "planar", 0, opts_off)
    assert np.array_equal(pano_auto, pano_given) and not np.array_equal(pano_auto, pano_none)
    # and the compensated composite is more uniform across the seam than the raw one
    both = (Ww[0] > 0.3) & (Ww[1] > 0.3)
    a, b = Iw[0][both].mean(0), Iw[1][both].mean(0)
    assert np.abs(a * g[0] - b * g[1]).max() < 0.2 * np.abs(a - b).max()
