"""BASELINE.json configs[1] at its own shape: 20 views of 1600 x 1200, f = 1400 px, 10 x 2 yaw/pitch
grid with 35 % overlap (a full ring), spherical projection, 3-band multiband blend, on one MI355X.
Per-stage oracle comparison on the 2 x 2 sub-block the oracle finishes in seconds; properties at full size."""
from importlib import import_module

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

NX, NY, W, H, F, OVERLAP = 10, 2, 1600, 1200, 1400.0, 0.35
bits = lambda x: np.ascontiguousarray(x).view(np.uint8)  # noqa: E731


@pytest.fixture(scope="module")
def scene(gpu):
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    views, cams = synth.make_scene(NX, NY, W, H, F, OVERLAP, seed=2024, device="cuda", finest_px=4.0)
    torch.cuda.synchronize()
    return views, cams


def test_full_set_stitches_into_one_ring_panorama(gpu, scene):
    import torch

    pl = import_module(gpu.__name__ + ".pipeline")
    views, cams = scene
    inp = pl.default_input(bands=3)
    panos, info = pl.stitch(inp, views, Ks=[c["K"] for c in cams], tile=(2048, 2048))
    assert info["n_components"] == 1 and len(panos) == 1
    comp = info["components"][0]
    assert comp["members"] == list(range(NX * NY))
    assert min(info["n_features"]) > 500
    # every grid neighbour (ring closure included: 10 x 38.7 deg > 360 deg) is a verified pair
    pairs = {tuple(sorted(p)) for p in info["result"]["pairs"]}
    for iy in range(NY):
        for ix in range(NX):
            a, b = iy * NX + ix, iy * NX + (ix + 1) % NX
            assert tuple(sorted((a, b))) in pairs, (a, b)
    pano = panos[0]
    # spherical canvas of a full ring: width ~ 2 pi f (+ 2 % margin), height ~ the two rows' pitch span
    assert abs(pano.shape[1] - 2 * np.pi * F * 1.02) < 0.03 * 2 * np.pi * F
    cov = (pano.amax(dim=2) > 0)
    mid = cov[pano.shape[0] // 3: 2 * pano.shape[0] // 3]
    assert mid.float().mean().item() > 0.97
    # estimated rotations agree with the scene's up to one global rotation
    Rg = [c["R"] for c in cams]
    Re = comp["cameras"]
    rel = Re[0]["R"].T @ Rg[0]
    for k in range(NX * NY):
        d = Re[k]["R"] @ rel @ Rg[k].T
        ang = np.degrees(np.arccos(np.clip((np.trace(d) - 1) / 2, -1, 1)))
        assert ang < 1.0, (k, ang)
    del torch


def test_two_by_two_block_matches_the_oracle_stage_by_stage(gpu, scene):
    fm = import_module(gpu.__name__ + ".featureMatching")
    im = import_module(gpu.__name__ + ".imageMatching")
    rp = import_module(gpu.__name__ + ".renderPanorama")
    views, cams = scene
    ids = [4, 5, NX + 4, NX + 5]  # the middle of the ring: no wrap of theta, a compact canvas
    imgs = [views[k].cpu().numpy() for k in ids]
    sub_cams = [cams[k] for k in ids]
    # SIFT: descriptors and locations bit-identical
    feats = []
    for img in imgs:
        f, pts = fm.sift_extract({"detector": "SIFT"}, img)
        od, ol, _ = oracle.sift(img)
        assert f.shape == od.shape and f.shape[0] > 500
        assert np.array_equal(bits(f), bits(od)) and np.array_equal(bits(pts), bits(ol))
        feats.append((f, pts))
    # all six pairs: match lists identical, RANSAC (explicit draws) model bits and masks identical
    inp = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 500}
    verified = 0
    for j in range(1, 4):
        for i in range(j):
            m, met = fm.matchFeaturesScratch(feats[i][0], feats[j][0], MatchThreshold=1.5, MaxRatio=0.6)
            om, omet = oracle.match_features(feats[i][0], feats[j][0], 0.6, 1.5, True, 2)
            assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))
            if len(m) < 8:
                continue
            p_i = feats[i][1][m[:, 0] - 1].astype(np.float64)
            p_j = feats[j][1][m[:, 1] - 1].astype(np.float64)
            samples = im.draw_samples([len(m)], 564, seed=3)[0]
            Hm, mask, found = im.estimateTransformationRANSAC(p_j, p_i, "projective", inp, sample_idx=samples)
            oH, omask, ofound, _ = oracle.ransac_homography(p_j, p_i, samples, 5.5, 99.9, 500)
            assert found == ofound and np.array_equal(mask, omask)
            if found:
                assert np.array_equal(bits(Hm), bits(oH))
                verified += int(mask.sum() > 8 + 0.3 * len(m))
    assert verified >= 4
    # render (ground-truth cameras, spherical, 3 bands): stated tolerance against the oracle
    sizes = [(H, W, 3)] * 4
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 3, "pyrSigma": 1.0, "tile": (1024, 1024), "cropBorder": False}
    pano, _, cov, geo = rp.renderPanorama({}, imgs, sizes, sub_cams, "spherical", 0, opts, return_covered=True)
    op, oc = oracle.render(imgs, sub_cams, geo, (1024, 1024), 2.0, "multiband", 3, 1.0)
    assert pano.shape == op.shape and oc.mean() > 0.4 and pano.shape[1] < 4000
    assert (cov != oc).mean() <= 1e-4
    both = (cov == 1) & (oc == 1)
    diff = np.abs(pano.astype(int) - op.astype(int))[both]
    assert (diff <= 1).mean() >= 0.9995, (diff <= 1).mean()
    assert diff.max() <= 2, diff.max()
