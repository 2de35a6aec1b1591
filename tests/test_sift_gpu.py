"""GPU parity: device SIFT against the oracle.  Every stage was written to one evaluation order, so
keypoint sets, locations and descriptors are expected to agree BIT FOR BIT (float tolerance 0)."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_sift_oracle import blob_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(gpu):
    return import_module(gpu.__name__ + ".featureMatching")


def textured(rng, h, w, c=3):
    base = rng.random((h // 6 + 2, w // 6 + 2, c))
    img = np.kron(base, np.ones((6, 6, 1)))[:h, :w]
    img = img + 0.15 * rng.random((h, w, c))
    from scipy.ndimage import gaussian_filter
    img = gaussian_filter(img, (1.0, 1.0, 0))
    img = (img - img.min()) / (img.max() - img.min())
    return (img * 255).astype(np.uint8)


INPUT = {"detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6}


@pytest.mark.parametrize("h,w", [(120, 160), (97, 131), (240, 320), (64, 200), (35, 47), (18, 515), (129, 66)])
def test_sift_bit_exact_rgb(fm, h, w):
    """(Odd sizes, sizes one past a tile of the fused gray / 2x-base kernel (128 x 16 outputs) and of the blur (64 x 32),
    and planes whose halved octaves have odd sides: the blur of plane nl writes the next octave's base itself.)"""
    rng = np.random.default_rng(h * 7 + w)
    img = textured(rng, h, w)
    f, pts, aux = fm.sift_extract(INPUT, img, want_aux=True)
    od, ol, oa = oracle.sift(img)
    assert len(od) > (50 if h * w > 5000 else 5)
    assert f.shape == od.shape
    assert np.array_equal(pts, ol)
    assert np.array_equal(aux.view(np.uint32), oa.view(np.uint32))
    assert np.array_equal(f.view(np.uint32), od.view(np.uint32))


def test_sift_gray_blobs_and_other_parameters(fm):
    rng = np.random.default_rng(5)
    blobs = [(rng.uniform(20, 280), rng.uniform(20, 180), rng.uniform(2, 7), rng.choice([-1, 1]) * rng.uniform(30, 110))
             for _ in range(40)]
    img = blob_image(200, 300, blobs)
    for prm in ({"Sigma": 1.6, "NumLayersInOctave": 3, "ContrastThreshold": 0.0133, "EdgeThreshold": 10},
                {"Sigma": 1.4142135623, "NumLayersInOctave": 4, "ContrastThreshold": 0.005, "EdgeThreshold": 6},
                {"Sigma": 2.0, "NumLayersInOctave": 2, "ContrastThreshold": 0.001, "EdgeThreshold": 12}):
        inp = dict(INPUT, **prm)
        f, pts = fm.getFeaturePoints(inp, img)
        od, ol, _ = oracle.sift(img, prm["Sigma"], prm["NumLayersInOctave"], prm["ContrastThreshold"], prm["EdgeThreshold"])
        assert len(od) >= 10
        assert np.array_equal(pts, ol) and np.array_equal(f.view(np.uint32), od.view(np.uint32))
        assert pts.dtype == np.float64 and f.dtype == np.float32


def test_sift_edge_cases(fm, gpu):
    f, pts = fm.getFeaturePoints(INPUT, np.full((64, 64, 3), 128, np.uint8))
    assert f.shape == (0, 128) and pts.shape == (0, 2)
    f, pts = fm.getFeaturePoints(INPUT, np.zeros((3, 5), np.uint8))
    assert len(f) == 0
    with pytest.raises(ValueError):
        fm.getFeaturePoints(dict(INPUT, detector="nope"), np.zeros((8, 8), np.uint8))
    with pytest.raises(NotImplementedError):
        fm.getFeaturePoints(dict(INPUT, detector="ORB"), np.zeros((8, 8), np.uint8))
    with pytest.raises(gpu.ApsError):
        fm.getFeaturePoints(dict(INPUT, NumLayersInOctave=9), np.zeros((32, 32), np.uint8))


def test_sift_device_resident_and_run_to_run_identical(fm):
    import torch

    rng = np.random.default_rng(6)
    img = textured(rng, 300, 400)
    t = torch.from_numpy(img).cuda()
    a, pa = fm.sift_extract(INPUT, t, device_out=True)
    b, pb = fm.sift_extract(INPUT, t, device_out=True)
    assert a.is_cuda and torch.equal(a, b) and np.array_equal(pa, pb)
    od, ol, _ = oracle.sift(img)
    assert np.array_equal(a.cpu().numpy().view(np.uint32), od.view(np.uint32)) and np.array_equal(pa, ol)


def test_sift_then_match_equals_oracle_chain(fm):
    """SIFT -> matchFeaturesScratch on two overlapping crops: the device chain equals the oracle chain."""
    rng = np.random.default_rng(7)
    img = textured(rng, 220, 420)
    a, b = img[:, :300], img[:, 120:]
    fa, pa = fm.getFeaturePoints(INPUT, a)
    fb, pb = fm.getFeaturePoints(INPUT, b)
    m, met = fm.matchFeaturesScratch(fa, fb, MatchThreshold=1.5, MaxRatio=0.6)
    oa, la, _ = oracle.sift(a)
    ob, lb, _ = oracle.sift(b)
    om, omet = oracle.match_features(oa, ob, 0.6, 1.5)
    assert len(om) > 50 and np.array_equal(m, om)
    dx = pa[m[:, 0] - 1, 0] - pb[m[:, 1] - 1, 0]
    assert abs(np.median(dx) - 120) < 0.1
