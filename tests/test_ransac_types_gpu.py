"""GPU parity for the non-projective transformTypes of estimateTransformationRANSAC.m (:227-452, :483-497): the device
fit / score / replay / refit against the oracle - model bits, masks, counts identical; draws explicit."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_ransac_types_oracle import TYPES, draws, world

pytestmark = pytest.mark.gpu

INP = {"maxDistance": 3.0, "inliersConfidence": 99.9, "maxIter": 500}


@pytest.fixture(scope="module")
def im(gpu):
    return import_module(gpu.__name__ + ".imageMatching")


def bits(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


@pytest.mark.parametrize("tform", TYPES)
def test_score_bit_exact(im, tform):
    rng = np.random.default_rng(30)
    p1, p2 = world(tform, rng, n=1500)
    p1[7] = [4000.0, -2.0]  # the largest |coordinate|: the scale of the translation path (:487)
    k = oracle.tform_min_points(tform)
    Hs = []
    for _ in range(80):
        H, ok = oracle.fit_tform(tform, p1, p2, rng.permutation(1500)[:k])
        Hs.append(H if ok else np.eye(3))
    Hs.append(np.eye(3))
    Hs.append(np.array([[1.0, 2, 3], [2, 4, 6], [0, 0, 1]]))
    Hs = np.stack(Hs)
    n, e, mask = im.ransac_score(Hs, p1, p2, 3.0, tform)
    on, oe, omask = oracle.ransac_score_tform(tform, Hs, p1, p2, 3.0)
    assert np.array_equal(n, on) and np.array_equal(mask, omask)
    assert np.array_equal(bits(e), bits(oe))
    assert n.max() > 700


def test_affine_score_rejects_collinear_inliers(im):
    x = np.linspace(0, 100, 40)
    p1 = np.r_[np.c_[x, 2 * x + 1], [[50.0, 500.0]]]
    p2 = p1 + [5, 5]
    p2[-1] += 400
    H = np.array([[1, 0, 5], [0, 1, 5], [0, 0, 1.0]])[None]
    for tform, want in (("affine", 0), ("similarity", 40), ("rigid", 40), ("translation", 40)):
        n, e, mask = im.ransac_score(H, p1, p2, 1.0, tform)
        on, oe, omask = oracle.ransac_score_tform(tform, H, p1, p2, 1.0)
        assert n[0] == want == on[0] and np.array_equal(mask, omask)


@pytest.mark.parametrize("tform", TYPES)
@pytest.mark.parametrize("m,outliers,noise", [(4, 0.0, 0.0), (5, 0.0, 0.3), (31, 0.3, 0.2), (400, 0.35, 0.3),
                                              (3001, 0.6, 0.5), (257, 0.0, 0.0), (1000, 0.9, 0.4)])
def test_whole_loop_bit_exact(im, tform, m, outliers, noise):
    rng = np.random.default_rng(40 + m)
    p1, p2 = world(tform, rng, n=m, outliers=outliers, noise=noise)
    s = im.draw_samples([m], 564, seed=m)[0]
    H, mask, found = im.estimateTransformationRANSAC(p1, p2, tform, INP, sample_idx=s)
    oH, omask, ofound, _ = oracle.ransac_tform(tform, p1, p2, s, 3.0, 99.9, 500)
    assert found == ofound
    assert np.array_equal(mask, omask)
    if found:
        assert np.array_equal(bits(H), bits(oH))
        if outliers <= 0.6 and m >= 31:
            assert mask.sum() >= 0.8 * (1 - outliers) * m


@pytest.mark.parametrize("tform", TYPES)
def test_rotated_scenes_follow_the_reference_too(im, tform):
    """A real rotation between the views: 'similarity' / 'rigid' answer with the transposed rotation inside the fit (the
    reference's formula), find few inliers and say so - the device must agree with the oracle there as well."""
    rng = np.random.default_rng(50)
    p1 = rng.uniform(0, 1500, (600, 2))
    th = 0.35
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    p2 = p1 @ (1.2 * R).T + [30, -40] + rng.normal(0, 0.3, p1.shape)
    s = im.draw_samples([600], 564, seed=9)[0]
    H, mask, found = im.estimateTransformationRANSAC(p1, p2, tform, INP, sample_idx=s)
    oH, omask, ofound, _ = oracle.ransac_tform(tform, p1, p2, s, 3.0, 99.9, 500)
    assert found == ofound and np.array_equal(mask, omask)
    if found:
        assert np.array_equal(bits(H), bits(oH))
    if tform == "affine":
        assert mask.sum() > 590


@pytest.mark.parametrize("tform", TYPES)
def test_tiny_inputs(im, tform):
    k = oracle.tform_min_points(tform)
    rng = np.random.default_rng(60)
    for m in range(0, 5):
        p1 = rng.uniform(0, 100, (m, 2))
        p2 = p1 + [3.0, 4.0]
        s = im.draw_samples([m], 40, seed=2)[0]
        H, mask, found = im.estimateTransformationRANSAC(p1, p2, tform, dict(INP, maxIter=20), sample_idx=s)
        oH, omask, ofound, _ = oracle.ransac_tform(tform, p1, p2, s, 3.0, 99.9, 20)
        assert found == ofound and np.array_equal(mask, omask), (tform, m)
        assert found == (m >= k) or tform == "affine"  # three points on a line would be degenerate for affine
        if found:
            assert np.array_equal(bits(H), bits(oH))


def test_device_draws_for_short_lists_equal_host_draws(im):
    counts = np.array([0, 1, 2, 3, 4, 5, 9], np.int64)
    host = im.draw_samples(counts, 300, seed=4)
    dev = im.draw_samples_device(counts, 300, seed=4).cpu().numpy().astype(np.uint32)
    assert np.array_equal(host, dev)
    for p, n in enumerate(counts):
        kk = min(int(n), 4)
        assert np.all(host[p, :, kk:] == 1)
        if kk:
            srt = np.sort(host[p, :, :kk], axis=1)
            assert np.all(np.diff(srt, axis=1) > 0) and srt.min() >= 1 and srt.max() <= n


@pytest.mark.parametrize("tform", TYPES)
def test_batch_equals_per_pair_oracle(im, tform):
    rng = np.random.default_rng(70)
    sizes = [1, 2, 3, 4, 60, 700, 2500, 64, 65]
    worlds = [world(tform, rng, n=m, outliers=0.3 if m > 10 else 0.0) for m in sizes]
    src = np.concatenate([w[0] for w in worlds])
    dst = np.concatenate([w[1] for w in worlds])
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    samples = im.draw_samples(sizes, 564, seed=5)
    inp = dict(INP, transformationType=tform)
    models, mask, found, ninl = im.ransac_batch(src, dst, ptr, samples, inp)
    for p, (p1, p2) in enumerate(worlds):
        oH, omask, ofound, _ = oracle.ransac_tform(tform, p1, p2, samples[p], 3.0, 99.9, 500)
        assert bool(found[p]) == ofound, (tform, p)
        assert np.array_equal(mask[ptr[p]:ptr[p + 1]].astype(bool), omask)
        if ofound:
            assert np.array_equal(bits(models[p]), bits(oH))
            assert ninl[p] == omask.sum()


def test_image_matching_with_an_affine_model(im):
    """imageMatching.m:121-156 with input.transformationType = 'affine': the candidate pairs go through the affine loop."""
    rng = np.random.default_rng(80)
    n = 3
    kp = [rng.uniform(0, 1000, (500, 2)) for _ in range(n)]
    A = np.array([[1.02, 0.05], [-0.04, 0.97]])
    kp[1][:300] = kp[0][:300] @ A.T + [40, 10] + rng.normal(0, 0.2, (300, 2))
    matches = [[None] * n for _ in range(n)]
    idx = np.arange(1, 361)
    mt = np.stack([idx, idx], 1)
    mt[300:, 1] = rng.permutation(np.arange(301, 501))[:60]  # 60 wrong matches
    matches[0][1] = mt
    inp = dict(INP, transformationType="affine", mBrownLowe=6)
    allM, num, tf = im.imageMatching(inp, n, kp, matches, seed=3)
    assert num[0, 1] >= 295 and tf[0][1] is not None
    # tforms{1,2} maps image-2 points to image-1 points: the inverse of A
    np.testing.assert_allclose(tf[0][1][:2, :2], np.linalg.inv(A), atol=2e-3)
    assert np.array_equal(tf[0][1][2], [0, 0, 1])
    allM2, num2, tf2 = im.imageMatching(dict(inp, imageMatchingMethod="mlesac", maxDistance=2.0, maxIter=1000), n, kp, matches, seed=3)
    assert num2[0, 1] >= 295
    np.testing.assert_allclose(tf2[0][1][:2, :2], np.linalg.inv(A), atol=2e-3)
    with pytest.raises(ValueError):
        im.estimateTransformationRANSAC(kp[0], kp[1], "perspective", INP)


# ---- estimateTransformationMLESAC for every transformationType ----------------------------------------------------------
from test_mlesac_types_oracle import TYPES as ML_TYPES, scene as ml_scene  # noqa: E402

ML_INP = {"maxDistance": 2.0, "inliersConfidence": 99.9, "maxIter": 1000, "imageMatchingMethod": "mlesac"}


@pytest.mark.parametrize("tform", ML_TYPES)
@pytest.mark.parametrize("m,outliers,noise", [(4, 0.0, 0.0), (6, 0.0, 0.2), (40, 0.3, 0.2), (500, 0.35, 0.3),
                                              (3001, 0.6, 0.4), (257, 0.0, 0.0)])
def test_mlesac_whole_loop_bit_exact(im, tform, m, outliers, noise):
    rng = np.random.default_rng(90 + m)
    p1, p2 = ml_scene(tform, rng, n=m, outliers=outliers, noise=noise)
    s = im.draw_samples([m], 1064, seed=m)[0]
    H, mask, found = im.estimateTransformationMLESAC(p1, p2, tform, ML_INP, sample_idx=s)
    oH, omask, ofound, _ = oracle.mlesac_tform(tform, p1, p2, s, 2.0, 99.9, 1000)
    assert found == ofound
    assert np.array_equal(mask, omask)
    if found:
        assert np.array_equal(bits(H), bits(oH))
        if m >= 40:
            assert mask.sum() >= 0.8 * (1 - outliers) * m


@pytest.mark.parametrize("tform", ML_TYPES)
def test_mlesac_tiny_inputs_and_batch(im, tform):
    rng = np.random.default_rng(95)
    sizes = [0, 1, 2, 3, 4, 5, 64, 65, 900]
    worlds = [ml_scene(tform, rng, n=m, outliers=0.25 if m > 10 else 0.0) for m in sizes]
    src = np.concatenate([w[0] for w in worlds])
    dst = np.concatenate([w[1] for w in worlds])
    ptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    samples = im.draw_samples(sizes, 1064, seed=6)
    models, mask, found, ninl = im.ransac_batch(src, dst, ptr, samples, dict(ML_INP, transformationType=tform))
    for p, (p1, p2) in enumerate(worlds):
        oH, omask, ofound, _ = oracle.mlesac_tform(tform, p1, p2, samples[p], 2.0, 99.9, 1000)
        assert bool(found[p]) == ofound, (tform, sizes[p])
        assert np.array_equal(mask[ptr[p]:ptr[p + 1]].astype(bool), omask), (tform, sizes[p])
        if ofound:
            assert np.array_equal(bits(models[p]), bits(oH)) and ninl[p] == omask.sum()
        if sizes[p] >= 1:  # the single-pair entry agrees with the batch
            H1, mask1, found1 = im.estimateTransformationMLESAC(p1, p2, tform, ML_INP, sample_idx=samples[p])
            assert found1 == ofound and np.array_equal(mask1, omask)


# ---- adversarial inputs: both estimators, every type, device == oracle ------------------------------------------------------
def _adversarial_sets():
    rng = np.random.default_rng(123)
    base = rng.uniform(0, 1000, (120, 2))
    out = {}
    p2 = base + [7.0, -3.0]
    nan1 = base.copy()
    nan1[::5] = np.nan  # a fifth of the source points missing
    out["nan_points"] = (nan1, p2)
    inf2 = p2.copy()
    inf2[3] = [np.inf, 1.0]
    out["inf_point"] = (base, inf2)
    out["all_identical"] = (np.full((50, 2), 5.0), np.full((50, 2), 9.0))
    t = np.linspace(0, 900, 80)
    line = np.stack([t, 0.5 * t + 20], 1)
    out["collinear"] = (line, line + [4.0, 4.0])
    out["huge_coordinates"] = (base * 1e7, base * 1e7 + 3e7)
    dup = np.repeat(base[:30], 4, axis=0)
    out["duplicates"] = (dup, dup * 1.01 + [2, 2])
    out["pure_noise"] = (rng.uniform(0, 1e3, (90, 2)), rng.uniform(0, 1e3, (90, 2)))
    return out


@pytest.mark.parametrize("name", sorted(_adversarial_sets()))
@pytest.mark.parametrize("tform", ML_TYPES)
def test_adversarial_inputs_agree_with_the_oracle(im, tform, name):
    p1, p2 = _adversarial_sets()[name]
    m = len(p1)
    s = im.draw_samples([m], 1064, seed=17)[0]
    with np.errstate(all="ignore"):
        H, mask, found = im.estimateTransformationRANSAC(p1, p2, tform, dict(INP, maxIter=300), sample_idx=s[:364])
        oH, omask, ofound, _ = oracle.ransac_tform(tform, p1, p2, s[:364], 3.0, 99.9, 300)
    assert found == ofound and np.array_equal(mask, omask), ("ransac", tform, name)
    if found:
        assert np.array_equal(bits(H), bits(oH)), ("ransac", tform, name)
    with np.errstate(all="ignore"):
        H, mask, found = im.estimateTransformationMLESAC(p1, p2, tform, ML_INP, sample_idx=s)
        oH, omask, ofound, _ = oracle.mlesac_tform(tform, p1, p2, s, 2.0, 99.9, 1000)
    assert found == ofound and np.array_equal(mask, omask), ("mlesac", tform, name)
    if found:
        assert np.array_equal(bits(H), bits(oH)), ("mlesac", tform, name)
