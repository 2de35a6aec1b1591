"""GPU parity: batched device RANSAC against the oracle — inlier masks and counts bit-exact."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_ransac_oracle import H_TRUE, make_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def im(gpu):
    return import_module(gpu.__name__ + ".imageMatching")


def test_score_bit_exact(im):
    rng = np.random.default_rng(10)
    p1, p2, truth = make_scene(rng, 1000, 400, H_TRUE, noise=0.5)
    Hs = []
    for t in range(64):
        sel = rng.permutation(1000)[:4]
        H, ok = oracle.fit_homography(p1, p2, sel)
        Hs.append(H if ok else np.eye(3))
    Hs.append(H_TRUE)
    Hs.append(np.array([[1., 2, 3], [2, 4, 6], [0, 0, 1]]))  # singular: errors inf/NaN -> 0 inliers
    Hs = np.stack(Hs)
    n, e, mask = im.ransac_score(Hs, p1, p2, 5.5)
    on, oe, omask = oracle.ransac_score(Hs, p1, p2, 5.5)
    assert np.array_equal(n, on)
    assert np.array_equal(mask, omask)
    assert np.array_equal(e.view(np.uint64), oe.view(np.uint64))
    assert n[-2] >= 590


@pytest.mark.parametrize("m,n_out,noise", [(4, 0, 0.0), (30, 10, 0.2), (400, 150, 0.3), (3000, 2000, 0.5),
                                           (257, 0, 0.0)])
def test_whole_loop_bit_exact(im, m, n_out, noise):
    rng = np.random.default_rng(20 + m)
    p1, p2, truth = make_scene(rng, m, n_out, H_TRUE, noise=noise)
    s = im.draw_samples([m], 564, seed=m)[0]
    inp = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 500}
    H, mask, found = im.estimateTransformationRANSAC(p1, p2, "projective", inp, sample_idx=s)
    oH, omask, ofound, _ = oracle.ransac_homography(p1, p2, s, 5.5, 99.9, 500)
    assert found == ofound
    assert np.array_equal(mask, omask)
    if found:
        assert np.array_equal(H.view(np.uint64), oH.view(np.uint64))
        assert mask[truth].mean() > 0.95


def test_degenerate_and_tiny_inputs(im):
    t = np.linspace(0, 1000, 40)
    line = np.stack([t, 2 * t + 5], 1)
    inp = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 100}
    s = im.draw_samples([40], 164, seed=1)[0]
    H, mask, found = im.estimateTransformationRANSAC(line, line.copy(), "projective", inp, sample_idx=s)
    oH, omask, ofound, _ = oracle.ransac_homography(line, line.copy(), s, 5.5, 99.9, 100)
    assert found == ofound and np.array_equal(mask, omask)
    H, mask, found = im.estimateTransformationRANSAC(line[:3], line[:3], "projective", inp)
    assert H is None and not found and mask.shape == (3,)
    with pytest.raises(ValueError):
        im.estimateTransformationRANSAC(line, line[:5], "projective", inp)
    with pytest.raises(ValueError):
        im.estimateTransformationRANSAC(line, line, "homography", inp)  # "Unknown transform type" (:658-659)


def test_image_matching_batch_equals_per_pair_oracle(im):
    rng = np.random.default_rng(30)
    n = 5
    kp = [np.stack([rng.uniform(0, 4000, 900), rng.uniform(0, 2000, 900)], 1) for _ in range(n)]
    matchesAll = [[None] * n for _ in range(n)]
    truth_H = {}
    for i in range(n):
        for j in range(i + 1, n):
            if (i + j) % 4 == 3:
                matchesAll[i][j] = np.zeros((0, 2))
                continue
            m = 100 + 10 * (i + j)
            # disjoint index blocks per partner so that pairs do not overwrite each other's keypoints
            a = j * 180 + rng.permutation(180)[:m] + 1
            b = i * 180 + rng.permutation(180)[:m] + 1
            H = H_TRUE.copy()
            H[0, 2] += 50 * i
            H[1, 2] -= 30 * j
            good = rng.random(m) < (0.7 if (i * j) % 3 else 0.15)
            # make image-i points the image of image-j points under H for the good ones
            q = np.c_[kp[j][b - 1], np.ones(m)] @ H.T
            proj = q[:, :2] / q[:, 2:3]
            kp[i][a[good] - 1] = proj[good] + 0.3 * rng.standard_normal((good.sum(), 2))
            matchesAll[i][j] = np.stack([a, b], 1).astype(np.float64)
            truth_H[(i, j)] = H
    inp = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 500, "mBrownLowe": 6,
           "transformationType": "projective"}
    allM, numM, tf = im.imageMatching(inp, n, kp, matchesAll, None, seed=3)
    pairs = im.candidate_pairs(matchesAll, n, 6)
    counts = [len(matchesAll[i][j]) for (i, j) in pairs if len(matchesAll[i][j]) >= 4]
    samples = im.draw_samples(counts, 564, 3)
    p = 0
    accepted = 0
    for (i, j) in pairs:
        mt = matchesAll[i][j]
        if len(mt) < 4:
            assert allM[i][j] is None
            continue
        a = mt[:, 0].astype(int) - 1
        b = mt[:, 1].astype(int) - 1
        oH, omask, ofound, _ = oracle.ransac_homography(kp[j][b], kp[i][a], samples[p], 5.5, 99.9, 500)
        p += 1
        ni = int(omask.sum()) if ofound else 0
        if ni > 8 + 0.3 * len(mt):
            accepted += 1
            assert np.array_equal(allM[i][j], mt[omask])
            assert numM[i, j] == ni
            assert np.array_equal(tf[i][j].view(np.uint64), oH.view(np.uint64))
            np.testing.assert_allclose(tf[j][i] @ tf[i][j], np.eye(3), atol=1e-8)
        else:
            assert allM[i][j] is None and numM[i, j] == 0 and tf[i][j] is None
    assert accepted >= 2


def test_device_draws_equal_host_draws(im):
    counts = [7, 3, 100, 5, 4, 20000, 123457]
    keys = [5, 9, 2, 77, 1, 0, 123456789]
    host = im.draw_samples(counts, 700, seed=11, keys=keys)
    dev = im.draw_samples_device(counts, 700, seed=11, keys=keys).cpu().numpy().astype(np.uint32)
    assert np.array_equal(host, dev)
    assert np.array_equal(im.draw_samples(counts, 64, seed=3), im.draw_samples_device(counts, 64, seed=3).cpu().numpy().astype(np.uint32))


# ---- MLESAC --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,m,n_out", [(0, 300, 80), (1, 64, 10), (2, 1000, 600), (3, 5, 0)])
def test_mlesac_bit_identical_to_oracle(gpu, seed, m, n_out):
    from test_ransac_oracle import _mlesac_scene
    im = import_module(gpu.__name__ + ".imageMatching")
    _, p1, p2, samples = _mlesac_scene(seed, m, n_out)
    inp = {"maxDistance": 2.0, "inliersConfidence": 99.9, "maxIter": 1000}
    H, mask, found = im.estimateTransformationMLESAC(p1, p2, "projective", inp, sample_idx=samples)
    oH, omask, ofound, _ = oracle.mlesac_homography(p1, p2, samples, 2.0, 99.9, 1000)
    assert found == ofound and np.array_equal(mask, omask)
    if found:
        assert np.array_equal(H.view(np.uint64), oH.view(np.uint64))
    else:
        assert H is None


def test_mlesac_degenerate_draws_and_method_switch(gpu):
    from test_ransac_oracle import _mlesac_scene
    im = import_module(gpu.__name__ + ".imageMatching")
    _, p1, p2, samples = _mlesac_scene(4)
    H, mask, found = im.estimateTransformationMLESAC(p1, p2, "projective", {}, sample_idx=np.ones((40, 4), np.uint32))
    assert not found and H is None and not mask.any()
    # the batched entry follows input.imageMatchingMethod (inputs.m:66)
    inp = {"maxDistance": 2.0, "inliersConfidence": 99.9, "maxIter": 1000, "imageMatchingMethod": "mlesac"}
    models, bmask, bfound, ninl = im.ransac_batch(p1, p2, np.array([0, len(p1)]), samples[None], inp)
    oH, omask, ofound, _ = oracle.mlesac_homography(p1, p2, samples, 2.0, 99.9, 1000)
    assert bool(bfound[0]) == ofound and np.array_equal(bmask.astype(bool), omask) and ninl[0] == omask.sum()
    assert np.array_equal(models[0].view(np.uint64), oH.view(np.uint64))
    with pytest.raises(ValueError):  # "Unknown transform type"
        im.estimateTransformationMLESAC(p1, p2, "homography", {})


def test_ransac_shards_do_not_depend_on_the_partition(gpu):
    """The multi-GPU verifier: candidate pairs go round-robin over ranks and the draws are keyed by the GLOBAL pair id
    (parallel.py step 4), so a pair's model must not depend on which other pairs share its batch."""
    im = import_module(gpu.__name__ + ".imageMatching")
    rng = np.random.default_rng(8)
    pts1, pts2, ptr = [], [], [0]
    for p in range(5):
        m = 120 + 40 * p
        a = rng.uniform(0, 800, (m, 2))
        H = np.array([[1 + 0.01 * p, 0.02, 30.0 * p], [-0.01, 1.0, -12.0], [1e-5, 0, 1.0]])
        q = np.c_[a, np.ones(m)] @ H.T
        b = q[:, :2] / q[:, 2:] + rng.normal(0, 0.4, (m, 2))
        b[: m // 4] = rng.uniform(0, 800, (m // 4, 2))
        pts1.append(a)
        pts2.append(b)
        ptr.append(ptr[-1] + m)
    cnts = np.diff(ptr)
    inp = {"maxDistance": 2.0, "inliersConfidence": 99.9, "maxIter": 500}
    keys = np.arange(100, 105)  # global pair ids
    samples = im.draw_samples(cnts, 564, seed=7, keys=keys)
    assert np.array_equal(im.draw_samples_device(cnts, 564, seed=7, keys=keys).cpu().numpy().astype(np.uint32), samples)
    models, mask, found, ninl = im.ransac_batch(np.concatenate(pts1), np.concatenate(pts2), np.asarray(ptr), samples, inp)
    assert found.all() and (ninl > 0.6 * cnts).all()
    for shard in ([0, 2, 4], [1, 3], [4], [3, 0]):
        s_ptr = np.concatenate([[0], np.cumsum(cnts[shard])])
        s_samples = im.draw_samples(cnts[shard], 564, seed=7, keys=keys[shard])
        m2, k2, f2, n2 = im.ransac_batch(np.concatenate([pts1[p] for p in shard]), np.concatenate([pts2[p] for p in shard]),
                                         s_ptr, s_samples, inp)
        for k, p in enumerate(shard):
            assert np.array_equal(m2[k].view(np.uint64), models[p].view(np.uint64)) and n2[k] == ninl[p]
            assert np.array_equal(k2[s_ptr[k]:s_ptr[k + 1]], mask[ptr[p]:ptr[p + 1]])


def test_draw_exhaustion_is_reported_and_the_drawn_form_retries(gpu, im):
    """Matches with non-finite coordinates make nearly every 4-subset unusable: the draw is skipped, and the loop only
    ends after 10*maxIter skipped draws (estimateTransformationRANSAC.m:94).  With fewer
    pre-drawn subsets the library stops where the draws stop and says so (aps_ransac_draws_exhausted);
    ransac_batch_drawn then redraws with a longer stream until the loop's own rule ends it."""
    capi = gpu._capi
    rng = np.random.default_rng(3)
    pts = rng.uniform(0, 400, (40, 2))
    pts[:36] = np.nan  # a subset that touches a NaN point cannot be fitted: the draw is skipped (:100-104)
    q = pts + 5.0
    inp = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 20}
    s = im.draw_samples([40], 84, seed=1)
    im.ransac_batch(pts, q, np.array([0, 40]), s, inp)
    assert capi.lib.aps_ransac_draws_exhausted() == 1
    s = im.draw_samples([40], 300, seed=1)  # 10 * maxIter = 200 skipped draws end the loop first
    im.ransac_batch(pts, q, np.array([0, 40]), s, inp)
    assert capi.lib.aps_ransac_draws_exhausted() == 0
    im.ransac_batch_drawn(pts, q, np.array([0, 40]), [40], inp, seed=1)
    assert capi.lib.aps_ransac_draws_exhausted() == 0
    # fewer draws than trials on an ordinary pair is reported too; the drawn form never runs out on one
    rng = np.random.default_rng(0)
    p1 = rng.uniform(0, 500, (80, 2))
    p2 = p1 + [30.0, -12.0] + rng.standard_normal((80, 2)) * 3.0
    p2[::2] = rng.uniform(0, 500, (40, 2))
    inp5 = {"maxDistance": 5.5, "inliersConfidence": 99.9, "maxIter": 500}
    im.ransac_batch(p1, p2, np.array([0, 80]), im.draw_samples([80], 3, seed=2), inp5)
    assert capi.lib.aps_ransac_draws_exhausted() == 1
    _, _, found, ninl = im.ransac_batch_drawn(p1, p2, np.array([0, 80]), [80], inp5)
    assert found[0] == 1 and ninl[0] >= 10 and capi.lib.aps_ransac_draws_exhausted() == 0


def test_gather_match_points_equals_host_indexing(gpu):
    """aps_gather_match_points (imageMatching.m:121-135 on the device): keypoints{i}(matches(:,1),:) / keypoints{j}(matches(:,2),:)
    for a work list of pairs, from per-image tables and the resident CSR match lists; out-of-table indices give NaN."""
    import torch
    from importlib import import_module

    im = import_module(gpu.__name__ + ".imageMatching")
    rng = np.random.default_rng(3)
    counts = [50, 0, 77, 31]
    kps = [rng.uniform(1, 500, (c, 2)) for c in counts]
    # three work pairs with slices scattered in longer lists
    work = [(0, 2, 40), (3, 0, 25), (2, 3, 0)]
    ia = rng.integers(1, 20, 200).astype(np.int32)
    ib = rng.integers(1, 20, 200).astype(np.int32)
    starts, wptr = [], [0]
    pos = 7
    for a, b, m in work:
        starts.append(pos)
        ia[pos:pos + m] = rng.integers(1, counts[a] + 1, m)
        ib[pos:pos + m] = rng.integers(1, counts[b] + 1, m)
        pos += m + 11
        wptr.append(wptr[-1] + m)
    ia[starts[0] + 3] = counts[0] + 5  # outside image 0's table
    kt = [torch.from_numpy(k).cuda() for k in kps]
    pa, pb = im.gather_match_points(kt, torch.from_numpy(ia).cuda(), torch.from_numpy(ib).cuda(), starts, wptr,
                                    [w[0] for w in work], [w[1] for w in work])
    pa, pb = pa.cpu().numpy(), pb.cpu().numpy()
    assert pa.shape == (2, 65) and pb.shape == (2, 65)
    for q, (a, b, m) in enumerate(work):
        for e in range(m):
            ra, rb = ia[starts[q] + e] - 1, ib[starts[q] + e] - 1
            col = wptr[q] + e
            if ra >= counts[a]:
                assert np.isnan(pa[:, col]).all()
            else:
                assert np.array_equal(pa[:, col], kps[a][ra])
            assert np.array_equal(pb[:, col], kps[b][rb])
