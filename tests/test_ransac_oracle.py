"""CPU tests pinning the RANSAC oracle with analytic cases."""
from importlib import import_module

import numpy as np

import oracle


def make_scene(rng, m, n_out, H, noise=0.0, w=4000.0, h=2000.0):
    p1 = np.stack([rng.uniform(0, w, m), rng.uniform(0, h, m)], 1)
    q = np.c_[p1, np.ones(m)] @ H.T
    p2 = q[:, :2] / q[:, 2:3] + noise * rng.standard_normal((m, 2))
    out = rng.permutation(m)[:n_out]
    p2[out] = np.stack([rng.uniform(0, w, n_out), rng.uniform(0, h, n_out)], 1)
    truth = np.ones(m, bool)
    truth[out] = False
    return p1, p2, truth


H_TRUE = np.array([[1.02, 0.03, 120.0], [-0.02, 0.98, -60.0], [2e-6, -1e-6, 1.0]])


def samples_for(m, s, seed=0):
    import apsamd
    im = import_module(apsamd.__name__ + ".imageMatching")
    return im.draw_samples([m], s, seed)[0]


def test_fit_exact_homography_from_four_points():
    rng = np.random.default_rng(0)
    p1, p2, _ = make_scene(rng, 50, 0, H_TRUE)
    H, ok = oracle.fit_homography(p1, p2, [3, 17, 29, 41])
    assert ok
    np.testing.assert_allclose(H / H[2, 2], H_TRUE, rtol=1e-7, atol=1e-7)
    Hall, ok = oracle.fit_homography(p1, p2, np.arange(50))
    np.testing.assert_allclose(Hall / Hall[2, 2], H_TRUE, rtol=1e-8, atol=1e-8)


def test_score_matches_analytic_inliers():
    rng = np.random.default_rng(1)
    p1, p2, truth = make_scene(rng, 300, 100, H_TRUE)
    n, e, mask = oracle.ransac_score(H_TRUE[None], p1, p2, 5.5)
    assert n[0] == mask[0].sum()
    assert np.all(mask[0][truth] == 1)
    assert mask[0][~truth].sum() <= 1  # a random outlier can land within 5.5 px by chance
    assert e[0] < 1e-6


def test_collinear_inliers_trip_the_degeneracy_check():
    # all correspondences on one line: any H consistent with them has collinear inliers -> all false (:506-513)
    t = np.linspace(0, 1000, 40)
    p1 = np.stack([t, 2 * t + 5], 1)
    n, e, mask = oracle.ransac_score(np.eye(3)[None], p1, p1.copy(), 5.5)
    assert n[0] == 0 and mask.sum() == 0 and np.isnan(e[0])


def test_singular_model_is_rejected_by_check_model():
    assert oracle.check_model(np.eye(3))
    assert not oracle.check_model(np.array([[1., 2, 3], [2, 4, 6], [0, 0, 1]]))
    assert not oracle.check_model(np.array([[1., 0, 0], [0, np.nan, 0], [0, 0, 1]]))
    assert not oracle.check_model(np.diag([1e-9, 1e-9, 1.0]) * 1e-3)  # |det| <= eps


def test_whole_loop_recovers_the_model_and_the_analytic_mask():
    rng = np.random.default_rng(2)
    p1, p2, truth = make_scene(rng, 400, 150, H_TRUE, noise=0.3)
    s = samples_for(400, 564)
    H, mask, found, trials = oracle.ransac_homography(p1, p2, s, 5.5, 99.9, 500)
    assert found
    assert np.all(mask[truth])
    assert mask[~truth].sum() <= 2
    np.testing.assert_allclose(H / H[2, 2], H_TRUE, rtol=2e-3, atol=0.5)
    # adaptive stop (:125-130): inlier ratio 0.625 -> ceil(log(1e-3)/log(1-0.625^4)) = 42 trials (+ the
    # draws spent before the first all-inlier sample and on invalid models), far below maxIter
    assert trials < 200


def test_adaptive_stop_replay_is_exact_for_a_known_ratio():
    rng = np.random.default_rng(3)
    p1, p2, truth = make_scene(rng, 100, 0, H_TRUE)  # all inliers: ratio 1 -> maxTrials becomes 0 after trial 1
    s = samples_for(100, 50)
    H, mask, found, trials = oracle.ransac_homography(p1, p2, s, 5.5, 99.9, 500)
    assert found and mask.all()
    assert trials == 1


def test_too_few_points_and_no_consensus():
    rng = np.random.default_rng(4)
    p = rng.uniform(0, 100, (3, 2))
    H, mask, found, trials = oracle.ransac_homography(p, p, np.ones((5, 4), np.uint32), 5.5, 99.9, 500)
    assert not found and not mask.any() and trials == 0
    p1 = rng.uniform(0, 4000, (60, 2))
    p2 = rng.uniform(0, 4000, (60, 2))
    H, mask, found, trials = oracle.ransac_homography(p1, p2, samples_for(60, 564), 1.0, 99.9, 500)
    # pure noise: every 4-sample fits its own four points exactly, so "found" with ~4 inliers is the
    # reference's behaviour; imageMatching's ni > 8 + 0.3 nf rule is what rejects the pair
    assert mask.sum() < 8


def test_draw_samples_are_distinct_in_range_and_reproducible():
    s = samples_for(7, 2000, seed=5)
    assert s.min() >= 1 and s.max() <= 7
    assert all(len(set(r)) == 4 for r in s.tolist())
    assert np.array_equal(s, samples_for(7, 2000, seed=5))
    counts = np.bincount(s.reshape(-1), minlength=8)[1:]
    assert counts.min() > 0.8 * counts.mean()  # roughly uniform


# ---- MLESAC (estimateTransformationMLESAC.m) ---------------------------------------------------------------
def _mlesac_scene(seed=0, m=300, n_out=80, noise=0.3):
    rng = np.random.default_rng(seed)
    Ht = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    p1 = rng.uniform(0, 500, (m, 2))
    q = np.c_[p1, np.ones(m)] @ Ht.T
    p2 = q[:, :2] / q[:, 2:]
    p2[:n_out] += rng.uniform(-60, 60, (n_out, 2))
    p2 += rng.normal(0, noise, p2.shape)
    samples = np.stack([rng.choice(m, 4, replace=False) + 1 for _ in range(1200)]).astype(np.uint32)
    return Ht, p1, p2, samples


def test_mlesac_recovers_model_and_stops_adaptively():
    Ht, p1, p2, samples = _mlesac_scene()
    H, mask, found, used = oracle.mlesac_homography(p1, p2, samples, 2.0, 99.9, 1000)
    assert found and mask[80:].mean() > 0.97 and mask[:80].mean() < 0.1
    assert np.allclose(H[:, :2], Ht[:, :2], atol=2e-3) and np.allclose(H[:, 2], Ht[:, 2], atol=0.3) and H[2, 2] == 1.0
    assert used < 200  # computeLoopNumber with ~73 % inliers needs a few dozen trials


def test_mlesac_eval_is_truncated_one_way_loss():
    Ht, p1, p2, _ = _mlesac_scene(seed=1)
    acc, n, mask = oracle.mlesac_eval(Ht, p1, p2, 2.0)
    q = np.c_[p1, np.ones(len(p1))] @ Ht.T
    d = np.hypot(*(q[:, :2] / q[:, 2:] - p2).T)
    assert n == int((d < 2.0).sum()) and np.array_equal(mask, d < 2.0)
    assert abs(acc - np.minimum(d, 2.0).sum()) < 1e-9 * acc


def test_mlesac_edge_cases():
    Ht, p1, p2, samples = _mlesac_scene(seed=2)
    H, mask, found, used = oracle.mlesac_homography(p1[:3], p2[:3], samples, 2.0, 99.9, 1000)
    assert not found and used == 0 and not mask.any()
    # all draws degenerate (repeated point): every fit is non-finite -> skipped -> nothing found
    bad = np.ones((50, 4), np.uint32)
    H, mask, found, used = oracle.mlesac_homography(p1, p2, bad, 2.0, 99.9, 1000)
    assert not found and used == 50 and not mask.any()
    # the HZ-normalised fit on 4 exact correspondences reproduces the homography
    Hf, ok = oracle.fit_homography_mlesac(p1, (np.c_[p1, np.ones(len(p1))] @ Ht.T)[:, :2] /
                                          (np.c_[p1, np.ones(len(p1))] @ Ht.T)[:, 2:], [0, 50, 120, 280])
    assert ok and np.allclose(Hf, Ht, atol=1e-8)
