"""CPU tests of the oracle's non-projective transformTypes (estimateTransformationRANSAC.m:227-452, :483-497).

The oracle replaces MATLAB's svd / pinv-by-svd / median by closed forms (oracle/ransac_oracle.c, second header).  Here
the reference's formulas are transcribed LITERALLY to numpy - numpy.linalg.svd (LAPACK) where the reference calls svd,
numpy.median where it calls median, numpy.linalg.solve for `T2 \\` - and the two must agree to rounding; known-answer
cases pin the conventions (including the reference's quirks: a minimal rigid sample never rotates, and the rotation of
'similarity' / 'rigid' comes out transposed)."""
import numpy as np
import pytest

import oracle

TYPES = ["affine", "similarity", "rigid", "translation"]


# ---- literal numpy transcription of the reference ------------------------------------------------------------------
def ref_normalize(p):  # :579-610
    c = p.mean(axis=0)
    s = 1.0 / np.mean(np.sqrt(((p - c) ** 2).sum(axis=1)))
    T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
    return (T @ np.c_[p, np.ones(len(p))].T).T[:, :2], T


def ref_affine(p1, p2):  # :227-288
    a, T1 = ref_normalize(p1)
    b, T2 = ref_normalize(p2)
    n = len(a)
    P = np.c_[a, np.ones(n)]
    A = np.block([[P, np.zeros((n, 3))], [np.zeros((n, 3)), P]])
    rhs = np.r_[b[:, 0], b[:, 1]]
    U, s, Vt = np.linalg.svd(A, full_matrices=False)
    s = np.where(s < 1e-10 * s[0], 0.0, s)
    s_inv = np.where(s > 0, 1.0 / np.where(s > 0, s, 1.0), 0.0)
    h = Vt.T @ (np.diag(s_inv) @ (U.T @ rhs))
    Hn = np.array([[h[0], h[1], h[2]], [h[3], h[4], h[5]], [0, 0, 1.0]])
    H = np.linalg.solve(T2, Hn) @ T1
    H[2] = [0, 0, 1]
    return H


def ref_rotation(M):  # [U,~,V] = svd(M); R = V * [1 0; 0 det(V*U')] * U'
    U, S, Vt = np.linalg.svd(M)
    V = Vt.T
    return V @ np.diag([1.0, np.linalg.det(V @ U.T)]) @ U.T, S


def ref_similarity(p1, p2):  # :290-356
    a, T1 = ref_normalize(p1)
    b, T2 = ref_normalize(p2)
    c1, c2 = a.mean(axis=0), b.mean(axis=0)
    ac, bc = a - c1, b - c2
    R, _ = ref_rotation(bc.T @ ac)
    scales = [np.linalg.norm(bc, "fro") / np.linalg.norm(ac, "fro")]
    ra, rb = np.sqrt((ac ** 2).sum(axis=1)), np.sqrt((bc ** 2).sum(axis=1))
    ok = ra > 1e-10
    if ok.any():
        scales.append(np.median(rb[ok] / ra[ok]))
    s = np.median(scales)
    t = c2 - s * R @ c1
    Hn = np.eye(3)
    Hn[:2, :2] = s * R
    Hn[:2, 2] = t
    H = np.linalg.solve(T2, Hn) @ T1
    H[2] = [0, 0, 1]
    return H


def ref_rigid(p1, p2):  # :358-421
    a, T1 = ref_normalize(p1)
    b, T2 = ref_normalize(p2)
    c1, c2 = a.mean(axis=0), b.mean(axis=0)
    ac, bc = a - c1, b - c2
    R, S = ref_rotation(bc.T @ ac)
    if S[0] / max(S[1], np.finfo(float).eps) > 1e6:
        R = np.eye(2)
    Ur, _, Vrt = np.linalg.svd(R)
    R = Ur @ Vrt
    t = c2 - R @ c1
    Hn = np.eye(3)
    Hn[:2, :2] = R
    Hn[:2, 2] = t
    H = np.linalg.solve(T2, Hn) @ T1
    H[2] = [0, 0, 1]
    return H


def ref_translation(p1, p2):  # :423-452
    d = p2 - p1
    return np.array([[1, 0, np.median(d[:, 0])], [0, 1, np.median(d[:, 1])], [0, 0, 1.0]])


REF_FIT = {"affine": ref_affine, "similarity": ref_similarity, "rigid": ref_rigid, "translation": ref_translation}


def ref_find_inliers(tform, H, p1, p2, thr):  # :444-516
    h1 = np.c_[p1, np.ones(len(p1))]
    tr = (H @ h1.T).T
    tr = tr / tr[:, 2:3]
    err = np.sqrt(((p2 - tr[:, :2]) ** 2).sum(axis=1))
    if tform == "translation":
        scale = max(np.abs(p1).max(), np.abs(p2).max(), 1.0)
        err, thr = err / scale, thr / scale
    err[~np.isfinite(err)] = np.inf
    inl = err < thr
    if tform == "affine" and inl.sum() >= 3:
        q = p1[inl] - p1[inl].mean(axis=0)
        sv = np.linalg.svd(q, compute_uv=False)
        if sv[1] / sv[0] < 1e-3:
            inl[:] = False
            err[:] = np.inf
    return inl, err


def ref_check_model(H):  # :518-535
    return bool(np.all(np.isfinite(H)) and 1.0 / np.linalg.cond(H, 1) > np.finfo(float).eps and abs(np.linalg.det(H)) > np.finfo(float).eps)


def ref_ransac(tform, p1, p2, samples, max_distance, confidence, max_iter):  # :54-183 on explicit draws
    k = oracle.tform_min_points(tform)
    m = len(p1)
    if m < k:
        return None, np.zeros(m, bool), False
    fit = REF_FIT[tform]
    max_trials, max_skip, trial, skip, it = max_iter, max_iter * 10, 1, 0, 0
    best_inl, best_H, best_err = np.zeros(m, bool), None, np.inf
    while trial <= max_trials and skip < max_skip and it < len(samples):
        sel = samples[it, :k].astype(int) - 1
        it += 1
        with np.errstate(all="ignore"):
            H = fit(p1[sel], p2[sel])
        if not ref_check_model(H):
            skip += 1
            continue
        inl, err = ref_find_inliers(tform, H, p1, p2, max_distance)
        n = int(inl.sum())
        if n >= k:
            me = err[inl].mean()
            if n > best_inl.sum() or (n == best_inl.sum() and me < best_err):
                best_inl, best_H, best_err = inl, H, me
                ratio = n / m
                if ratio > 0:
                    with np.errstate(divide="ignore"):
                        max_trials = min(max_trials, np.ceil(np.log(1 - confidence / 100) / np.log(1 - ratio ** k)))
        trial += 1
    if best_inl.sum() >= k:
        H = fit(p1[best_inl], p2[best_inl])
        if ref_check_model(H):
            inl, _ = ref_find_inliers(tform, H, p1, p2, max_distance)
            if inl.sum() >= k:
                return H, inl, True
        return best_H, best_inl, True
    return best_H, best_inl, False


# ---- data ---------------------------------------------------------------------------------------------------------------
def world(tform, rng, n=300, outliers=0.3, noise=0.4):
    p1 = rng.uniform(0, 1500, (n, 2))
    th = rng.uniform(-0.15, 0.15)  # small: the reference's similarity / rigid only work near zero rotation
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    L = {"affine": np.array([[1.05, 0.12], [-0.08, 0.93]]), "similarity": 1.0 * R, "rigid": R,
         "translation": np.eye(2)}[tform]
    if tform in ("similarity", "rigid"):
        L = np.eye(2)  # see test_rotation_comes_out_transposed
    p2 = p1 @ L.T + rng.uniform(-200, 200, 2) + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    p2[bad] = rng.uniform(0, 1500, (int(bad.sum()), 2))
    return p1, p2


def draws(rng, m, n_samples):
    return np.stack([rng.permutation(m)[:4] + 1 for _ in range(n_samples)]).astype(np.uint32)


# ---- the fits against the transcription ----------------------------------------------------------------------------------
@pytest.mark.parametrize("tform", TYPES)
@pytest.mark.parametrize("n", ["minimal", 7, 400])
def test_fit_equals_the_literal_transcription(tform, n):
    rng = np.random.default_rng(sum(map(ord, tform + str(n))))
    k = oracle.tform_min_points(tform)
    for _ in range(20):
        p1, p2 = world(tform, rng, n=500, outliers=0.0, noise=2.0)
        if tform in ("similarity", "rigid"):  # exercise real rotations and scales in the formulas themselves
            th = rng.uniform(-3, 3)
            R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
            p2 = p1 @ (rng.uniform(0.5, 2) * R).T + rng.normal(0, 2.0, p1.shape)
        sel = rng.permutation(500)[: (k if n == "minimal" else n)]
        H, ok = oracle.fit_tform(tform, p1, p2, sel)
        ref = REF_FIT[tform](p1[sel], p2[sel])
        assert ok
        np.testing.assert_allclose(H, ref, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max()))


def test_exact_recovery_and_the_reference_quirks():
    rng = np.random.default_rng(3)
    p1 = rng.uniform(0, 1000, (60, 2))
    A, t = np.array([[1.1, 0.2], [-0.1, 0.9]]), np.array([5.0, -7.0])
    H, _ = oracle.fit_tform("affine", p1, p1 @ A.T + t, np.arange(60))
    np.testing.assert_allclose(H, np.r_[np.c_[A, t], [[0, 0, 1]]], atol=1e-9)
    H, _ = oracle.fit_tform("affine", p1, p1 @ A.T + t, np.array([3, 17, 41]))  # minimal: exact interpolation
    np.testing.assert_allclose(H, np.r_[np.c_[A, t], [[0, 0, 1]]], atol=1e-8)
    H, _ = oracle.fit_tform("translation", p1, p1 + [12.5, -3.0], np.arange(60))
    np.testing.assert_array_equal(H, [[1, 0, 12.5], [0, 1, -3.0], [0, 0, 1]])
    # median, not mean: 29 of 60 displaced points do not move the estimate
    p2 = p1 + [12.5, -3.0]
    p2[:29] += 100
    H, _ = oracle.fit_tform("translation", p1, p2, np.arange(60))
    np.testing.assert_allclose(H[:2, 2], [12.5, -3.0], atol=1e-12)
    # even count: MATLAB's a + (b - a)/2
    H, _ = oracle.fit_tform("translation", np.zeros((4, 2)), np.array([[1, 5], [2, 6], [4, 8], [9, 7.0]]), np.arange(4))
    assert H[0, 2] == 3.0 and H[1, 2] == 6.5


def test_rotation_comes_out_transposed():
    """estimateSimilarity / estimateRigid build R = V*D*U' from svd(pts2c' * pts1c): that is the TRANSPOSE of the
    least-squares rotation (Kabsch has U*D*V' for this matrix).  The oracle restates the reference, so a pure rotation
    by theta is answered with a rotation by -theta; at theta = 0 (and for pure scale / translation) both agree."""
    rng = np.random.default_rng(4)
    p1 = rng.uniform(-500, 500, (80, 2))
    th = 0.4
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    for tform, s in (("similarity", 1.7), ("rigid", 1.0)):
        H, _ = oracle.fit_tform(tform, p1, p1 @ (s * R).T, np.arange(80))
        np.testing.assert_allclose(H[:2, :2], s * R.T, atol=1e-9)
        np.testing.assert_allclose(H[:2, :2], REF_FIT[tform](p1, p1 @ (s * R).T)[:2, :2], atol=1e-9)
        H, _ = oracle.fit_tform(tform, p1, s * p1 + [3, 4], np.arange(80))
        np.testing.assert_allclose(H, [[s, 0, 3], [0, s, 4], [0, 0, 1]], atol=1e-9)


def test_a_minimal_rigid_sample_never_rotates():
    """Two centred points are negatives of each other, their 2x2 cross-covariance has rank one, the condition test
    (:397-399) fires and R = eye(2): inside the loop 'rigid' is a translation times the ratio of the two normalisation
    scales.  Only the refit on the inliers can rotate."""
    rng = np.random.default_rng(5)
    p1 = rng.uniform(0, 1000, (10, 2))
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    p2 = p1 @ R.T + [10, 20]
    H, ok = oracle.fit_tform("rigid", p1, p2, np.array([2, 7]))
    assert ok and H[0, 1] == 0 and H[1, 0] == 0 and H[0, 0] == H[1, 1]
    np.testing.assert_allclose(H, ref_rigid(p1[[2, 7]], p2[[2, 7]]), atol=1e-9)


def test_degenerate_samples_are_rejected():
    p1 = np.array([[0, 0], [1, 1], [2, 2], [5, 1.0]])
    p2 = p1 + 3
    H, ok = oracle.fit_tform("affine", p1, p2, np.array([0, 1, 2]))  # collinear: rank-deficient pseudo-inverse
    assert not (ok and oracle.check_model(H))
    assert not ref_check_model(ref_affine(p1[:3], p2[:3]))
    H, ok = oracle.fit_tform("similarity", p1, p2, np.array([1, 1]))  # coincident points: no scale
    assert not ok
    H, ok = oracle.fit_tform("translation", p1, np.full((4, 2), np.nan), np.array([0]))
    assert not ok


# ---- findInliers -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tform", TYPES)
def test_find_inliers_equals_the_transcription(tform):
    rng = np.random.default_rng(11)
    p1, p2 = world(tform, rng, n=700)
    k = oracle.tform_min_points(tform)
    Hs = []
    for _ in range(30):
        sel = rng.permutation(700)[:k]
        H, ok = oracle.fit_tform(tform, p1, p2, sel)
        assert ok
        Hs.append(H)
    n, me, mask = oracle.ransac_score_tform(tform, np.stack(Hs), p1, p2, 3.0)
    for t, H in enumerate(Hs):
        inl, err = ref_find_inliers(tform, H, p1, p2, 3.0)
        assert n[t] == inl.sum() and np.array_equal(mask[t].astype(bool), inl)
        if inl.any():
            np.testing.assert_allclose(me[t], err[inl].mean(), rtol=1e-9, atol=1e-9)


def test_affine_inliers_on_a_line_are_degenerate():
    x = np.linspace(0, 100, 40)
    p1 = np.c_[x, 2 * x + 1]
    p1 = np.r_[p1, [[50.0, 500.0]]]
    p2 = p1 + [5, 5]
    p2[-1] += 400  # the one point off the line is an outlier
    H = np.array([[1, 0, 5], [0, 1, 5], [0, 0, 1.0]])
    n, me, mask = oracle.ransac_score_tform("affine", H[None], p1, p2, 1.0)
    assert n[0] == 0 and not mask.any() and np.isnan(me[0])
    n, _, mask = oracle.ransac_score_tform("similarity", H[None], p1, p2, 1.0)  # no such test for the other types (:506)
    assert n[0] == 40


def test_translation_threshold_is_scaled_on_both_sides():
    p1 = np.array([[0, 0], [2000, 1000.0], [10, 10]])
    p2 = p1 + [[0, 0.9], [0, 1.1], [0, 0.99999]]
    n, _, mask = oracle.ransac_score_tform("translation", np.eye(3)[None], p1, p2, 1.0)
    assert mask[0].tolist() == [1, 0, 1]


# ---- the loop -------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tform", TYPES)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_loop_equals_the_transcription(tform, seed):
    rng = np.random.default_rng(100 + seed)
    p1, p2 = world(tform, rng, n=250 + 50 * seed, outliers=0.25 + 0.1 * seed)
    s = draws(rng, len(p1), 600)
    H, mask, found, used = oracle.ransac_tform(tform, p1, p2, s, max_distance=3.0, confidence=99.9, max_iter=500)
    rH, rmask, rfound = ref_ransac(tform, p1, p2, s, 3.0, 99.9, 500)
    assert found and rfound
    assert np.array_equal(mask, rmask)
    np.testing.assert_allclose(H, rH, rtol=1e-8, atol=1e-8 * np.abs(rH).max())
    assert mask.sum() > 0.5 * len(p1) * (1 - 0.25 - 0.1 * seed)


@pytest.mark.parametrize("tform", TYPES)
def test_too_few_points_and_no_consensus(tform):
    k = oracle.tform_min_points(tform)
    rng = np.random.default_rng(7)
    p = rng.uniform(0, 100, (k - 1, 2))
    H, mask, found, used = oracle.ransac_tform(tform, p, p, np.ones((5, 4), np.uint32))
    assert not found and used == 0 and np.isnan(H).all()
    if tform != "translation":  # unrelated point sets: at most the sample itself agrees
        p1, p2 = rng.uniform(0, 1e4, (40, 2)), rng.uniform(0, 1e4, (40, 2))
        H, mask, found, used = oracle.ransac_tform(tform, p1, p2, draws(rng, 40, 200), max_distance=0.5)
        assert mask.sum() <= k + 1
