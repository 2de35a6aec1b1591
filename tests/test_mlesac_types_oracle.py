"""CPU tests of the oracle's MLESAC for every transformationType (estimateTransformationMLESAC.m:94-254, :345-598):
the closed forms / Jacobi null vectors of oracle/ransac_oracle.c against a LITERAL numpy transcription of the reference
(numpy.linalg.svd where it calls svd, numpy.hypot, numpy.linalg.solve for `normMatrix2 \\`), plus known answers."""
import numpy as np
import pytest

import oracle
from test_ransac_types_oracle import draws

TYPES = ["projective", "affine", "similarity", "rigid", "translation"]
EPS = np.finfo(float).eps


# ---- literal numpy transcription ----------------------------------------------------------------------------------------
def ref_hz(p):  # normalizePointsHartleyZisserman (:640-690) for 2 x n input without a homogeneous row
    c = p.mean(axis=0)
    q = p - c
    md = np.mean(np.sqrt((q ** 2).sum(axis=1)))
    s = np.sqrt(2.0) / md if md > 0 else 1.0
    T = np.diag([s, s, 1.0])
    T[:2, 2] = -s * c
    return q * s, T


def ref_denorm(T, N1, N2):  # :705-716
    T = np.linalg.solve(N2, T) @ N1
    return T / T[2, 2]


def ref_null(A):  # [~,~,V] = svd(A, 0); V(:, end) - for fewer rows than columns MATLAB's economy form is the full one
    return np.linalg.svd(A, full_matrices=A.shape[0] < A.shape[1])[2][-1]


def ref_projective(p1, p2):  # :345-387
    a, N1 = ref_hz(p1)
    b, N2 = ref_hz(p2)
    n = len(a)
    A = np.zeros((2 * n, 9))
    A[0::2] = np.c_[np.zeros((n, 3)), -a, -np.ones(n), a[:, 0] * b[:, 1], a[:, 1] * b[:, 1], b[:, 1]]
    A[1::2] = np.c_[a, np.ones(n), np.zeros((n, 3)), -a[:, 0] * b[:, 0], -a[:, 1] * b[:, 0], -b[:, 0]]
    h = ref_null(A)
    return ref_denorm(h.reshape(3, 3) / h[8], N1, N2)


def ref_affine(p1, p2):  # :389-424
    a, N1 = ref_hz(p1)
    b, N2 = ref_hz(p2)
    n = len(a)
    A = np.zeros((2 * n, 7))
    A[0::2] = np.c_[np.zeros((n, 3)), -a, -np.ones(n), b[:, 1]]
    A[1::2] = np.c_[a, np.ones(n), np.zeros((n, 3)), -b[:, 0]]
    h = ref_null(A)
    T = np.eye(3)
    T[:2] = h[:6].reshape(2, 3) / h[6]
    return ref_denorm(T, N1, N2)


def ref_similarity(p1, p2):  # :426-458
    a, N1 = ref_hz(p1)
    b, N2 = ref_hz(p2)
    n = len(a)
    A = np.zeros((2 * n, 5))
    A[0::2] = np.c_[-a[:, 1], a[:, 0], np.zeros(n), -np.ones(n), b[:, 1]]
    A[1::2] = np.c_[a, np.ones(n), np.zeros(n), -b[:, 0]]
    h = ref_null(A)
    T = np.eye(3)
    T[:2] = np.array([[h[0], h[1], h[2]], [-h[1], h[0], h[3]]]) / h[4]
    return ref_denorm(T, N1, N2)


def ref_rigid(p1, p2):  # :460-490
    c1, c2 = p1.mean(axis=0), p2.mean(axis=0)
    Cm = (p1 - c1).T @ (p2 - c2)
    U, _, Vt = np.linalg.svd(Cm)
    V = Vt.T
    R = V @ np.diag([1.0, np.sign(np.linalg.det(U @ V.T))]) @ U.T
    T = np.eye(3)
    T[:2, :2] = R
    T[:2, 2] = c2 - R @ c1
    return T


def ref_translation(p1, p2):  # :492-510
    T = np.eye(3)
    T[:2, 2] = (p2 - p1).mean(axis=0)
    return T


REF_FIT = {"projective": ref_projective, "affine": ref_affine, "similarity": ref_similarity, "rigid": ref_rigid,
           "translation": ref_translation}


def ref_eval(tform, T, p1, p2, thr):  # evaluateModel (:258-295) over evaluateTransform2d / evaluateTranslation2d
    if tform == "translation":
        tp = p1 + T[:2, 2]
        d = np.sqrt((tp[:, 0] - p2[:, 0]) ** 2 + (tp[:, 1] - p2[:, 1]) ** 2)
    else:
        ph = (T @ np.c_[p1, np.ones(len(p1))].T).T
        with np.errstate(all="ignore"):
            pt = ph[:, :2] / ph[:, 2:3]
        d = np.hypot(pt[:, 0] - p2[:, 0], pt[:, 1] - p2[:, 1])
        d[np.abs(ph[:, 2]) < EPS] = np.inf
    d = np.minimum(d, thr)
    return d, d.sum()


def ref_loop_number(k, conf, n, inl):  # vision.internal.ransac.computeLoopNumber as the oracle restates it
    pr = (inl / n) ** k
    if pr < EPS:
        return 2 ** 31 - 1
    with np.errstate(divide="ignore"):
        v = np.ceil(np.log10(1 - 0.01 * conf) / np.log10(1 - pr))
    return 2 ** 31 - 1 if not v < 2 ** 31 - 1 else max(int(v), 0)


def ref_mlesac(tform, p1, p2, samples, thr, conf, max_trials):
    k = oracle.tform_min_points(tform)
    m = len(p1)
    if m < k:
        return None, np.zeros(m, bool), False
    fit = REF_FIT[tform]
    idx, skip, it, num_trials = 1, 0, 0, max_trials
    best_dis, best_inl, best_T = thr * m, np.zeros(m, bool), None
    while idx <= num_trials and skip < 10000 and it < len(samples):
        sel = samples[it, :k].astype(int) - 1
        it += 1
        with np.errstate(all="ignore"):
            T = fit(p1[sel], p2[sel])
        if not np.all(np.isfinite(T)):
            skip += 1
            continue
        d, acc = ref_eval(tform, T, p1, p2, thr)
        if acc < best_dis:
            best_dis, best_inl, best_T = acc, d < thr, T
            num_trials = min(num_trials, ref_loop_number(k, conf, m, int((d < thr).sum())))
        idx += 1
    if best_T is None or best_inl.sum() < k:
        return None, np.zeros(m, bool), False
    T = fit(p1[best_inl], p2[best_inl])
    d, _ = ref_eval(tform, T, p1, p2, thr)
    if not np.all(np.isfinite(T)) or not (d < thr).any():
        return None, np.zeros(m, bool), False
    return T, d < thr, True


def scene(tform, rng, n=300, outliers=0.3, noise=0.3):
    p1 = rng.uniform(0, 1500, (n, 2))
    th = rng.uniform(-0.5, 0.5)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    h1 = np.c_[p1, np.ones(n)]
    if tform == "projective":
        Hm = np.array([[1.02, 0.03, 20], [-0.02, 0.98, -10], [1e-5, -2e-5, 1]])
        q = (Hm @ h1.T).T
        p2 = q[:, :2] / q[:, 2:3]
    else:
        L = {"affine": np.array([[1.05, 0.12], [-0.08, 0.93]]), "similarity": 1.2 * R, "rigid": R, "translation": np.eye(2)}[tform]
        p2 = p1 @ L.T + rng.uniform(-200, 200, 2)
    p2 = p2 + rng.normal(0, noise, (n, 2))
    bad = rng.random(n) < outliers
    p2[bad] = rng.uniform(0, 1500, (int(bad.sum()), 2))
    return p1, p2


@pytest.mark.parametrize("tform", TYPES)
@pytest.mark.parametrize("n", ["minimal", 9, 400])
def test_fit_equals_the_literal_transcription(tform, n):
    rng = np.random.default_rng(sum(map(ord, tform + str(n))))
    k = oracle.tform_min_points(tform)
    for _ in range(20):
        p1, p2 = scene(tform, rng, n=500, outliers=0.0, noise=1.5)
        sel = rng.permutation(500)[: (k if n == "minimal" else n)]
        H, ok = oracle.fit_tform_mlesac(tform, p1, p2, sel)
        ref = REF_FIT[tform](p1[sel], p2[sel])
        assert ok
        np.testing.assert_allclose(H, ref, rtol=2e-8, atol=2e-8 * max(1.0, np.abs(ref).max()))


def test_known_answers():
    rng = np.random.default_rng(1)
    p1 = rng.uniform(0, 1000, (40, 2))
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    H, _ = oracle.fit_tform_mlesac("rigid", p1, p1 @ R.T + [3, 4], np.arange(40))  # Kabsch's order here: no transposition
    np.testing.assert_allclose(H, np.r_[np.c_[R, [3, 4]], [[0, 0, 1]]], atol=1e-10)
    H, _ = oracle.fit_tform_mlesac("rigid", p1, p1 @ R.T + [3, 4], np.array([5, 11]))  # and a 2-point sample rotates
    np.testing.assert_allclose(H[:2, :2], R, atol=1e-9)
    H, _ = oracle.fit_tform_mlesac("similarity", p1, p1 @ (2 * R).T, np.array([5, 11]))
    np.testing.assert_allclose(H[:2, :2], 2 * R, atol=1e-9)
    p2 = p1 + [10.0, -2.0]
    p2[0] += [40, 0]  # the MEAN (not the median): one displaced point of 40 moves the estimate by 1
    H, _ = oracle.fit_tform_mlesac("translation", p1, p2, np.arange(40))
    np.testing.assert_allclose(H[:2, 2], [11.0, -2.0], atol=1e-12)
    _, ok = oracle.fit_tform_mlesac("similarity", p1, p2, np.array([3, 3]))  # coincident sample: no finite model
    assert not ok


@pytest.mark.parametrize("tform", TYPES)
def test_eval_equals_the_transcription(tform):
    rng = np.random.default_rng(2)
    p1, p2 = scene(tform, rng, n=600)
    k = oracle.tform_min_points(tform)
    for _ in range(10):
        H, ok = oracle.fit_tform_mlesac(tform, p1, p2, rng.permutation(600)[:k])
        assert ok
        acc, n, mask = oracle.mlesac_eval_tform(tform, H, p1, p2, 2.0)
        d, racc = ref_eval(tform, H, p1, p2, 2.0)
        assert n == (d < 2.0).sum() and np.array_equal(mask, d < 2.0)
        np.testing.assert_allclose(acc, racc, rtol=1e-12)


@pytest.mark.parametrize("tform", TYPES)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_loop_equals_the_transcription(tform, seed):
    rng = np.random.default_rng(200 + seed)
    p1, p2 = scene(tform, rng, n=250 + 40 * seed, outliers=0.2 + 0.15 * seed)
    s = draws(rng, len(p1), 1100)
    H, mask, found, used = oracle.mlesac_tform(tform, p1, p2, s, 2.0, 99.9, 1000)
    rH, rmask, rfound = ref_mlesac(tform, p1, p2, s, 2.0, 99.9, 1000)
    assert found and rfound
    assert np.array_equal(mask, rmask)
    np.testing.assert_allclose(H, rH, rtol=1e-7, atol=1e-7 * np.abs(rH).max())
    assert mask.sum() > 0.85 * len(p1) * (1 - 0.2 - 0.15 * seed)
    if tform == "projective":  # the all-types entry equals the projective-only one
        H2, mask2, found2, used2 = oracle.mlesac_homography(p1, p2, s, 2.0, 99.9, 1000)
        assert np.array_equal(H, H2) and np.array_equal(mask, mask2) and used == used2


@pytest.mark.parametrize("tform", TYPES)
def test_too_few_points(tform):
    k = oracle.tform_min_points(tform)
    p = np.random.default_rng(3).uniform(0, 100, (k - 1, 2))
    H, mask, found, used = oracle.mlesac_tform(tform, p, p, np.ones((5, 4), np.uint32))
    assert not found and used == 0 and np.isnan(H).all()
