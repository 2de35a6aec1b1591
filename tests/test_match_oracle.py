"""CPU tests pinning the matching oracle with analytic known-answer cases.
The reference holds no golden vectors for this path, so the oracle is checked against hand-derived
answers and an independent float64 brute force."""
import numpy as np
import pytest

import oracle
from util import planted_pair, sift_like


def brute_ssd(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return (a * a).sum(1)[:, None] + (b * b).sum(1)[None, :] - 2 * a @ b.T


def test_2nn_matches_float64_bruteforce_when_gaps_are_large():
    rng = np.random.default_rng(1)
    a, b, _, _ = planted_pair(rng, 257, 300, 100)
    idx, d1, d2 = oracle.match_2nn_ssd(a, b)
    D = brute_ssd(a, b)
    order = np.argsort(D, axis=1, kind="stable")
    gap_ok = (np.take_along_axis(D, order[:, 1:2], 1) - np.take_along_axis(D, order[:, :1], 1))[:, 0] > 1e-4
    assert gap_ok.sum() > 200
    assert np.array_equal(idx[gap_ok] - 1, order[gap_ok, 0])
    np.testing.assert_allclose(d1, np.take_along_axis(D, order[:, :1], 1)[:, 0], atol=2e-6)
    np.testing.assert_allclose(d2, np.take_along_axis(D, order[:, 1:2], 1)[:, 0], atol=2e-6)


def test_planted_duplicates_and_first_index_tie_rule():
    rng = np.random.default_rng(2)
    b = sift_like(rng, 50)
    b[30] = b[10]  # exact duplicate rows -> identical distances; min() returns the FIRST (matchFeaturesScratch.m:356)
    a = b[[10, 3, 30]].copy()
    idx, d1, d2 = oracle.match_2nn_ssd(a, b)
    assert idx.tolist() == [11, 4, 11]
    # the duplicate is the runner-up at the same distance as the best (:357-358 masks one entry only);
    # the value itself is the rounding residue of a2+b2-2G, not an exact zero
    assert d1[0] == d2[0] and d1[2] == d2[2]
    assert np.all(np.abs(d1) < 1e-6)
    assert d2[1] > 1e-3


def test_single_candidate_gives_inf_second_and_no_match():
    rng = np.random.default_rng(3)
    a = sift_like(rng, 5)
    b = a[:1].copy()
    idx, d1, d2 = oracle.match_2nn_ssd(a, b)
    assert idx.tolist() == [1] * 5 and np.all(np.isinf(d2))
    m, met = oracle.match_features(a, b, 0.6, 10.0)
    assert m.shape == (0, 2)  # isfinite(dSecond) fails (:178)


def test_ratio_threshold_and_unique_greedy():
    # hand-built 4-D case embedded in 128-D: distances are exact small integers
    def vec(*xy):
        v = np.zeros(128, np.float32)
        v[:len(xy)] = xy
        return v
    B = np.stack([vec(0, 0), vec(10, 0), vec(0, 10), vec(20, 20)])
    A = np.stack([vec(1, 0),     # nearest B0 d=1, second B1 d=81      -> ratio ok
                  vec(0, 2),     # nearest B0 d=4, second B2 d=64      -> ratio ok, loses B0 to row 0
                  vec(9, 0),     # nearest B1 d=1, second B0 d=81      -> ok
                  vec(5, 0.1)])  # nearest B0 25.01 vs B1 25.01-ish    -> ratio fails
    m, met = oracle.match_features(A, B, 0.6, 100.0, unique=True, normalize=0)
    assert m.tolist() == [[1, 1], [3, 2]]  # ascending d with stable order: rows 0 and 2 tie at d=1
    assert met.tolist() == [1.0, 1.0]
    m2, met2 = oracle.match_features(A, B, 0.6, 100.0, unique=False, normalize=0)
    assert m2.tolist() == [[1, 1], [2, 1], [3, 2]] and met2.tolist() == [1.0, 4.0, 1.0]
    m3, _ = oracle.match_features(A, B, 0.6, 3.0, unique=False, normalize=0)  # MatchThreshold drops d=4
    assert m3.tolist() == [[1, 1], [3, 2]]


def test_normalisation_rule_triggers_only_for_large_magnitudes():
    rng = np.random.default_rng(4)
    a, b, _, _ = planted_pair(rng, 64, 64, 40, unit=False)   # 0..255 valued -> max > 2 -> normalised (:105)
    m_auto, met_auto = oracle.match_features(a, b, 0.8, 1.5, normalize=2)
    m_forced, met_forced = oracle.match_features(a, b, 0.8, 1.5, normalize=1)
    assert np.array_equal(m_auto, m_forced) and len(m_auto) > 10
    assert np.all(met_auto <= 1.5)
    an = oracle.normalize_rows(a)
    np.testing.assert_allclose(np.linalg.norm(an, axis=1), 1.0, atol=1e-5)


def test_knn_is_sorted_exact_and_self_first():
    rng = np.random.default_rng(5)
    x = sift_like(rng, 200)
    idx, dist = oracle.knn(x, x, 4)
    assert np.array_equal(idx[:, 0], np.arange(1, 201))  # self at distance ~0
    assert np.all(np.diff(dist, axis=1) >= 0)
    D = brute_ssd(x, x)
    np.testing.assert_allclose(np.sort(D, axis=1)[:, :4], dist, atol=2e-6)


def test_global_filter_toy_pool():
    # 3 images, 2+2+1 features; neighbours hand-written (1-based), k=4
    img = np.array([1, 1, 2, 2, 3], np.uint32)
    loc = np.array([1, 2, 1, 2, 1], np.uint32)
    nn_idx = np.array([[1, 3, 2, 5],    # q1: self, img2(d=.1), same-image, img3(d=.5) -> ratio .2 ok -> (1,2):[1 1]
                       [2, 1, 4, 0],    # q2: self, same-image, img2 only one left -> skip (<2, :140)
                       [3, 1, 5, 4],    # q3: self, img1(.1), img3(.15), same -> ratio .667 > .6 reject
                       [4, 5, 2, 3],    # q4: self, img3(.2), img1(.9), same -> ok -> pair (2,3): [2 1]
                       [5, 4, 1, 3]],   # q5: self, img2(.2), img1(.5), img2 -> ok -> pair (2,3): [2 1] again (no dedup)
                      np.uint32)
    nn_dist = np.array([[0, .1, .3, .5], [0, .2, .4, np.inf], [0, .1, .15, .2], [0, .2, .9, 1.0],
                        [0, .2, .5, .6]], np.float32)
    rows = oracle.global_filter(nn_idx, nn_dist, img, loc, 0.6)
    assert rows.tolist() == [[1, 2, 1, 1], [2, 3, 2, 1], [2, 3, 2, 1]]


def test_hamming_tie_rules_and_edges():
    A = np.array([[0b00000000, 0xFF], [0b00001111, 0x00]], np.uint8)
    B = np.array([[0b00000001, 0xFF],    # d(A0)=1
                  [0b00000010, 0xFF],    # d(A0)=1  (tie: best stays B0 (strict <), second = B1 via <=)
                  [0b00001111, 0x00]],   # d(A1)=0
                 np.uint8)
    idx, d1, d2 = oracle.hamming_2nn(A, B)
    assert idx.tolist() == [1, 3] and d1.tolist() == [1.0, 0.0]
    assert d2[0] == 1.0
    idx, d1, d2 = oracle.hamming_2nn(A, B[:1])      # single candidate -> second = nb*8 (:71-74)
    assert d2.tolist() == [16.0, 16.0]
    idx, d1, d2 = oracle.hamming_2nn(A, B[:0])      # N2 == 0 -> idx 0, NaN (:42-45)
    assert idx.tolist() == [0, 0] and np.all(np.isnan(d1)) and np.all(np.isnan(d2))


# ---- a6: nearest2ApproxFloatFast (oracle/pca_oracle.c) against an independent numpy / LAPACK evaluation --------------------
def _pca_sets(seed=3, n1=300, n2=700):
    rng = np.random.default_rng(seed)
    base = sift_like(rng, n2 + 100)
    A = np.maximum(base[:n1] + 0.02 * rng.standard_normal((n1, 128)).astype(np.float32), 0)
    A = (A / np.linalg.norm(A, axis=1, keepdims=True)).astype(np.float32)
    return A, base[50:50 + n2].copy()


def test_pca_basis_spans_lapacks_principal_subspace_with_pcas_sign_convention():
    """The oracle's cyclic-Jacobi axes against numpy.linalg.eigh of a float64 covariance: same eigenvalues, the same axes up
    to what float32 storage and the f32 covariance chain allow, every axis signed so that its largest entry is positive
    (pca's documented convention), orthonormal, ordered by descending variance."""
    _, B = _pca_sets()
    mu, coeff, cov = oracle.pca_basis(B, 48)
    assert np.allclose(mu, B.astype(np.float64).mean(0), rtol=0, atol=1e-7)
    Bc = B.astype(np.float64) - B.astype(np.float64).mean(0)
    cov64 = Bc.T @ Bc / (len(B) - 1)
    assert np.abs(cov - cov64).max() < 2e-6 * np.abs(cov64).max()
    w, V = np.linalg.eigh(cov)
    V = V[:, ::-1][:, :48]
    C = coeff.astype(np.float64)
    assert np.abs(C.T @ C - np.eye(48)).max() < 1e-6
    assert np.all(C[np.abs(C).argmax(0), np.arange(48)] > 0)
    var = np.einsum("kc,kl,lc->c", C, cov, C)
    assert np.all(np.diff(var) <= 1e-12) and np.allclose(var, w[::-1][:48], rtol=1e-5)
    # axis by axis where the spectrum is well separated, as a subspace otherwise
    gaps = np.abs(np.diff(w[::-1][:49]))
    for c in range(48):
        if min(gaps[c], gaps[c - 1] if c else np.inf) > 1e-3 * w[-1]:
            assert abs(abs(C[:, c] @ V[:, c]) - 1) < 1e-5, c
    assert np.linalg.norm(C - V @ (V.T @ C)) < 1e-4


def test_pca2nn_equals_a_numpy_evaluation_of_the_reference_steps():
    """matchFeaturesScratch.m:476-490, 552-570 step by step in numpy (float64 products, so only clear winners are compared bit
    for bit): indices agree wherever the two best similarities are separated, distances to 1e-5; UsePCA = false and a
    dimension not above ApproxNumComponents skip the projection (:478)."""
    A, B = _pca_sets(4)
    idx, d1, d2 = oracle.pca2nn(A, B, 48, True)
    mu, coeff, _ = oracle.pca_basis(B, 48)
    Ap, Bp = (A - mu).astype(np.float64) @ coeff, (B - mu).astype(np.float64) @ coeff
    eps = np.finfo(np.float32).eps
    Ap /= np.sqrt((Ap * Ap).sum(1, keepdims=True)) + eps
    Bp /= np.sqrt((Bp * Bp).sum(1, keepdims=True)) + eps
    G = Ap @ Bp.T
    order = np.argsort(-G, axis=1, kind="stable")
    s1, s2 = G[np.arange(len(A)), order[:, 0]], G[np.arange(len(A)), order[:, 1]]
    clear = s1 - s2 > 1e-5
    assert clear.mean() > 0.95 and np.array_equal(idx[clear], order[clear, 0] + 1)
    assert np.abs(d1 - (2 - 2 * s1)).max() < 1e-5 and np.abs(d2 - (2 - 2 * s2)).max() < 1e-5
    i0, e1, e2 = oracle.pca2nn(A, B, 48, False)
    i128, f1, f2 = oracle.pca2nn(A, B, 128, True)
    assert np.array_equal(i0, i128) and np.array_equal(e1, f1) and np.array_equal(e2, f2)
    An = A / (np.sqrt((A.astype(np.float64) ** 2).sum(1, keepdims=True)) + eps)
    Bn = B / (np.sqrt((B.astype(np.float64) ** 2).sum(1, keepdims=True)) + eps)
    G = An @ Bn.T
    assert np.mean(i0 == G.argmax(1) + 1) > 0.99
    # first-index rule and the twin as the second (:558-560)
    B2 = np.vstack([B[:40], B[5:6]])
    i2, g1, g2 = oracle.pca2nn(B[5:6], B2, 48, False)
    assert i2[0] == 6 and g1[0] == g2[0]
    # a single B row: the second similarity is max of an all -inf row
    i1, h1, h2 = oracle.pca2nn(A[:3], B[:1], 48, False)
    assert np.all(i1 == 1) and np.all(np.isinf(h2)) and np.all(h2 > 0)


@pytest.mark.parametrize("n2", [1, 2, 10, 48, 49])
def test_pca2nn_with_fewer_rows_than_components(n2):
    """pca(B - muB, 'NumComponents', 48) returns min(n2 - 1, 48) columns when B has few rows (matchFeaturesScratch.m:481-482;
    the centred data has rank <= n2 - 1).  The restatement keeps the other columns of the basis at zero; the result must equal
    a float64 numpy evaluation on the narrower basis (axes of B's null space are not in the reference's basis - they
    would lengthen every A row and shrink its cosines)."""
    A, B = _pca_sets(6)
    B = B[:n2]
    keep = min(n2 - 1, 48)
    mu, coeff, _ = oracle.pca_basis(B, 48)
    assert coeff.shape == (128, 48) and not coeff[:, keep:].any() and (keep == 0 or coeff[:, :keep].any(0).all())
    idx, d1, d2 = oracle.pca2nn(A, B, 48, True)
    eps = np.finfo(np.float32).eps
    Ap, Bp = (A - mu).astype(np.float64) @ coeff[:, :keep], (B - mu).astype(np.float64) @ coeff[:, :keep]
    Ap /= np.sqrt((Ap * Ap).sum(1, keepdims=True)) + eps
    Bp /= np.sqrt((Bp * Bp).sum(1, keepdims=True)) + eps
    G = Ap @ Bp.T
    s = -np.sort(-G, axis=1)
    assert np.abs(d1 - (2 - 2 * s[:, 0])).max() < 2e-5
    if n2 >= 2:
        assert np.abs(d2 - (2 - 2 * s[:, 1])).max() < 2e-5
        clear = s[:, 0] - s[:, 1] > 1e-4
        assert np.array_equal(idx[clear], G.argmax(1)[clear] + 1)
    else:
        assert np.all(d1 == 2.0) and np.all(np.isinf(d2))
    if 2 <= n2 <= 48:
        # the full-width basis of rounds 1-5 (axes of the null space included) is measurably different: the test would catch it
        assert np.abs(d1 - 2.0).max() > 0.05
