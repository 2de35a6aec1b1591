"""The toolbox semantics the oracle fixes by fiat, cross-checked against INDEPENDENT
implementations that exist in this image: scipy.ndimage (Gaussian filter, linear interpolation), PIL and
torch.nn.functional.interpolate (antialiased bilinear / bicubic resize), numpy.linalg.svd (the DLT null vector),
scikit-learn (brute-force kNN).  The reference cannot pin these (it ships no fixtures and its toolbox code is closed);
third parties narrow what the oracle could have got wrong.  CPU only."""
import numpy as np
import pytest

import oracle


# ---- imgaussfilt ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sigma", [0.8, 1.0, 1.5, 2.0])
def test_gaussfilt_equals_scipy_gaussian_filter(sigma):
    """imgaussfilt(I, sigma, 'Padding','replicate'): filter size 2*ceil(2*sigma)+1, normalised samples of the Gaussian.
    scipy: radius = int(truncate*sigma + 0.5) with truncate = ceil(2 sigma)/sigma, mode 'nearest' = replicate."""
    from scipy import ndimage

    rng = np.random.default_rng(int(sigma * 10))
    img = rng.random((67, 91), dtype=np.float32)
    got = oracle.gaussfilt(img, sigma)
    want = ndimage.gaussian_filter(img.astype(np.float64), sigma, mode="nearest", truncate=np.ceil(2 * sigma) / sigma)
    assert np.abs(got - want).max() < 2e-6
    # three channels are filtered independently
    rgb = rng.random((40, 33, 3), dtype=np.float32)
    got3 = oracle.gaussfilt(rgb, sigma)
    for c in range(3):
        w3 = ndimage.gaussian_filter(rgb[..., c].astype(np.float64), sigma, mode="nearest", truncate=np.ceil(2 * sigma) / sigma)
        assert np.abs(got3[..., c] - w3).max() < 2e-6


# ---- imresize (f32, bilinear, antialiased on shrink) ---------------------------------------------------------------
@pytest.mark.parametrize("shape,out", [((64, 96), (32, 48)), ((65, 97), (32, 48)), ((33, 47), (16, 23)),
                                       ((32, 48), (64, 96)), ((16, 23), (33, 47)), ((50, 70), (20, 70))])
def test_imresize_bilinear_equals_torch_antialias_away_from_the_border(shape, out):
    """imresize(I, [oh ow], 'bilinear') with its default antialiasing = triangle kernel widened by 1/scale, half-pixel
    centres - the same filter torch (and PIL) use for antialias=True.  The treatments of the border differ by design:
    MATLAB replicates the edge pixel (indices clamped, weights kept), torch/PIL truncate the kernel and renormalise;
    so the comparison excludes the output pixels whose kernel support leaves the image."""
    import torch
    import torch.nn.functional as F

    rng = np.random.default_rng(shape[0] + out[0])
    img = rng.random(shape, dtype=np.float32)
    got = oracle.imresize(img, out[0], out[1])
    t = torch.from_numpy(img)[None, None]
    want = F.interpolate(t, size=out, mode="bilinear", antialias=True, align_corners=False)[0, 0].numpy()
    ky = int(np.ceil(max(1.0, shape[0] / out[0]))) + 1
    kx = int(np.ceil(max(1.0, shape[1] / out[1]))) + 1
    my = int(np.ceil(ky * out[0] / shape[0])) + 1
    mx = int(np.ceil(kx * out[1] / shape[1])) + 1
    inner = (slice(my, out[0] - my), slice(mx, out[1] - mx))
    assert got[inner].size > 50
    assert np.abs(got[inner] - want[inner]).max() < 5e-6
    if out[0] >= shape[0] and out[1] >= shape[1]:
        # enlarging: the kernel is the plain triangle and clamping == torch's edge handling: equal everywhere
        assert np.abs(got - want).max() < 5e-6


def test_imresize_u8_equals_pil():
    """imresize(I, [oh ow]) on uint8 (the preprocessing in front of SIFT, resizeImagesToLimits.m:103): bicubic (Keys
    a = -0.5) and bilinear, antialiased on shrink, half-pixel centres, uint8 rounding after each pass - PIL's BICUBIC /
    BILINEAR reducing filters do the same.  PIL always resizes horizontally first; MATLAB takes the dimension with the
    smaller scale first (ties: rows), so PIL is given the transposed image whenever MATLAB starts with the rows.  PIL
    works with 22-bit fixed-point coefficients: identical on >= 96 % of the pixels, never more than 2 grey levels apart.
    (The scalar form imresize(I, s) keeps s as the kernel scale while the size is ceil(s*size): PIL has no such mode.)"""
    from PIL import Image

    rng = np.random.default_rng(3)
    base = rng.random((40, 56, 3))
    img = (np.kron(base, np.ones((6, 6, 1))) * 255).astype(np.uint8)  # 240 x 336, blocky: edges exercise the lobes
    h, w = img.shape[:2]
    for method, pil in (("bicubic", Image.Resampling.BICUBIC), ("bilinear", Image.Resampling.BILINEAR)):
        for (oh, ow) in ((120, 168), (89, 124), (192, 269), (100, 300), (200, 100)):
            got = oracle.imresize_u8(img, (oh, ow), method)
            if oh / h <= ow / w:  # MATLAB: rows first
                want = np.asarray(Image.fromarray(np.ascontiguousarray(img.transpose(1, 0, 2))).resize((oh, ow), pil)).transpose(1, 0, 2)
            else:
                want = np.asarray(Image.fromarray(img).resize((ow, oh), pil))
            d = np.abs(got.astype(int) - want.astype(int))
            assert d.max() <= 2, (method, oh, ow, d.max())
            assert (d == 0).mean() >= 0.96 and (d <= 1).mean() >= 0.995, (method, oh, ow, (d == 0).mean())


# ---- interp2 / imageWarp bilinear ------------------------------------------------------------------------------------
def test_bilinear_warp_equals_scipy_map_coordinates():
    """imageWarp 'bilinear' (imageWarp.m:125-168) = interp2 'linear' at the inverse-mapped points; scipy's
    map_coordinates(order=1) is an independent linear interpolator.  Valid region only (the reference fills the rest)."""
    from scipy import ndimage

    rng = np.random.default_rng(4)
    img = rng.random((60, 80)).astype(np.float32)
    a, s = 0.2, 1.1
    Hm = np.array([[s * np.cos(a), -s * np.sin(a), 7.0], [s * np.sin(a), s * np.cos(a), -3.0], [1e-4, -2e-4, 1.0]])
    oh, ow = 70, 90
    got = oracle.image_warp_h(img, Hm, oh, ow, 0.5, 0.5, 1.0, 1.0, fill=-1.0)
    # the reference's output grid is x0 + (0:ow-1)*sx with x0 = XWorldLimits(1) (imageWarp.m:43-50) -> source through H^-1
    X, Y = np.meshgrid(0.5 + np.arange(ow, dtype=np.float64), 0.5 + np.arange(oh, dtype=np.float64))
    q = np.linalg.inv(Hm) @ np.stack([X.ravel(), Y.ravel(), np.ones(X.size)])
    sx, sy = q[0] / q[2], q[1] / q[2]
    want = ndimage.map_coordinates(img.astype(np.float64), [sy - 1, sx - 1], order=1, mode="constant", cval=np.nan).reshape(oh, ow)
    valid = (np.floor(sx) >= 1) & (np.floor(sx) + 1 <= 80) & (np.floor(sy) >= 1) & (np.floor(sy) + 1 <= 60)
    valid = valid.reshape(oh, ow)
    assert valid.sum() > 2000
    assert np.abs(got[valid] - want[valid]).max() < 2e-6
    assert np.all(got[~valid] == -1.0)


def _keys(x):
    a = np.abs(x)
    return np.where(a <= 1, 1.5 * a ** 3 - 2.5 * a ** 2 + 1, np.where(a <= 2, -0.5 * a ** 3 + 2.5 * a ** 2 - 4 * a + 2, 0.0))


def test_nearest_and_bicubic_warp_against_a_vectorised_numpy_form_and_closed_forms():
    """imageWarp 'nearest' (imageWarp.m:109-123) and 'bicubic' (:170-264, Keys a = -0.5 over the 4 x 4 taps around
    floor(src)).  No third-party interpolator in this image uses that kernel for arbitrary maps (torch: a = -0.75, scipy:
    B-splines), so the independent form is a vectorised numpy evaluation (einsum over the separable weights); closed
    forms: an integer translation reproduces the image, a constant stays constant, nearest rounds half away from zero."""
    rng = np.random.default_rng(14)
    img = rng.random((50, 70)).astype(np.float32)
    a, s = -0.15, 0.93
    Hm = np.array([[s * np.cos(a), -s * np.sin(a), 5.25], [s * np.sin(a), s * np.cos(a), 2.5], [2e-4, 1e-4, 1.0]])
    oh, ow = 64, 84
    X, Y = np.meshgrid(1.0 + np.arange(ow, dtype=np.float64), 1.0 + np.arange(oh, dtype=np.float64))
    q = np.linalg.inv(Hm) @ np.stack([X.ravel(), Y.ravel(), np.ones(X.size)])
    sx, sy = (q[0] / q[2]).reshape(oh, ow), (q[1] / q[2]).reshape(oh, ow)
    # bicubic
    got = oracle.image_warp_h(img, Hm, oh, ow, 1.0, 1.0, 1.0, 1.0, fill=-1.0, method="bicubic")
    fx, fy = np.floor(sx), np.floor(sy)
    valid = (fx >= 2) & (fx <= 70 - 2) & (fy >= 2) & (fy <= 50 - 2)
    assert valid.sum() > 1500 and np.all(got[~valid] == -1.0)
    xi, yi = fx[valid].astype(int), fy[valid].astype(int)
    offs = np.arange(-1, 3)
    wx = _keys(offs[None, :] - (sx[valid] - fx[valid])[:, None])
    wy = _keys(offs[None, :] - (sy[valid] - fy[valid])[:, None])
    taps = img.astype(np.float64)[(yi[:, None, None] + offs[None, :, None] - 1), (xi[:, None, None] + offs[None, None, :] - 1)]
    want = np.clip(np.einsum("nyx,ny,nx->n", taps, wy, wx), 0.0, 1.0)
    assert np.abs(got[valid] - want).max() < 1e-6
    assert np.abs(wx.sum(1) - 1).max() < 1e-12  # the Keys weights are a partition of unity
    # integer translation: the interior is reproduced exactly, by both methods
    T = np.array([[1, 0, 3.0], [0, 1, -2.0], [0, 0, 1.0]])
    for method in ("bicubic", "nearest"):
        out = oracle.image_warp_h(img, T, 50, 70, 1.0, 1.0, 1.0, 1.0, fill=-1.0, method=method)
        assert np.array_equal(out[4:40, 8:60], img[6:42, 5:57]), method
    u8 = rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)
    const = np.full((30, 30), 0.625, np.float32)
    cw = oracle.image_warp_h(const, Hm, 20, 20, 8.0, 8.0, 1.0, 1.0, fill=-1.0, method="bicubic")
    assert (cw != -1.0).sum() > 200 and np.abs(cw[cw != -1.0] - 0.625).max() < 1e-6
    # nearest: source x = X - 0.5 -> round half away from zero picks the pixel to the right of the tie
    half = np.array([[1, 0, 0.5], [0, 1, 0.0], [0, 0, 1.0]])
    out = oracle.image_warp_h(u8, half, 40, 40, 1.0, 1.0, 1.0, 1.0, fill=7, method="nearest")
    assert np.array_equal(out[:, 1:], u8[:, 1:]) and np.array_equal(out[:, 0], u8[:, 0])  # src 0.5 rounds to 1
    out2 = oracle.image_warp_h(u8, np.array([[1, 0, 1.5], [0, 1, 0.0], [0, 0, 1.0]]), 40, 40, 1.0, 1.0, 1.0, 1.0, fill=7, method="nearest")
    assert np.all(out2[:, 0] == 7) and np.array_equal(out2[:, 1:], u8[:, :-1])  # src -0.5 rounds to -1 (outside), 0.5 -> 1


# ---- DLT: Jacobi on the Gram matrix vs LAPACK svd ------------------------------------------------------------------------
def _dlt_svd(p1, p2):
    """estimateHomography (estimateTransformationRANSAC.m:188-225) with numpy's LAPACK svd: Hartley normalisation,
    2 rows per correspondence, null vector = last right singular vector, denormalise."""
    def norm(p):
        c = p.mean(0)
        d = np.sqrt(((p - c) ** 2).sum(1)).mean()
        s = 1.0 / d  # normalizePoints (:579-610) scales to MEAN distance 1 (not sqrt(2))
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        return (p - c) * s, T

    a, T1 = norm(p1)
    b, T2 = norm(p2)
    rows = []
    for (x, y), (u, v) in zip(a, b):
        rows.append([-x, -y, -1, 0, 0, 0, u * x, u * y, u])
        rows.append([0, 0, 0, -x, -y, -1, v * x, v * y, v])
    _, _, Vt = np.linalg.svd(np.asarray(rows))
    Hn = Vt[-1].reshape(3, 3)
    Hd = np.linalg.inv(T2) @ Hn @ T1
    return Hd / Hd[2, 2]


@pytest.mark.parametrize("m,noise", [(4, 0.0), (4, 0.3), (12, 0.5), (500, 0.7)])
def test_dlt_by_jacobi_gram_equals_lapack_svd(m, noise):
    """The oracle takes V(:,end) of svd(A) as the smallest eigenvector of A'A by cyclic Jacobi (squares the condition
    number).  Against LAPACK's svd on the same normalised system: the homographies agree to ~1e-9 relative, and scoring
    the matches with either model flips no inlier at the reference's threshold 5.5 (and none at a 100x tighter one)."""
    rng = np.random.default_rng(m)
    Ht = np.array([[1.02, 0.03, 40.0], [-0.02, 0.98, -25.0], [2e-5, -1e-5, 1.0]])
    p1 = rng.uniform([1, 1], [1600, 1200], size=(m, 2))
    q = np.c_[p1, np.ones(m)] @ Ht.T
    p2 = q[:, :2] / q[:, 2:] + noise * rng.standard_normal((m, 2))
    Ho, ok = oracle.fit_homography(p1, p2, np.arange(m))
    assert ok
    Ho = Ho / Ho[2, 2]
    Hs = _dlt_svd(p1, p2)
    assert np.abs(Ho - Hs).max() / np.abs(Hs).max() < 1e-8, np.abs(Ho - Hs).max()
    # the refit on the inliers sums in the wave order (64 lane-strided partials + butterfly, round 4) instead of sequentially:
    # a different rounding of the same sums - within 1e-11 of the sequential fit, and the same 1e-8 of LAPACK
    Hw, ok = oracle.fit_homography(p1, p2, np.arange(m), refit=True)
    assert ok
    Hw = Hw / Hw[2, 2]
    assert np.abs(Hw - Ho).max() / np.abs(Ho).max() < 1e-11 and np.abs(Hw - Hs).max() / np.abs(Hs).max() < 1e-8
    if m > 64:
        assert not np.array_equal(Hw, Ho)  # (with more than 64 points the two orders do round differently)
    # scoring ALL matches of a larger set with either model: identical inlier masks
    M = 2000
    a = rng.uniform([1, 1], [1600, 1200], size=(M, 2))
    qb = np.c_[a, np.ones(M)] @ Ht.T
    b = qb[:, :2] / qb[:, 2:] + rng.standard_normal((M, 2)) * np.where(rng.random((M, 1)) < 0.5, 1.5, 30.0)
    for thr in (5.5, 0.055):
        n1, e1, m1 = oracle.ransac_score(np.stack([Ho, Hs]), a, b, thr)
        assert n1[0] == n1[1] and np.array_equal(m1[0], m1[1])
        assert (np.isnan(e1[0]) and np.isnan(e1[1])) or abs(e1[0] - e1[1]) < 1e-6


# ---- kNN --------------------------------------------------------------------------------------------------------------
def test_knn_and_2nn_equal_sklearn_brute_force():
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(6)
    x = rng.gamma(0.6, 1.0, size=(700, 128)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = rng.gamma(0.6, 1.0, size=(300, 128)).astype(np.float32)
    y /= np.linalg.norm(y, axis=1, keepdims=True)
    idx, dist = oracle.knn(x, y, 4)
    nn = NearestNeighbors(n_neighbors=4, algorithm="brute", metric="sqeuclidean").fit(x.astype(np.float64))
    d, i = nn.kneighbors(y.astype(np.float64))
    assert np.array_equal(idx.astype(np.int64) - 1, i)           # random data: no ties
    assert np.abs(dist - d).max() < 1e-5                          # f32 accumulation against f64
    i2, d1, d2 = oracle.match_2nn_ssd(y, x)
    assert np.array_equal(i2.astype(np.int64) - 1, i[:, 0])
    assert np.abs(d1 - d[:, 0]).max() < 1e-5 and np.abs(d2 - d[:, 1]).max() < 1e-5
