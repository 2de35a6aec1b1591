"""An independent check of the SIFT oracle's semantics (CPU).

getFeaturePoints.m:26-40 calls closed toolbox code, so the oracle (oracle/sift_oracle.c) restates the published algorithm
and nothing in the reference pins it.  This file rebuilds the same stage a second time from the published description
alone - float64, scipy.ndimage's Gaussian filter and interpolation, libm transcendentals, vectorised numpy instead of
the oracle's loops, no shared code - and requires the two to agree up to what f32-vs-f64 rounding can move:
  * the scale space: every oracle keypoint sits on a 26-neighbour extremum of the scipy DoG stack (same octave, layer,
    cell, coordinate convention), and every clearly-passing scipy extremum is an oracle keypoint;
  * the keypoint's scale follows sigma * 2^(layer/L) * 2^octave, the location the 2x base / 1-based convention;
  * the orientation is a peak of the float64 gradient histogram;
  * the 128-D descriptor equals the float64 textbook construction (cosine similarity).
"""
import math

import numpy as np
import pytest
from scipy import ndimage

import oracle

SIGMA, NL, CONTRAST, EDGE = 1.6, 4, 0.00133, 6.0  # the reference's call (getFeaturePoints.m:28-31 with input defaults)


def textured_image(h=240, w=320, seed=5):
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w))
    for s, a in ((1.0, 30.0), (2.5, 60.0), (6.0, 90.0), (14.0, 120.0)):
        n = ndimage.gaussian_filter(rng.normal(size=(h, w)), s)
        img += a * n / n.std() / 4
    yy, xx = np.mgrid[0:h, 0:w]
    img += 40 * ((xx // 40 + yy // 30) % 2)  # some corners and edges
    img = 128 + img
    g = np.clip(np.round(img), 0, 255).astype(np.uint8)
    return np.repeat(g[:, :, None], 3, axis=2)  # grey RGB: rgb2gray returns the value itself


def radius_of(sigma):
    return ((int(round(sigma * 8 + 1)) | 1) - 1) // 2  # cv::GaussianBlur's kernel size for float images


def blur(img, sigma):
    return ndimage.gaussian_filter(img, sigma, mode="mirror", radius=radius_of(sigma))


def scale_space(gray):
    """Gaussian and DoG stacks per octave, float64."""
    h, w = gray.shape
    yy, xx = np.mgrid[0:2 * h, 0:2 * w].astype(np.float64)
    up = ndimage.map_coordinates(gray, [(yy + 0.5) / 2 - 0.5, (xx + 0.5) / 2 - 0.5], order=1, mode="nearest")
    base = blur(up, math.sqrt(max(SIGMA ** 2 - 1.0, 0.01)))  # the input is assumed to carry sigma 0.5 (1.0 after 2x)
    n_oct = int(round(math.log2(min(2 * h, 2 * w)) - 2)) + 1
    k = 2.0 ** (1.0 / NL)
    G, D = [], []
    for o in range(n_oct):
        g = [base if o == 0 else G[o - 1][NL][::2, ::2]]
        for i in range(1, NL + 3):
            prev = SIGMA * k ** (i - 1)
            g.append(blur(g[-1], math.sqrt((prev * k) ** 2 - prev ** 2)))
        G.append(g)
        D.append([g[i + 1] - g[i] for i in range(NL + 2)])
    return G, D


def discrete_extrema(D):
    """{(octave, layer, r, c)} of the 26-neighbour extrema above OpenCV's pre-threshold, inside the 5-pixel border."""
    thr = math.floor(0.5 * CONTRAST / NL * 255.0)
    found = set()
    for o, dog in enumerate(D):
        st = np.stack(dog)
        if min(st.shape[1:]) <= 10:
            continue
        mx = ndimage.maximum_filter(st, size=3, mode="nearest")
        mn = ndimage.minimum_filter(st, size=3, mode="nearest")
        ext = (np.abs(st) > thr) & (((st > 0) & (st >= mx)) | ((st < 0) & (st <= mn)))
        ext[0] = ext[-1] = False
        ext[:, :5] = ext[:, -5:] = False
        ext[:, :, :5] = ext[:, :, -5:] = False
        for layer, r, c in zip(*np.nonzero(ext)):
            found.add((o, int(layer), int(r), int(c)))
    return found


def unpack(loc, aux):
    """Octave, layer and the octave-grid position of the oracle's keypoints."""
    o = (aux[:, 3].astype(int)) % 256
    layer = (aux[:, 3].astype(int)) // 256
    s = 2.0 ** o * 0.5
    return o, layer, (loc[:, 0] - 1) / s, (loc[:, 1] - 1) / s  # x, y in the octave's pixel grid


@pytest.fixture(scope="module")
def world():
    img = textured_image()
    desc, loc, aux = oracle.sift(img, SIGMA, NL, CONTRAST, EDGE)
    assert len(loc) > 300
    G, D = scale_space(img[:, :, 0].astype(np.float64))
    return img, desc, loc, aux, G, D


def test_octave_shapes_and_dog_agree_with_the_oracle_blur(world):
    img, _, _, _, G, D = world
    assert len(G) == oracle.sift_num_octaves(*img.shape[:2])
    assert G[0][0].shape == (2 * img.shape[0], 2 * img.shape[1]) and G[1][0].shape == img.shape[:2]
    # the oracle's f32 blur against scipy's on a pyramid level (its own chain order, same taps)
    a = oracle.sift_blur(G[1][0].astype(np.float32), 1.6)
    np.testing.assert_allclose(a, blur(G[1][0], 1.6), atol=5e-4)


def test_every_keypoint_sits_on_an_extremum_of_an_independent_scale_space(world):
    _, _, loc, aux, _, D = world
    ext = discrete_extrema(D)
    o, layer, x, y = unpack(loc, aux)
    hit = 0
    for i in range(len(loc)):
        r, c = int(round(y[i])), int(round(x[i]))
        hit += any((o[i], layer[i] + dl, r + dr, c + dc) in ext
                   for dl in (-1, 0, 1) for dr in (-1, 0, 1) for dc in (-1, 0, 1))
    # Newton steps may leave the discrete extremum's cell by one; f32 rounding moves a handful of near-ties
    assert hit >= 0.985 * len(loc), (hit, len(loc))


def test_every_clear_extremum_of_the_independent_scale_space_is_a_keypoint(world):
    _, _, loc, aux, _, D = world
    o, layer, x, y = unpack(loc, aux)
    have = {(int(o[i]), int(layer[i]), int(round(y[i])), int(round(x[i]))) for i in range(len(loc))}
    clear = missing = 0
    for (oo, ll, r, c) in discrete_extrema(D):
        d, p, q = D[oo][ll], D[oo][ll - 1], D[oo][ll + 1]
        v = d[r, c]
        g = np.array([(d[r, c + 1] - d[r, c - 1]) / 2, (d[r + 1, c] - d[r - 1, c]) / 2, (q[r, c] - p[r, c]) / 2])
        dxx, dyy, dss = d[r, c + 1] + d[r, c - 1] - 2 * v, d[r + 1, c] + d[r - 1, c] - 2 * v, q[r, c] + p[r, c] - 2 * v
        dxy = (d[r + 1, c + 1] - d[r + 1, c - 1] - d[r - 1, c + 1] + d[r - 1, c - 1]) / 4
        dxs = (q[r, c + 1] - q[r, c - 1] - p[r, c + 1] + p[r, c - 1]) / 4
        dys = (q[r + 1, c] - q[r - 1, c] - p[r + 1, c] + p[r - 1, c]) / 4
        Hm = np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]])
        if abs(np.linalg.det(Hm)) < 1e-12:
            continue
        off = -np.linalg.solve(Hm, g)  # the sub-pixel / sub-layer position of the fitted quadratic (Lowe 2004 §4)
        tr, det = dxx + dyy, dxx * dyy - dxy * dxy
        # "clear": no Newton step pending (offsets well under 0.5), interpolated contrast 3x the threshold, principal
        # curvature ratio well under the limit ((e+1)^2/e = 8.17 for e = 6; 5.5 here)
        if np.abs(off).max() > 0.4 or abs(v + 0.5 * g @ off) / 255 * NL < 3 * CONTRAST:
            continue
        if not (det > 0 and tr * tr < 5.5 * det):
            continue
        clear += 1
        missing += not any((oo, ll, r + dr, c + dc) in have for dr in (-1, 0, 1) for dc in (-1, 0, 1))
    assert clear > 150
    assert missing <= 0.01 * clear, (missing, clear)


def test_scale_and_location_conventions(world):
    img, _, loc, aux, _, _ = world
    o, layer, x, y = unpack(loc, aux)
    # kpt.size = sigma * 2^((layer + xi)/L) * 2^octave * 2, halved for the 2x base: the layer brackets the size
    size_oct = aux[:, 0] / 2.0 ** o
    lo, hi = SIGMA * 2 ** ((layer - 0.5) / NL), SIGMA * 2 ** ((layer + 0.5) / NL)
    assert np.all(size_oct >= lo * 0.999) and np.all(size_oct <= hi * 1.001)
    assert layer.min() >= 1 and layer.max() <= NL
    assert loc[:, 0].min() >= 1 and loc[:, 0].max() <= img.shape[1] and loc[:, 1].max() <= img.shape[0]


def gradient_field(g):
    gx = np.zeros_like(g)
    gy = np.zeros_like(g)
    gx[:, 1:-1] = g[:, 2:] - g[:, :-2]
    gy[1:-1, :] = g[2:, :] - g[:-2, :]  # image convention: y grows downwards
    return gx, gy


def orientation_peaks(g, r, c, scl):
    """Peaks (degrees, image convention: from +x towards +y) of the 36-bin smoothed gradient histogram, Lowe 2004 §5."""
    rad = int(round(4.5 * scl))
    gx, gy = gradient_field(g)
    rr, cc = np.mgrid[r - rad:r + rad + 1, c - rad:c + rad + 1]
    ok = (rr > 0) & (rr < g.shape[0] - 1) & (cc > 0) & (cc < g.shape[1] - 1)
    rr, cc = rr[ok], cc[ok]
    wgt = np.exp(-((rr - r) ** 2 + (cc - c) ** 2) / (2 * (1.5 * scl) ** 2)) * np.hypot(gx[rr, cc], gy[rr, cc])
    # OpenCV bins the y-up angle and reports 360 - peak: that is the y-down (image) angle, binned with the mirrored rounding
    ang_up = np.degrees(np.arctan2(-gy[rr, cc], gx[rr, cc])) % 360
    hist = np.bincount(np.round(ang_up / 10).astype(int) % 36, weights=wgt, minlength=36)
    sm = (np.roll(hist, 2) + np.roll(hist, -2)) / 16 + (np.roll(hist, 1) + np.roll(hist, -1)) * 4 / 16 + hist * 6 / 16
    peaks = []
    for j in range(36):
        l, rgt = sm[(j - 1) % 36], sm[(j + 1) % 36]
        if sm[j] > l and sm[j] > rgt and sm[j] >= 0.8 * sm.max():
            b = (j + 0.5 * (l - rgt) / (l - 2 * sm[j] + rgt)) % 36
            peaks.append((360 - 10 * b) % 360)
    return peaks


def test_orientations_are_peaks_of_a_float64_gradient_histogram(world):
    _, _, loc, aux, G, _ = world
    o, layer, x, y = unpack(loc, aux)
    rng = np.random.default_rng(0)
    pick = rng.choice(len(loc), 150, replace=False)
    good = 0
    for i in pick:
        scl = aux[i, 0] / 2.0 ** o[i]  # the keypoint's scale in its octave (size * 0.5 / 2^o, on the 2x base: no 0.5)
        peaks = orientation_peaks(G[o[i]][layer[i]], int(round(y[i])), int(round(x[i])), scl)
        d = [min(abs(p - aux[i, 1]), 360 - abs(p - aux[i, 1])) for p in peaks]
        good += bool(d) and min(d) < 3.0
    assert good >= 0.95 * len(pick), good


def textbook_descriptor(g, x, y, angle_img, scl):
    """4x4x8 descriptor, float64: the window is rotated into the keypoint's frame, samples weighted by a Gaussian of
    half the window width, votes spread trilinearly, normalise - clip at 0.2 - normalise (Lowe 2004 §6)."""
    d, n = 4, 8
    width = 3.0 * scl
    rad = int(round(width * math.sqrt(2) * (d + 1) * 0.5))
    gx, gy = gradient_field(g)
    c0, r0 = int(round(x)), int(round(y))  # OpenCV samples around the rounded position
    th = math.radians(angle_img)  # image convention
    ii, jj = np.mgrid[-rad:rad + 1, -rad:rad + 1]  # row / column offsets
    # coordinates in the keypoint's frame (u along the keypoint direction, v a quarter turn towards +y), in bin units
    u = (jj * math.cos(th) + ii * math.sin(th)) / width
    v = (-jj * math.sin(th) + ii * math.cos(th)) / width
    # grid columns follow u, grid rows v (calcSIFTDescriptor's c_rot / r_rot, written there with the y-up angle 360 - angle)
    rb, cb = v + d / 2 - 0.5, u + d / 2 - 0.5
    rr, cc = r0 + ii, c0 + jj
    ok = (rb > -1) & (rb < d) & (cb > -1) & (cb < d) & (rr > 0) & (rr < g.shape[0] - 1) & (cc > 0) & (cc < g.shape[1] - 1)
    rb, cb, rr, cc, u, v = rb[ok], cb[ok], rr[ok], cc[ok], u[ok], v[ok]
    mag = np.hypot(gx[rr, cc], gy[rr, cc]) * np.exp(-(u * u + v * v) / (0.5 * d * d))
    ori_up = np.degrees(np.arctan2(-gy[rr, cc], gx[rr, cc]))
    ob = ((ori_up - (360 - angle_img)) % 360) * n / 360
    hist = np.zeros((d + 2, d + 2, n))
    r_i, c_i, o_i = np.floor(rb).astype(int), np.floor(cb).astype(int), np.floor(ob).astype(int)
    fr, fc, fo = rb - r_i, cb - c_i, ob - o_i
    for dr in (0, 1):
        for dc in (0, 1):
            for do in (0, 1):
                wv = mag * (fr if dr else 1 - fr) * (fc if dc else 1 - fc) * (fo if do else 1 - fo)
                np.add.at(hist, (r_i + 1 + dr, c_i + 1 + dc, (o_i + do) % n), wv)
    vec = hist[1:d + 1, 1:d + 1].reshape(-1)
    vec = np.minimum(vec, 0.2 * np.linalg.norm(vec))
    return vec / max(np.linalg.norm(vec), 1e-12)


def test_descriptors_equal_a_float64_textbook_construction(world):
    _, desc, loc, aux, G, _ = world
    o, layer, x, y = unpack(loc, aux)
    rng = np.random.default_rng(1)
    pick = rng.choice(len(loc), 120, replace=False)
    cos = []
    for i in pick:
        scl = aux[i, 0] / 2.0 ** o[i]
        ref = textbook_descriptor(G[o[i]][layer[i]], x[i], y[i], float(aux[i, 1]), scl)
        got = desc[i].astype(np.float64)
        cos.append(float(ref @ got / np.linalg.norm(got)))
    cos = np.array(cos)
    # the oracle quantises to 0..255 integers before the final normalisation (OpenCV's uint8 descriptor): ~1e-3 off
    assert np.median(cos) > 0.9999 and cos.min() > 0.999, (np.median(cos), cos.min())  # measured 0.99998 / 0.99997
