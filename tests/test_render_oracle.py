"""CPU tests pinning the render/blend oracle with analytic cases."""
import math

import numpy as np

import oracle


def cam(f, W, H, yaw=0.0, pitch=0.0):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return {"K": np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]), "R": (Ry @ Rx).T}


def geo_planar(f, W, H, u0, v0, Rref=np.eye(3)):
    return {"mode": "planar", "H": H, "W": W, "fPan": f, "o0": u0, "o1": v0, "Rref": Rref}


def test_tent_weights_match_linspace_definition():
    assert oracle.tent(1).tolist() == [0.0]  # MATLAB linspace(a,b,1) returns b: the second assignment wins
    for n in (2, 5, 8):
        w = oracle.tent(n)
        a = (n + 1) // 2
        ref = np.ones(n)
        mls = lambda lo, hi, k: np.array([hi]) if k == 1 else np.linspace(lo, hi, k)  # MATLAB linspace(.,.,1) = hi
        ref[:a] = mls(0, 1, a)
        ref[n // 2:] = mls(1, 0, n - n // 2)
        np.testing.assert_allclose(w, ref.astype(np.float32), atol=0)


def test_identity_planar_camera_reproduces_the_image():
    # canvas pixel (x,y) 0-based -> u = u0 + x/f ; image coord = f*u + cx.  With u0 = (1-cx)/f the canvas
    # pixel x lands exactly on image pixel x+1 (1-based): bilinear weights are 0/1.
    rng = np.random.default_rng(0)
    W, H, f = 40, 30, 50.0
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    c = cam(f, W, H)
    geo = geo_planar(f, W, H, (1 - W / 2) / f, (1 - H / 2) / f)
    S, M, Wa, Wf = oracle.warp_tile(img, c, geo, 0, 0, H, W, 2.0)
    # f32 rounding of the normalised ray can push the outermost ring a hair outside [1,w]x[1,h]
    assert M[1:-1, 1:-1].all() and M.mean() > 0.9
    np.testing.assert_allclose(S[M], (img.astype(np.float32) / 255.0)[M], atol=2e-4)
    np.testing.assert_allclose(Wf[M], np.outer(oracle.tent(H), oracle.tent(W))[M], atol=2e-4)
    assert np.all(Wa[M] > 0.5) and np.all(Wa <= 1.0)


def test_out_of_frame_and_back_facing_are_masked():
    W, H, f = 32, 24, 40.0
    img = np.full((H, W, 3), 200, np.uint8)
    geo = geo_planar(f, 3 * W, H, (1 - 3 * W / 2) / f, (1 - H / 2) / f)
    S, M, Wa, Wf = oracle.warp_tile(img, cam(f, W, H), geo, 0, 0, H, 3 * W, 2.0)
    assert M[:, :W - 1].sum() == 0 and M[:, 2 * W + 1:].sum() == 0 and M[:, W + 2:2 * W - 2].all()
    assert np.all(S[~M] == 0) and np.all(Wa[~M] == 0) and np.all(Wf[~M] == 0)
    S, M, Wa, Wf = oracle.warp_tile(img, cam(f, W, H, yaw=math.pi), geo, 0, 0, H, 3 * W, 2.0)
    assert not M.any()  # camera looks the other way: front test fails


def test_constant_image_warps_to_constant_on_sphere_and_gain_applies():
    W, H, f = 64, 48, 80.0
    img = np.full((H, W, 3), 128, np.uint8)
    geo = {"mode": "spherical", "H": 40, "W": 60, "fPan": f, "o0": -0.3, "o1": -0.2, "Rref": np.eye(3)}
    S, M, Wa, Wf = oracle.warp_tile(img, cam(f, W, H, yaw=0.05), geo, 0, 0, 40, 60, 2.0, gain=(1.0, 0.5, 2.0))
    assert M.sum() > 1000
    base = np.float32(128) / np.float32(255)
    np.testing.assert_allclose(S[M], np.tile([base, base * 0.5, base * 2.0], (M.sum(), 1)), rtol=3e-7)


def test_yaw_shifts_the_cylinder_by_f_times_angle():
    rng = np.random.default_rng(1)
    W, H, f = 200, 40, 300.0
    img = (rng.random((H, W, 1)) * 255).astype(np.uint8).repeat(3, 2)
    geo = {"mode": "cylindrical", "H": 30, "W": 400, "fPan": f, "o0": -0.6, "o1": -0.04, "Rref": np.eye(3)}
    yaw = 30 / f  # 30 canvas pixels
    S0, M0, _, _ = oracle.warp_tile(img, cam(f, W, H), geo, 0, 0, 30, 400)
    S1, M1, _, _ = oracle.warp_tile(img, cam(f, W, H, yaw=yaw), geo, 0, 0, 30, 400)
    both = M0[:, :-30] & M1[:, 30:]
    assert both.sum() > 2000
    np.testing.assert_allclose(S0[:, :-30][both], S1[:, 30:][both], atol=2e-3)


def test_gauss_and_resize_building_blocks():
    a = np.full((9, 7), 3.0, np.float32)
    np.testing.assert_allclose(oracle.gaussfilt(a, 1.0), 3.0, rtol=1e-6)    # normalised taps, replicate pad
    np.testing.assert_allclose(oracle.imresize(a, 4, 3), 3.0, rtol=1e-6)
    np.testing.assert_allclose(oracle.imresize(a, 18, 14), 3.0, rtol=1e-6)
    # exact 2:1 antialiased bilinear = taps (1,3,3,1)/8 on an interior pixel
    r = np.arange(16, dtype=np.float32)[None, :].repeat(4, 0)
    d = oracle.imresize(r, 4, 8)
    np.testing.assert_allclose(d[0, 1:-1], (r[0, 1:-4:2] + 3 * r[0, 2:-3:2] + 3 * r[0, 3:-2:2] + r[0, 4:-1:2]) / 8, rtol=1e-6)
    # a linear ramp is reproduced by bilinear upsampling away from the clamped border
    u = oracle.imresize(d, 4, 16)
    np.testing.assert_allclose(u[0, 3:-3], r[0, 3:-3], atol=1e-5)
    # impulse response of the 5-tap sigma=1 filter sums to one and is symmetric
    imp = np.zeros((11, 11), np.float32)
    imp[5, 5] = 1
    g = oracle.gaussfilt(imp, 1.0)
    assert abs(g.sum() - 1) < 1e-6 and np.allclose(g, g.T) and np.count_nonzero(g) == 25


def test_multiband_single_layer_telescopes_to_the_input():
    rng = np.random.default_rng(2)
    for (h, w, L) in ((32, 48, 3), (33, 47, 4), (5, 9, 5)):
        c = rng.random((1, h, w, 3), dtype=np.float32)
        wt = np.ones((1, h, w), np.float32)
        F = oracle.multiband_blend(c, wt, L, 1.0)
        np.testing.assert_allclose(F, c[0], atol=3e-6)  # G - up(D) + up(D) with w == 1 everywhere


def test_multiband_two_constant_layers_give_weighted_mean_and_levels_clamp():
    h, w = 24, 40
    c = np.stack([np.full((h, w, 3), 0.2, np.float32), np.full((h, w, 3), 0.8, np.float32)])
    wt = np.stack([np.full((h, w), 3.0, np.float32), np.full((h, w), 1.0, np.float32)])
    F = oracle.multiband_blend(c, wt, 3, 1.0)
    np.testing.assert_allclose(F, 0.75 * 0.2 + 0.25 * 0.8, atol=2e-6)
    # levels is clamped to floor(log2(min(h,w))) = 4 (:98-99): 50 levels == 4 levels
    rng = np.random.default_rng(3)
    c = rng.random((2, h, w, 3), dtype=np.float32)
    wt = rng.random((2, h, w), dtype=np.float32)
    assert np.array_equal(oracle.multiband_blend(c, wt, 50, 1.0), oracle.multiband_blend(c, wt, 4, 1.0))
    # zero total weight -> zeros; result clamped to [0,1]
    F = oracle.multiband_blend(c * 3, np.zeros_like(wt), 2, 1.0)
    assert np.all(F == 0)
    assert oracle.multiband_blend(c * 3, wt, 2, 1.0).max() <= 1.0


def test_linear_blend_formula():
    rng = np.random.default_rng(4)
    c = rng.random((3, 6, 7, 3), dtype=np.float32)
    wt = rng.random((3, 6, 7), dtype=np.float32)
    wt[:, 0, 0] = 0
    F = oracle.linear_blend(c, wt)
    ref = (c * wt[..., None]).sum(0) / np.maximum(wt.sum(0), np.finfo(np.float32).eps)[..., None]
    np.testing.assert_allclose(F, ref, rtol=1e-5, atol=1e-7)
    assert np.all(F[0, 0] == 0)


def test_render_blend_modes_on_two_overlapping_constant_images():
    W, H, f = 64, 48, 100.0
    imgs = [np.full((H, W, 3), 100, np.uint8), np.full((H, W, 3), 200, np.uint8)]
    cams = [cam(f, W, H, yaw=-0.15), cam(f, W, H, yaw=0.15)]
    geo = {"mode": "spherical", "H": 50, "W": 100, "fPan": f, "o0": -0.5, "o1": -0.25, "Rref": np.eye(3)}
    for blending in ("none", "linear", "multiband"):
        pano, cov = oracle.render(imgs, cams, geo, (32, 64), 2.0, blending, 3, 1.0)
        assert cov.sum() > 2000 and np.all(pano[cov == 0] == 0)
        vals = pano[cov == 1]
        assert vals.max() <= 205
        if blending == "multiband":
            # coarse-level weights are blurred across the coverage border and NOT renormalised per level
            # (multiBandBlending.m:131-147), so the reference darkens a rim around the covered region
            assert 95 <= np.median(vals) <= 205 and vals.min() >= 40
        else:
            assert vals.min() >= 95
        if blending == "none":  # 'last' policy: image 2 overwrites the overlap
            assert set(np.unique(vals)) <= {100, 200}
    white, cov = oracle.render(imgs, cams, geo, (32, 64), 2.0, "linear", 3, 1.0, canvas_white=True)
    assert np.all(white[cov == 0] == 255)
    # 'first' keeps image 1 in the overlap, 'maxangle' switches at the bisector
    p_first, _ = oracle.render(imgs, cams, geo, (50, 100), 2.0, "none", none_policy="first")
    p_last, _ = oracle.render(imgs, cams, geo, (50, 100), 2.0, "none", none_policy="last")
    p_max, _ = oracle.render(imgs, cams, geo, (50, 100), 2.0, "none", none_policy="maxangle")
    assert (p_first == 100).sum() > (p_last == 100).sum()
    assert (p_first == 100).sum() > (p_max == 100).sum() > (p_last == 100).sum()


def test_multiband_depends_on_tile_size_like_the_reference():
    """Pyramids are built per tile (renderPanorama.m:1038): a different tile size changes border pixels —
    the reason opts.tile must be an explicit input."""
    rng = np.random.default_rng(5)
    W, H, f = 64, 48, 100.0
    imgs = [rng.integers(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(2)]
    cams = [cam(f, W, H, yaw=-0.1), cam(f, W, H, yaw=0.1)]
    geo = {"mode": "spherical", "H": 48, "W": 96, "fPan": f, "o0": -0.45, "o1": -0.24, "Rref": np.eye(3)}
    a, _ = oracle.render(imgs, cams, geo, (48, 96), 2.0, "multiband", 3, 1.0)
    b, _ = oracle.render(imgs, cams, geo, (24, 48), 2.0, "multiband", 3, 1.0)
    assert (a != b).any()


def test_image_warp_identity_and_shift():
    rng = np.random.default_rng(6)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    out = oracle.image_warp_h(img, np.eye(3), 20, 30, 1.0, 1.0, 1.0, 1.0, fill=7)
    assert np.array_equal(out[:-1, :-1], img[:-1, :-1])      # last row/col: x2 <= W fails (:133) -> fill
    assert np.all(out[-1] == 7) and np.all(out[:, -1] == 7)
    T = np.array([[1, 0, 4.0], [0, 1, 2.0], [0, 0, 1]])      # forward shift by (4,2): out(x,y) = in(x-4,y-2)
    out = oracle.image_warp_h(img, T, 20, 30, 1.0, 1.0, 1.0, 1.0, fill=0)
    assert np.array_equal(out[2:-1, 4:-1], img[:-3, :-5])
    assert np.all(out[:2] == 0) and np.all(out[:, :4] == 0)


def test_gain_overlap_statistics_analytic_cases():
    """gainCompensationRKf.m:239-367 on cases with a known answer: two identical cameras looking at constant images
    count every sampled point whose tent weight is positive, sums are count x the constants; a third image that
    looks the other way never pairs; the canvas is sampled at 1-based 1:stride:W (:106-107)."""
    W, H, f = 64, 48, 80.0
    imgs = [np.full((H, W, 3), v, np.uint8) for v in (100, 150, 200)]
    for k in range(3):
        imgs[k][..., 1] = imgs[k][..., 0] // 2
    cams = [cam(f, W, H), cam(f, W, H), cam(f, W, H, yaw=math.pi)]
    # planar canvas whose 1-based point (xp, yp) is image pixel (xp, yp): u = f*(u0 + xp/f) + W/2 = xp  =>  u0 = -W/(2f)
    geo = geo_planar(f, W, H, -W / (2 * f), -H / (2 * f))
    for stride in (1, 3):
        N, sI, sJ = oracle.gain_overlap_stats(imgs, cams, geo, stride)
        xs, ys = np.arange(1, W + 1, stride), np.arange(1, H + 1, stride)
        wx, wy = oracle.tent(W), oracle.tent(H)
        expect = int(((wx[xs - 1] > 0)[None, :] & (wy[ys - 1] > 0)[:, None]).sum())  # border rows/cols have weight 0
        assert N[0, 1] == expect and N.sum() == expect  # only the pair (0, 1); image 2 faces away
        assert np.allclose(sI[0, 1], [100 * expect, 50 * expect, 100 * expect])
        assert np.allclose(sJ[0, 1], [150 * expect, 75 * expect, 150 * expect])
        assert np.all(np.tril(N) == 0)


def test_imresize_u8_analytic_cases():
    """imresize on uint8 (resizeImagesToLimits.m:57-61): a constant stays constant at any scale (weights are
    normalised), the scalar form yields ceil(s * size), halving a linear ramp averages neighbouring pairs, and
    upscaling by an integer factor keeps monotone ramps monotone (bicubic overshoot is clamped by uint8 saturation)."""
    const = np.full((37, 53, 3), 91, np.uint8)
    for s in (0.21, 0.5, 1.0, 1.7):
        out = oracle.imresize_u8(const, s)
        assert out.shape == (int(np.ceil(37 * s)), int(np.ceil(53 * s)), 3) and np.all(out == 91)
    ramp = np.tile((np.arange(64, dtype=np.uint8) * 4)[None, :], (8, 1))
    half = oracle.imresize_u8(ramp, (8, 32), "bilinear")
    assert np.array_equal(half[0, 1:-1], ((ramp[0, 2:-2:2].astype(int) + ramp[0, 3:-1:2]) // 2).astype(np.uint8))
    up = oracle.imresize_u8(ramp, 2.0, "bicubic")
    assert up.shape == (16, 128) and np.all(np.diff(up[3].astype(int)) >= 0)
    sat = np.zeros((4, 16), np.uint8)
    sat[:, 8:] = 255  # a step: the cubic overshoot must saturate, not wrap
    o = oracle.imresize_u8(sat, 3.0, "bicubic")
    assert o.min() == 0 and o.max() == 255


def test_gain_overlap_statistics_of_warped_canvases_analytic_cases():
    """gainCompensationH.m:45-52,78-149 on cases with a known answer: constant canvases with rectangular weight supports
    count the ds-strided samples of the supports' intersection (1:ds:end = 0-based 0, ds, ...), the sums are count x the
    constants per channel; a non-finite colour or a zero weight removes the sample for that image only; and a literal
    numpy transcription of the reference's vectorised accumulation agrees on random data."""
    Hc, Wc = 37, 53
    Iw = [np.zeros((Hc, Wc, 3), np.float32) for _ in range(3)]
    Ww = [np.zeros((Hc, Wc), np.float32) for _ in range(3)]
    vals = [(0.2, 0.4, 0.6), (0.5, 0.25, 0.125), (0.9, 0.8, 0.7)]
    boxes = [(0, 30, 0, 40), (5, 37, 10, 53), (20, 37, 0, 20)]  # r0, r1, c0, c1
    for k in range(3):
        Iw[k][...] = vals[k]
        r0, r1, c0, c1 = boxes[k]
        Ww[k][r0:r1, c0:c1] = 0.5
    for ds in (1, 4, 5):
        N, sI, sJ = oracle.gain_overlap_stats_warped(Iw, Ww, ds)
        ys, xs = np.arange(0, Hc, ds), np.arange(0, Wc, ds)

        def cnt(a, b):
            r0, r1 = max(boxes[a][0], boxes[b][0]), min(boxes[a][1], boxes[b][1])
            c0, c1 = max(boxes[a][2], boxes[b][2]), min(boxes[a][3], boxes[b][3])
            return int(((ys >= r0) & (ys < r1)).sum() * ((xs >= c0) & (xs < c1)).sum())

        for a, b in ((0, 1), (0, 2), (1, 2)):
            assert N[a, b] == cnt(a, b)
            assert np.allclose(sI[a, b], np.array(vals[a], np.float32).astype(np.float64) * cnt(a, b), rtol=1e-12)
            assert np.allclose(sJ[a, b], np.array(vals[b], np.float32).astype(np.float64) * cnt(a, b), rtol=1e-12)
        assert np.all(np.tril(N) == 0)
    # invalid samples: a NaN in one channel of image 1 and a zero weight of image 0 at sampled points inside the overlap
    Iw[1][8, 12, 2] = np.nan
    Ww[0][12, 16] = 0.0
    N2, _, _ = oracle.gain_overlap_stats_warped(Iw, Ww, 4)
    N0, _, _ = oracle.gain_overlap_stats_warped([np.nan_to_num(a) for a in Iw], [np.where(w == 0, w, w) for w in Ww], 4)
    assert N2[0, 1] == N0[0, 1] - 1  # (8, 12): image 1 invalid there; (12, 16) is counted in N0 only if Ww[0] > 0 - it is not
    # the reference's vectorised form, transcribed: valid = (W > 0) & all(isfinite(I), channel); overlap = vi & vj; sums in double
    rng = np.random.default_rng(5)
    n = 4
    Iw = [rng.random((29, 41, 3)).astype(np.float32) for _ in range(n)]
    Ww = [(rng.random((29, 41)) > 0.4).astype(np.float32) * rng.random((29, 41)).astype(np.float32) for _ in range(n)]
    Iw[2][::3, ::5, 1] = np.inf
    ds = 3
    N, sI, sJ = oracle.gain_overlap_stats_warped(Iw, Ww, ds)
    I4 = np.stack([a[::ds, ::ds] for a in Iw], 3).reshape(-1, 3, n)
    W3 = np.stack([w[::ds, ::ds] for w in Ww], 2).reshape(-1, n)
    valid = (W3 > 0) & np.isfinite(I4).all(1)
    for i in range(n - 1):
        for j in range(i + 1, n):
            ov = valid[:, i] & valid[:, j]
            assert N[i, j] == ov.sum()
            for ch in range(3):
                assert np.isclose(sI[i, j, ch], np.where(ov, I4[:, ch, i], 0).astype(np.float64).sum(), rtol=1e-12)
                assert np.isclose(sJ[i, j, ch], np.where(ov, I4[:, ch, j], 0).astype(np.float64).sum(), rtol=1e-12)
