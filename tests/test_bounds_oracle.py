"""The canvas geometry of renderPanorama.m:84-232,1459-1754 - four bounds functions, auto-reference, margins, pixel
padding, the megapixel cap, cropNonzeroBbox - restated in C (oracle/bounds_oracle.c) and compared with the product's
host mirror (<pkg>/renderPanorama.py) for every projection mode, plus closed-form known answers for the oracle itself.
No GPU involved: this is host logic on both sides."""
import math
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_render_oracle import cam


@pytest.fixture(scope="module")
def rp(aps):
    return import_module(aps.__name__ + ".renderPanorama")


def _rig(n=5, W=640, H=480, f=700.0, seed=0, step=0.4):
    rng = np.random.default_rng(seed)
    cams = []
    for i in range(n):
        c = cam(f * rng.uniform(0.9, 1.1), W, H, yaw=(i - (n - 1) / 2) * step + rng.uniform(-0.03, 0.03),
                pitch=rng.uniform(-0.15, 0.15))
        r = rng.uniform(-0.05, 0.05)
        Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1.0]])
        c["R"] = Rz.T @ c["R"]
        cams.append(c)
    sizes = [(H, W, 3)] * n
    return cams, sizes


def test_single_axis_aligned_camera_has_closed_form_bounds():
    W, H, f = 800, 600, 500.0
    c = cam(f, W, H)
    K = np.asarray(c["K"])
    cx, cy = K[0, 2], K[1, 2]
    th0, th1 = math.atan2(1 - cx, f), math.atan2(W - cx, f)
    t0, t1, p0, p1 = oracle.bounds("spherical", [c], [(H, W, 3)])
    assert abs(t0 - th0) < 1e-14 and abs(t1 - th1) < 1e-14
    # phi is extreme on the top/bottom border at the grid column closest to the centre line
    dx = min(abs(np.linspace(1, W, 48) - cx))
    assert abs(p0 - math.atan2(1 - cy, math.hypot(f, dx))) < 1e-14 and abs(p1 - math.atan2(H - cy, math.hypot(f, dx))) < 1e-14
    t0c, t1c, h0, h1 = oracle.bounds("cylindrical", [c], [(H, W, 3)])
    assert abs(t0c - th0) < 1e-14 and abs(t1c - th1) < 1e-14
    assert abs(h1 - (H - cy) / math.hypot(f, min(abs(np.linspace(1, W, 48) - cx)))) < 1e-12
    # planar, reference = the camera itself: u = (x - cx)/f, clipped at the 1st / 99th percentile of the samples
    u0, u1, v0, v1 = oracle.bounds("planar", [c], [(H, W, 3)], Rref=c["R"], robust_pct=(0, 100))
    assert abs(u0 - (1 - cx) / f) < 1e-14 and abs(u1 - (W - cx) / f) < 1e-14
    assert abs(v0 - (1 - cy) / f) < 1e-14 and abs(v1 - (H - cy) / f) < 1e-14
    # stereographic of the same: a = x / (1 + z) of the unit ray
    a0, a1, b0, b1 = oracle.bounds("stereographic", [c], [(H, W, 3)], Rref=c["R"], robust_pct=(0, 100))
    x, y = (W - cx) / f, 0.0
    # the extreme a is reached on the right border at the row closest to the centre line
    rows = np.linspace(1, H, 512)
    yy = (rows[np.argmin(abs(rows - cy))] - cy) / f
    nr = math.sqrt(x * x + yy * yy + 1)
    assert abs(a1 - (x / nr) / (1 + 1 / nr)) < 1e-12 and y == 0.0


@pytest.mark.parametrize("mode", ["cylindrical", "spherical", "planar", "stereographic"])
def test_python_mirror_bounds_equal_the_oracle(rp, mode):
    cams, sizes = _rig()
    Rref = cams[2]["R"]
    if mode == "cylindrical":
        got = rp.cylindricalBounds(cams, sizes)
    elif mode == "spherical":
        got = rp.sphericalBounds(cams, sizes)
    elif mode == "planar":
        got = rp.planarBounds(cams, sizes, Rref, (1, 99), 8.0)
    else:
        got = rp.stereographicBounds(cams, sizes, Rref, (1, 99), 8.0)
    want = oracle.bounds(mode, cams, sizes, Rref=Rref)
    np.testing.assert_allclose(got, want, rtol=0, atol=5e-13)  # LAPACK solve vs back substitution: a few ulp


@pytest.mark.parametrize("mode", ["cylindrical", "spherical", "equirectangular", "planar", "perspective", "stereographic"])
@pytest.mark.parametrize("variant", ["default", "scaled", "capped", "fixed_ref"])
def test_canvas_geometry_equals_the_oracle(rp, mode, variant):
    cams, sizes = _rig(n=6, seed=3)
    o = rp.default_opts({"anglePower": 2}, cams, 1)
    if variant == "scaled":
        o.update(resScale=0.6, margin=0.05, pixelPad=10)
    elif variant == "capped":
        o.update(maxMegapixel=0.5)       # forces the global pixel cap of the planar / stereographic branches
    elif variant == "fixed_ref":
        o.update(autoRef=False)
    got = rp.canvas_geometry(cams, sizes, mode, 1, o)
    want = oracle.canvas_geometry(cams, sizes, mode, 1, o)
    assert (got["W"], got["H"]) == (want["W"], want["H"])
    assert got["refIdx"] == want["refIdx"]
    assert abs(got["o0"] - want["o0"]) < 1e-12 and abs(got["o1"] - want["o1"]) < 1e-12
    if variant == "capped" and mode in ("planar", "perspective", "stereographic"):
        assert got["W"] * got["H"] <= 0.5e6 * 1.01 and want["resScale"] < 1.0
    if variant == "fixed_ref":
        assert got["refIdx"] == 1


def test_auto_reference_picks_the_smallest_canvas(rp):
    cams, sizes = _rig(n=7, seed=5, step=0.18)
    o = rp.default_opts({}, cams, 0)
    g = rp.canvas_geometry(cams, sizes, "planar", 0, o)
    w = oracle.canvas_geometry(cams, sizes, "planar", 0, o)
    assert g["refIdx"] == w["refIdx"] and g["refIdx"] in (2, 3, 4)  # a middle camera minimises the plane extent
    areas = []
    for k in range(7):
        ok = dict(o, autoRef=False)
        gk = oracle.canvas_geometry(cams, sizes, "planar", k, ok)
        areas.append(gk["W"] * gk["H"])
    assert int(np.argmin(areas)) == w["refIdx"]


def test_crop_nonzero_bbox_known_answers():
    """cropNonzeroBbox of the oracle against hand-derived rectangles (the device version is compared with the oracle in
    tests/test_crop_gpu.py)."""
    img = np.zeros((60, 90, 3), np.uint8)
    img[20:31, 40:56] = (10, 200, 30)
    img[5, 7] = (0, 0, 4)    # rgb2gray = round(0.456) = 0: NOT foreground
    img[50, 80] = (0, 0, 5)  # rgb2gray = round(0.570) = 1: foreground
    rect, did = oracle.crop_nonzero_bbox(img)
    assert did and rect == (15, 57, 35, 87)
    white = np.full((40, 50, 3), 255, np.uint8)
    white[10:12, 20:25] = (255, 255, 250)  # gray 254.43 -> 254 < 255: foreground
    white[30, 40] = (255, 255, 252)        # gray 254.66 -> 255: canvas
    rect, did = oracle.crop_nonzero_bbox(white, True)
    assert did and rect == (5, 18, 15, 31)
    assert oracle.crop_nonzero_bbox(np.zeros((8, 9, 3), np.uint8)) == ((1, 8, 1, 9), False)
    full = np.full((8, 9, 3), 9, np.uint8)
    assert oracle.crop_nonzero_bbox(full) == ((1, 8, 1, 9), True)


def test_solve_K_back_substitution_equals_general_solver(rp):
    """renderPanorama._solve_K: back substitution for the upper-triangular intrinsics (mldivide's triangular branch, what
    oracle/bounds_oracle.c::solve_K does) against LAPACK's general solver; a K with a lower entry takes the general path."""
    rng = np.random.default_rng(4)
    Ks = np.zeros((5, 3, 3))
    for k in range(5):
        f = rng.uniform(500, 9000)
        Ks[k] = [[f, rng.uniform(-2, 2), rng.uniform(100, 2000)], [0, f * rng.uniform(0.9, 1.1), rng.uniform(100, 1500)], [0, 0, 1]]
    B = rng.uniform(-1, 4000, (5, 3, 200))
    B[:, 2] = 1.0
    got = rp._solve_K(Ks, B)
    ref = np.linalg.solve(Ks, B)
    assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)
    Kg = Ks.copy()
    Kg[:, 2, 0] = 1e-4  # not triangular any more
    assert np.array_equal(rp._solve_K(Kg, B), np.linalg.solve(Kg, B))
