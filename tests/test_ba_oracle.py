"""CPU: the bundle-adjustment block oracle (oracle/ba_oracle.c) against a numpy transcription of
bundleAdjustmentRKf.m:793-899,1641-1829 and against finite differences of its own residual."""
import numpy as np

import oracle


def _rot(rng, scale=0.2):
    w = rng.normal(0, scale, 3)
    a = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / a
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def _cam(rng, f=None, cx=320.0, cy=240.0):
    return {"f": float(rng.uniform(500, 900) if f is None else f), "cx": cx, "cy": cy, "R": _rot(rng)}


def _pack(c):
    return np.concatenate([[c["f"], c["cx"], c["cy"]], c["R"].ravel(order="F")])


def _K(c):
    return np.array([[c["f"], 0, c["cx"]], [0, c["f"], c["cy"]], [0, 0, 1.0]])


def _residual(uo, us, co, cs):  # computeSingleResidual :1668-1680
    pH = _K(co) @ co["R"] @ cs["R"].T @ np.linalg.solve(_K(cs), np.array([us[0], us[1], 1.0]))
    if abs(pH[2]) < 1e-10:
        pH[2] = 1e-10
    return uo - pH[:2] / pH[2], pH


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def _jac(us, co, cs, pH, typ):  # computeJacobianWrtCamera :1712-1781
    x, y, z = pH
    Jc = -np.array([[1 / z, 0, -x / z ** 2], [0, 1 / z, -y / z ** 2]])
    uh = np.array([us[0], us[1], 1.0])
    xs = np.linalg.solve(_K(cs), uh)
    J = np.zeros((2, 4))
    for m in range(3):
        S = _skew(np.eye(3)[m])
        if typ == "obs":
            d = _K(co) @ co["R"] @ S @ cs["R"].T @ xs
        else:
            d = _K(co) @ co["R"] @ (-cs["R"].T @ S) @ xs
        J[:, m] = Jc @ d
    if typ == "obs":
        J[:, 3] = Jc @ (np.diag([1.0, 1.0, 0.0]) @ co["R"] @ cs["R"].T @ xs)
    else:
        f = cs["f"]
        dKi = np.array([[-1 / f ** 2, 0, cs["cx"] / f ** 2], [0, -1 / f ** 2, cs["cy"] / f ** 2], [0, 0, 0.0]])
        J[:, 3] = Jc @ (_K(co) @ co["R"] @ cs["R"].T @ dKi @ uh)
    return J


def _pair_numpy(Ui, Uj, ci, cj, li, lj, sigma, both):  # jacobianPair + the parfor body
    rows_r, rows_i, rows_j = [], [], []
    E = r2 = cnt = 0.0
    for ui, uj in zip(Ui, Uj):
        for (uo, us, co, cs, lo, ls, swap) in [(ui, uj, ci, cj, li, lj, False)] + ([(uj, ui, cj, ci, lj, li, True)] if both else []):
            _, pH = _residual(uo, us, co, cs)
            Jo, Js = _jac(us, co, cs, pH, "obs"), _jac(us, co, cs, pH, "src")
            r, _ = _residual(uo, us, lo, ls)
            nr = np.sqrt(r @ r)
            w = 1.0 if nr < sigma else sigma / nr
            sw = np.sqrt(w)
            rows_r.append(sw * r)
            rows_i.append(sw * (Js if swap else Jo))
            rows_j.append(sw * (Jo if swap else Js))
            E += 0.5 * sw ** 2 * (r @ r)
            r2 += sw ** 2 * (r @ r)
            cnt += 2
    r = np.concatenate(rows_r)
    Ji, Jj = np.concatenate(rows_i), np.concatenate(rows_j)
    return Ji.T @ Ji, Jj.T @ Jj, Ji.T @ Jj, Ji.T @ r, Jj.T @ r, E, r2, cnt


def _scene(rng, m):
    ci, cj = _cam(rng), _cam(rng)
    X = rng.normal(0, 1, (m, 3)) + np.array([0, 0, 4.0])  # rays in front of both cameras
    def proj(c):
        p = (_K(c) @ c["R"] @ X.T).T
        return p[:, :2] / p[:, 2:3]
    return ci, cj, proj(ci) + rng.normal(0, 1.5, (m, 2)), proj(cj) + rng.normal(0, 1.5, (m, 2))


def test_blocks_equal_the_numpy_transcription():
    rng = np.random.default_rng(1)
    for both in (True, False):
        for m in (1, 5, 64, 65, 300):
            ci, cj, Ui, Uj = _scene(rng, m)
            li, lj = dict(ci, f=ci["f"] + 3.0, R=_rot(rng, 0.01) @ ci["R"]), dict(cj, R=_rot(rng, 0.01) @ cj["R"])
            if m == 5:
                Ui[2] += 200.0  # an outlier beyond sigma: the Huber branch
            out = oracle.ba_pair_blocks(Ui, Uj, [0, m], np.stack([_pack(c) for c in (ci, cj, li, lj)])[None], 2.0, both)[0]
            Hii, Hjj, Hij, gi, gj, E, r2, cnt = _pair_numpy(Ui, Uj, ci, cj, li, lj, 2.0, both)
            scale = max(1.0, np.abs(Hii).max(), np.abs(Hjj).max())
            assert np.allclose(out[0:16].reshape(4, 4, order="F"), Hii, rtol=1e-9, atol=1e-9 * scale)
            assert np.allclose(out[16:32].reshape(4, 4, order="F"), Hjj, rtol=1e-9, atol=1e-9 * scale)
            assert np.allclose(out[32:48].reshape(4, 4, order="F"), Hij, rtol=1e-9, atol=1e-9 * scale)
            assert np.allclose(out[48:52], gi, rtol=1e-9, atol=1e-9 * scale) and np.allclose(out[52:56], gj, rtol=1e-9, atol=1e-9 * scale)
            assert np.isclose(out[56], E, rtol=1e-12) and np.isclose(out[57], r2, rtol=1e-12) and out[58] == cnt


def test_jacobians_are_the_derivatives_of_the_residual():
    """Pins the restated analytic Jacobians: with one match, no Huber down-weighting and equal base/incremented cameras,
    g = J' r and H = J' J; J is recovered column by column from finite differences of the residual under f <- f + df
    and the rotation perturbations the reference's formulas differentiate: R_obs <- R_obs (I + [dth]x) ("dR/dth =
    R [e_m]x", :1738-1741) for the observing camera and R_src <- (I + [dth]x) R_src ("d(R')/dth = -R' [e_m]x",
    :1758-1761) for the source camera."""
    rng = np.random.default_rng(2)
    ci, cj, Ui, Uj = _scene(rng, 1)

    def res(cam_i, cam_j):
        return _residual(Ui[0], Uj[0], cam_i, cam_j)[0]

    def bump(c, k, h, right):
        if k < 3:
            w = np.zeros(3)
            w[k] = h
            return dict(c, R=c["R"] @ (np.eye(3) + _skew(w)) if right else (np.eye(3) + _skew(w)) @ c["R"])
        return dict(c, f=c["f"] + h)

    h = 1e-6
    Ji = np.stack([(res(bump(ci, k, h, True), cj) - res(bump(ci, k, -h, True), cj)) / (2 * h) for k in range(4)], 1)
    Jj = np.stack([(res(ci, bump(cj, k, h, False)) - res(ci, bump(cj, k, -h, False))) / (2 * h) for k in range(4)], 1)
    r = res(ci, cj)
    out = oracle.ba_pair_blocks(Ui, Uj, [0, 1], np.stack([_pack(c) for c in (ci, cj, ci, cj)])[None], 1e9, False)[0]
    assert np.allclose(out[48:52], Ji.T @ r, rtol=1e-5, atol=1e-4) and np.allclose(out[52:56], Jj.T @ r, rtol=1e-5, atol=1e-4)
    assert np.allclose(out[0:16].reshape(4, 4, order="F"), Ji.T @ Ji, rtol=1e-4, atol=1e-2)
    assert np.allclose(out[32:48].reshape(4, 4, order="F"), Ji.T @ Jj, rtol=1e-4, atol=1e-2)


def test_pairs_are_independent_and_empty_pairs_are_zero():
    rng = np.random.default_rng(3)
    packs, Uis, Ujs, ptr = [], [], [], [0]
    for m in (7, 0, 130):
        ci, cj, Ui, Uj = _scene(rng, max(m, 1))
        packs.append(np.stack([_pack(c) for c in (ci, cj, ci, cj)]))
        Uis.append(Ui[:m])
        Ujs.append(Uj[:m])
        ptr.append(ptr[-1] + m)
    allo = oracle.ba_pair_blocks(np.concatenate(Uis), np.concatenate(Ujs), ptr, np.stack(packs), 2.0, True)
    assert not allo[1].any()
    for p in (0, 2):
        one = oracle.ba_pair_blocks(Uis[p], Ujs[p], [0, len(Uis[p])], packs[p][None], 2.0, True)[0]
        assert np.array_equal(one.view(np.uint64), allo[p].view(np.uint64))
