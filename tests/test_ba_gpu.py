"""GPU parity: aps_ba_pair_blocks (csrc/ba.hip) against oracle/ba_oracle.c, bit for bit (f64, fixed evaluation order),
and the host mirror of accumulateNormalEqnsBlock around it."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from test_ba_oracle import _cam, _pack, _rot, _scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ba(gpu):
    return import_module(gpu.__name__ + ".bundleAdjustment")


def _batch(rng, sizes):
    packs, Uis, Ujs, ptr = [], [], [], [0]
    for m in sizes:
        ci, cj, Ui, Uj = _scene(rng, max(m, 1))
        li = dict(ci, f=ci["f"] + rng.normal(0, 2), R=_rot(rng, 0.01) @ ci["R"])
        lj = dict(cj, f=cj["f"] + rng.normal(0, 2), R=_rot(rng, 0.01) @ cj["R"])
        if m > 4:
            Ui[rng.integers(0, m, max(1, m // 10))] += rng.normal(0, 60, (max(1, m // 10), 2))  # Huber outliers
        packs.append(np.stack([_pack(c) for c in (ci, cj, li, lj)]))
        Uis.append(Ui[:m])
        Ujs.append(Uj[:m])
        ptr.append(ptr[-1] + m)
    return np.concatenate(Uis), np.concatenate(Ujs), ptr, np.stack(packs)


@pytest.mark.parametrize("both", [True, False])
def test_blocks_bit_exact(ba, both):
    rng = np.random.default_rng(10)
    Ui, Uj, ptr, cams = _batch(rng, [1, 2, 63, 64, 65, 0, 500, 4097, 7])
    got = ba.ba_pair_blocks(Ui, Uj, ptr, cams, 2.0, both)
    want = oracle.ba_pair_blocks(Ui, Uj, ptr, cams, 2.0, both)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    assert not got[5].any() and got[6, 58] == (4 if both else 2) * 500


def test_degenerate_depth_and_sigma_edge(ba):
    """|z| < 1e-10 is clamped (computeSingleResidual :1671-1673), and a residual norm exactly at sigma takes the L1 branch."""
    rng = np.random.default_rng(11)
    ci, cj = _cam(rng), _cam(rng)
    ci["R"] = np.eye(3)
    cj["R"] = np.array([[0, 0, 1.0], [0, 1, 0], [-1, 0, 0]])  # the source ray along x of the observer: z = 0 for the principal ray
    Uj = np.array([[cj["cx"], cj["cy"]], [cj["cx"] + 1, cj["cy"] - 2]])
    Ui = np.array([[10.0, 20.0], [300.0, 200.0]])
    cams = np.stack([_pack(c) for c in (ci, cj, ci, cj)])[None]
    for both in (True, False):
        got = ba.ba_pair_blocks(Ui, Uj, [0, 2], cams, 3.0, both)
        want = oracle.ba_pair_blocks(Ui, Uj, [0, 2], cams, 3.0, both)
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)) and np.isfinite(got).all()


def test_accumulate_mirror_assembles_like_the_reference(ba):
    """Three cameras (one the one-parameter seed), two edges: H/g from the device blocks equal the assembly of the
    oracle's blocks, H is symmetric, the seed contributes one row/column (its dthx column, as in the reference), and a
    zero increment gives residuals at the base cameras."""
    rng = np.random.default_rng(12)
    n = 3
    cams = [dict(_cam(rng), K=None) for _ in range(n)]
    for c in cams:
        c["K"] = np.array([[c["f"], 0, c["cx"]], [0, c["f"], c["cy"]], [0, 0, 1.0]])
    kps = [rng.uniform(0, 600, (200, 2)) for _ in range(n)]
    matches = [[None] * n for _ in range(n)]
    matches[0][1] = np.stack([rng.permutation(200)[:80] + 1, rng.permutation(200)[:80] + 1], 1)
    matches[1][2] = np.stack([rng.permutation(200)[:50] + 1, rng.permutation(200)[:50] + 1], 1)
    Phi, pmap = ba.buildDeltaVector(cams, [0, 1, 2], 1)
    assert len(Phi) == 9 and [e["startIdx"] for e in pmap] == [0, 4, 5]
    Phi[:] = rng.normal(0, 1e-3, 9)
    H, g, E, rmse = ba.accumulateNormalEqnsBlock(Phi, pmap, cams, [0, 1, 2], 1, matches, kps, None, 2.0)
    Ho, go, Eo, ro = ba.accumulateNormalEqnsBlock(Phi, pmap, cams, [0, 1, 2], 1, matches, kps, None, 2.0,
                                                  blocks=lambda *a: oracle.ba_pair_blocks(*a))
    assert np.array_equal(H, Ho) and np.array_equal(g, go) and E == Eo and rmse == ro
    assert np.array_equal(H, H.T) and H.shape == (9, 9) and E > 0 and rmse > 0
    assert not H[0:4, 5:9].any()  # cameras 0 and 2 share no edge
    lin = ba.applyIncrements(cams, Phi, pmap)
    assert abs(lin[1]["f"] - cams[1]["f"] - Phi[4]) < 1e-12 and np.allclose(lin[0]["R"] @ lin[0]["R"].T, np.eye(3), atol=1e-12)


def test_argument_errors(ba):
    z = np.zeros((4, 2))
    with pytest.raises(ValueError):
        ba.ba_pair_blocks(z, z, [0, 4], np.zeros((2, 4, 12)), 2.0)
    with pytest.raises(RuntimeError):
        ba.ba_pair_blocks(z, z, [0, 4], np.ones((1, 4, 12)), -1.0)
    with pytest.raises(RuntimeError):
        ba.ba_pair_blocks(z, z, [0, 9], np.ones((1, 4, 12)), 2.0)


def test_bench_scale_batch(ba, gpu):
    """384 pairs x ~3.7 k matches (the 64-view scene's verified edges): one launch, identical to the oracle on a
    sample of the pairs."""
    rng = np.random.default_rng(13)
    Ui, Uj, ptr, cams = _batch(rng, [int(v) for v in rng.integers(3000, 4500, 384)])
    gpu._capi.profile_enable(True)
    gpu._capi.profile_reset()
    got = ba.ba_pair_blocks(Ui, Uj, ptr, cams, 2.0, True)
    prof = gpu._capi.profile_all()
    gpu._capi.profile_enable(False)
    sel = [0, 17, 383]
    for p in sel:
        one = oracle.ba_pair_blocks(Ui[ptr[p]:ptr[p + 1]], Uj[ptr[p]:ptr[p + 1]], [0, ptr[p + 1] - ptr[p]], cams[p][None], 2.0, True)[0]
        assert np.array_equal(got[p].view(np.uint64), one.view(np.uint64))
    assert prof["ba_pair_blocks"][0] < 50.0
