"""GPU parity: the HIP matcher (through the C ABI) against the CPU oracle, bit for bit."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from util import bits, planted_pair, sift_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(gpu):
    return import_module(gpu.__name__ + ".featureMatching")


@pytest.mark.parametrize("n1,n2", [(1, 2), (2, 1), (3, 3), (33, 65), (257, 129), (128, 64), (1000, 4096),
                                   (4096, 1000)])
def test_2nn_bit_exact(fm, n1, n2):
    rng = np.random.default_rng(100 + n1 + n2)
    a, b, _, _ = planted_pair(rng, n1, n2, min(n1, n2) // 2)
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert np.array_equal(idx, oi)
    assert np.array_equal(bits(d1), bits(o1))
    assert np.array_equal(bits(d2), bits(o2))


def test_2nn_ties_first_index_and_duplicates(fm):
    rng = np.random.default_rng(7)
    b = sift_like(rng, 300)
    b[250] = b[17]
    b[40] = b[17]
    a = np.concatenate([b[[17, 40, 250]], sift_like(rng, 61)])
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert idx[:3].tolist() == [18, 18, 18]
    assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))


def test_2nn_column_major_input_is_identical(fm):
    rng = np.random.default_rng(8)
    a, b, _, _ = planted_pair(rng, 200, 333, 100)
    r = fm.nearest2SSDExhaustive(a, b)
    c = fm.nearest2SSDExhaustive(np.asfortranarray(a), np.asfortranarray(b))
    for x, y in zip(r[1:], c[1:]):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))


@pytest.mark.parametrize("unit", [True, False])
@pytest.mark.parametrize("unique", [True, False])
def test_match_features_equals_oracle(fm, unit, unique):
    rng = np.random.default_rng(11)
    a, b, ia, ib = planted_pair(rng, 1500, 1700, 600, noise=0.03, unit=unit)
    m, met = fm.matchFeaturesScratch(a, b, Method="Exhaustive", MatchThreshold=1.5, MaxRatio=0.6,
                                     Unique=unique)
    om, omet = oracle.match_features(a, b, 0.6, 1.5, unique, 2)
    assert len(om) > 300
    assert np.array_equal(m, om)
    assert np.array_equal(bits(met), bits(omet))


def test_unique_resolves_collisions_like_the_greedy_loop(fm):
    """Many A rows competing for few B rows: the atomicMin formulation must equal the sequential greedy."""
    rng = np.random.default_rng(12)
    b = sift_like(rng, 40)
    src = rng.integers(0, 8, size=500)
    a = b[src] + 0.01 * rng.standard_normal((500, 128)).astype(np.float32)
    a = np.maximum(a, 0).astype(np.float32)
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=10.0, MaxRatio=0.9, Unique=True)
    om, omet = oracle.match_features(a, b, 0.9, 10.0, True, 2)
    assert 1 <= len(om) <= 8
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))


def test_pairwise_cells_equal_per_pair_oracle(fm):
    rng = np.random.default_rng(13)
    base = sift_like(rng, 900)
    descs = []
    for i in range(5):
        keep = rng.permutation(900)[: 500 + 37 * i]
        d = base[keep] + 0.02 * rng.standard_normal((len(keep), 128)).astype(np.float32)
        d = np.maximum(d, 0)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        descs.append(d.astype(np.float32))
    inp = {"Matchingthreshold": 1.5, "Ratiothreshold": 0.6}
    cells = fm.featureMatchingPairwise(inp, descs, 5)
    total = 0
    for i in range(5):
        for j in range(5):
            if i < j:
                om, _ = oracle.match_features(descs[i], descs[j], 0.6, 1.5, True, 2)
                assert np.array_equal(cells[i][j], om.astype(np.float64)), (i, j)
                assert cells[i][j].dtype == np.float64
                total += len(om)
            else:
                assert cells[i][j] is None
    assert total > 500


def test_pairwise_with_empty_image(fm):
    rng = np.random.default_rng(14)
    descs = [sift_like(rng, 100), np.zeros((0, 128), np.float32), sift_like(rng, 80)]
    cells = fm.featureMatchingPairwise({"Matchingthreshold": 10.0, "Ratiothreshold": 0.9}, descs, 3)
    assert cells[0][1].shape == (0, 2) and cells[1][2].shape == (0, 2)
    om, _ = oracle.match_features(descs[0], descs[2], 0.9, 10.0, True, 2)
    assert np.array_equal(cells[0][2], om.astype(np.float64))


def test_torch_device_tensors_are_accepted(fm):
    import torch

    rng = np.random.default_rng(15)
    a, b, _, _ = planted_pair(rng, 700, 900, 300)
    m, met = fm.matchFeaturesScratch(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(),
                                     MatchThreshold=1.5, MaxRatio=0.6)
    om, omet = oracle.match_features(a, b, 0.6, 1.5, True, 2)
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))


def test_argument_errors(fm, gpu):
    a = np.zeros((4, 64), np.float32)
    with pytest.raises((gpu.ApsError, ValueError)):
        fm.matchFeaturesScratch(a, a)
    with pytest.raises(ValueError):
        fm.matchFeaturesScratch(np.zeros((0, 128), np.float32), np.zeros((3, 128), np.float32))
    with pytest.raises(ValueError):
        fm.matchFeaturesScratch(np.zeros((3, 128), np.float32), np.zeros((3, 128), np.float32), MaxRatio=1.5)


def test_split_precision_path_equals_f32_path_and_oracle(fm, monkeypatch):
    """Default mode = f16 screening product + exact rescoring + exact fallback rows; APS_MATCH_MODE=f32 = the
    all-f32 MFMA kernel.  Both must be bit-identical to the oracle."""
    rng = np.random.default_rng(21)
    a, b, _, _ = planted_pair(rng, 3000, 5000, 1200, noise=0.02)
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    for mode in ("split", "f32"):
        monkeypatch.setenv("APS_MATCH_MODE", mode)
        _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
        assert np.array_equal(idx, oi), mode
        assert np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2)), mode


def test_split_path_fallback_rows_on_near_ties(fm, gpu, monkeypatch):
    """Clusters of near-duplicate B rows make the 2nd..4th neighbours closer than the split-precision error bound:
    those rows cannot be certified and must go through the exact fallback — still bit-identical."""
    rng = np.random.default_rng(22)
    centres = sift_like(rng, 60)
    b = np.repeat(centres, 8, axis=0) + 2e-5 * rng.standard_normal((480, 128)).astype(np.float32)
    b = np.maximum(b, 0).astype(np.float32)
    b[100] = b[101]  # exact duplicate pair too
    a = np.concatenate([centres[:40] + 1e-4 * rng.standard_normal((40, 128)).astype(np.float32), sift_like(rng, 300)])
    a = np.maximum(a, 0).astype(np.float32)
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    gpu._capi.profile_enable(True)
    gpu._capi.profile_reset()
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    prof = gpu._capi.profile_all()
    gpu._capi.profile_enable(False)
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))
    assert "match2nn_fallback" in prof and "match_cand_f16" in prof  # the fallback really ran


def test_split_path_unnormalised_descriptors(fm, monkeypatch):
    """0..255-valued descriptors are normalised on the device first; large-norm raw inputs (normalize=0 via 2-NN API)
    scale the error bound with sqrt(a2 * max b2)."""
    rng = np.random.default_rng(23)
    a, b, _, _ = planted_pair(rng, 700, 900, 300, unit=False)
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)  # raw values up to 255: distances ~1e5
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=1.5, MaxRatio=0.6)
    om, omet = oracle.match_features(a, b, 0.6, 1.5, True, 2)
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))


@pytest.mark.parametrize("scale", [1e-7, 3e-5, 1.0 / 512, 37.0, 2000.0, 7e4, 3e9])
def test_split_path_any_magnitude(fm, monkeypatch, scale):
    """The screening product is f16: magnitudes that under- or overflow it (entries below 2^-14, above 65504, norms
    beyond the f16 slot) must only widen the per-row bound - every row then takes the exact fallback - never change
    a bit of the result."""
    rng = np.random.default_rng(31)
    a, b, _, _ = planted_pair(rng, 600, 1100, 250, noise=0.02)
    a = (a * np.float32(scale)).astype(np.float32)
    b = (b * np.float32(scale)).astype(np.float32)
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))


@pytest.mark.parametrize("n2", [1, 3, 4, 5, 31, 32, 33, 127, 128, 129, 255, 256, 257, 383, 384, 385, 511, 512, 513, 640])
def test_split_path_tile_boundaries(fm, monkeypatch, n2):
    """Column counts around the 32-column block and 128-column tile sizes (one, two, three ... LDS buffers in flight, ragged
    last tile) against row counts around the 64-row wave and 512-row workgroup sizes."""
    rng = np.random.default_rng(4000 + n2)
    b = sift_like(rng, n2)
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    for n1 in (1, 63, 64, 65, 511, 512, 513):
        a = sift_like(rng, n1)
        if n2 >= 2:
            a[0] = b[n2 - 1]  # the last column is somebody's nearest neighbour
        _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
        oi, o1, o2 = oracle.match_2nn_ssd(a, b)
        assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2)), (n1, n2)


def test_split_path_mixed_norms_signed_and_zero_rows(fm, monkeypatch):
    """Rows whose norms differ by six orders of magnitude inside one set (the three-piece b2/2 loses its low pieces
    for the small ones), signed entries, all-zero rows on both sides, and a B set smaller than the candidate list."""
    rng = np.random.default_rng(32)
    b = rng.standard_normal((700, 128)).astype(np.float32)
    b *= (10.0 ** rng.uniform(-3, 3, (700, 1))).astype(np.float32)
    b[5] = 0
    b[77] = 0
    a = np.concatenate([b[rng.permutation(700)[:200]] * np.float32(1.001), rng.standard_normal((150, 128)).astype(np.float32)])
    a[3] = 0
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    for bb in (b, b[:3], b[:4], b[:5]):
        _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, bb)
        oi, o1, o2 = oracle.match_2nn_ssd(a, bb)
        assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))


def test_split_path_certifies_most_rows_of_unit_sift(fm, monkeypatch, capfd):
    """On unit-norm SIFT-like sets the f16 bound must leave only a small share of the rows to the fallback (design
    point: well under 1 %); a regression of the bound would show up as time, not as a wrong bit, so the row count the
    library reports under APS_TRACE is checked here."""
    import re

    rng = np.random.default_rng(33)
    a, b, _, _ = planted_pair(rng, 4096, 6000, 1500, noise=0.02)
    monkeypatch.setenv("APS_MATCH_MODE", "split")
    monkeypatch.setenv("APS_TRACE", "1")
    capfd.readouterr()
    _, idx, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    err = capfd.readouterr().err
    oi, o1, o2 = oracle.match_2nn_ssd(a, b)
    assert np.array_equal(idx, oi) and np.array_equal(bits(d1), bits(o1)) and np.array_equal(bits(d2), bits(o2))
    m = re.search(r"fallback list (\d+) rows", err)
    rows = int(m.group(1)) if m else 0  # no line at all = every row certified
    assert rows <= 0.02 * len(a), (rows, err)


# ---- the 'Approximate' float back ends of matchFeaturesScratch.m:142-160 ---------------------------------------
def _approx_sets(seed=0, n1=600, n2=900, planted=250):
    rng = np.random.default_rng(seed)
    B = rng.gamma(0.6, 1.0, (n2, 128)).astype(np.float32) * 40  # unnormalised, like raw SIFT bins
    A = rng.gamma(0.6, 1.0, (n1, 128)).astype(np.float32) * 40
    A[:planted] = B[rng.permutation(n2)[:planted]] + rng.normal(0, 1.0, (planted, 128)).astype(np.float32)
    return np.maximum(A, 0), B


def test_kdtree_backend_equals_exhaustive(gpu):
    """knnsearch on a kd-tree is exact: same pairs as 'Exhaustive'; the metric differs only by the
    sqrt-then-square round trip of :152-153 (<= 2 ulp)."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    A, B = _approx_sets()
    me, de = fm.matchFeaturesScratch(A, B, Method="Exhaustive", MatchThreshold=1.5, MaxRatio=0.6)
    mk, dk = fm.matchFeaturesScratch(A, B, Method="Approximate", ApproxFloatNNMethod="kdtree", MatchThreshold=1.5, MaxRatio=0.6)
    # same pairs; order and metric may move by a few ulp: the back end normalises rows on the host (numpy
    # summation order) and takes sqrt-then-square of the distances (:152-153)
    assert len(me) > 200 and len(me) == len(mk)
    oe, ok = np.lexsort((me[:, 1], me[:, 0])), np.lexsort((mk[:, 1], mk[:, 0]))
    assert np.array_equal(me[oe], mk[ok])
    assert np.allclose(de[oe], dk[ok], rtol=2e-6, atol=1e-6)  # d = a2 + b2 - 2ab: a few ulp of 2 in absolute terms
    assert np.all(np.diff(dk) >= 0)


def test_subset_backend_is_exact_on_the_subset(gpu):
    fm = import_module(gpu.__name__ + ".featureMatching")
    A, B = _approx_sets(1)
    cand = np.random.default_rng(5).permutation(len(B))[:400] + 1
    idx1, idx2, e1, e2 = fm.nearest2SubsetPdist2(*fm._normalize_like_reference(A, B), 400, candB=cand)
    An, Bn = fm._normalize_like_reference(A, B)
    oi, od1, od2 = oracle.match_2nn_ssd(An, Bn[cand - 1])
    assert np.array_equal(idx2, cand[oi.astype(np.int64) - 1]) and np.array_equal(idx1, np.arange(1, len(A) + 1))
    assert np.array_equal(e1, np.sqrt(od1)) and np.array_equal(e2, np.sqrt(od2))
    m, d = fm.matchFeaturesScratch(A, B, Method="Approximate", ApproxFloatNNMethod="subsetpdist2", MatchThreshold=1.5,
                                   MaxRatio=0.6, seed=3)
    assert m.shape[1] == 2 and len(np.unique(m[:, 1])) == len(m) and np.all(np.diff(d) >= 0)
    with pytest.raises(ValueError):
        fm.nearest2SubsetPdist2(An, Bn, 400, candB=np.ones(400, np.int64))


def test_pca_backend_equals_the_oracle_bit_for_bit(gpu):
    """Row a6 (matchFeaturesScratch.m:442-573) on the device against oracle/pca_oracle.c: the mean, the principal axes (sign
    convention: largest-magnitude entry of every axis positive), and idx2 / d1 / d2 of the cosine 2-NN are identical bits,
    for a ragged size, a one-row B set, UsePCA off, and column-major (MATLAB) storage.  PCA-48 + cosine is an approximation
    OF the exhaustive search: the planted near-duplicates must still come out of the filtered matcher."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    A, B = _approx_sets(2)
    me, _ = fm.matchFeaturesScratch(A, B, Method="Exhaustive", MatchThreshold=1.5, MaxRatio=0.6)
    mp, dp = fm.matchFeaturesScratch(A, B, Method="Approximate", ApproxFloatNNMethod="pca2nn", MatchThreshold=1.5, MaxRatio=0.6)
    se, sp = {tuple(r) for r in me.tolist()}, {tuple(r) for r in mp.tolist()}
    assert len(se & sp) >= 0.9 * len(se)
    An, Bn = fm._normalize_like_reference(A, B)
    omu, oco, _ = oracle.pca_basis(Bn, 48)
    _, idx2, d1, d2, (mu, co) = fm.nearest2ApproxFloatFast(An, Bn, return_basis=True)
    assert np.array_equal(mu, omu) and np.array_equal(co, oco)
    assert np.all(co[np.abs(co).argmax(0), np.arange(48)] > 0)          # the sign convention
    oi, od1, od2 = oracle.pca2nn(An, Bn, 48, True)
    assert np.array_equal(idx2, oi) and np.array_equal(d1, od1) and np.array_equal(d2, od2)
    # the same through MATLAB's column-major storage
    _, i_f, d1_f, d2_f = fm.nearest2ApproxFloatFast(np.asfortranarray(An), np.asfortranarray(Bn))
    assert np.array_equal(i_f, oi) and np.array_equal(d1_f, od1) and np.array_equal(d2_f, od2)
    # other shapes: ragged tile tails (n2 not a multiple of 32, n1 not of 128), 20 components, a duplicated B row (the
    # first-index rule of max, :558, and its twin as the second), no projection, a single B row (d2 = 2 - 2 * -inf = inf)
    rng = np.random.default_rng(5)
    Bs = np.vstack([Bn[:333], Bn[7:8]])
    As = np.vstack([An[:131], Bn[7:8]])
    for k, use in ((48, True), (20, True), (48, False)):
        oi, od1, od2 = oracle.pca2nn(As, Bs, k, use)
        _, i2, e1, e2 = fm.nearest2ApproxFloatFast(As, Bs, {"ApproxNumComponents": k, "UsePCA": use})
        assert np.array_equal(i2, oi) and np.array_equal(e1, od1) and np.array_equal(e2, od2), (k, use)
        assert i2[-1] == 8 and e1[-1] == e2[-1]                        # the twin rows 8 and 334: the first wins, the other is second
    oi, od1, od2 = oracle.pca2nn(As, Bs[:1], 48, False)
    _, i2, e1, e2 = fm.nearest2ApproxFloatFast(As, Bs[:1], {"UsePCA": False})
    assert np.array_equal(i2, oi) and np.array_equal(e1, od1) and np.all(np.isinf(e2)) and np.all(np.isinf(od2))
    # fewer rows in B than components + 1: pca() returns min(n2 - 1, 48) columns, the rest of the basis stays zero
    for n2 in (2, 10, 48, 49):
        oi, od1, od2 = oracle.pca2nn(As, Bs[:n2], 48, True)
        omu, oco, _ = oracle.pca_basis(Bs[:n2], 48)
        _, i2, e1, e2, (mu2, co2) = fm.nearest2ApproxFloatFast(As, Bs[:n2], return_basis=True)
        assert np.array_equal(mu2, omu) and np.array_equal(co2, oco) and not co2[:, min(n2 - 1, 48):].any(), n2
        assert np.array_equal(i2, oi) and np.array_equal(e1, od1) and np.array_equal(e2, od2), n2
    with pytest.raises(ValueError):
        fm.nearest2ApproxFloatFast(As[:0], Bs)


def test_pca_backend_at_bench_scale_equals_the_oracle(gpu):
    """20 k x 20 k rows (what one pair of 4K views holds): every index and distance equal to the oracle's."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    rng = np.random.default_rng(9)
    base = sift_like(rng, 26000)
    A = base[:20000]
    B = np.maximum(base[3000:23000] + 0.03 * rng.standard_normal((20000, 128)).astype(np.float32), 0)
    B = (B / np.linalg.norm(B, axis=1, keepdims=True)).astype(np.float32)
    oi, od1, od2 = oracle.pca2nn(A, B, 48, True)
    _, i2, e1, e2 = fm.nearest2ApproxFloatFast(A, B)
    assert np.array_equal(i2, oi) and np.array_equal(e1, od1) and np.array_equal(e2, od2)


def test_pair_shards_compose_to_the_all_pairs_result(fm):
    """The multi-GPU matcher: the pair list is partitioned over ranks and each rank calls aps_match_pairs on its
    part (parallel.py step 3).  Any partition must reproduce the all-pairs CSR exactly, with resident inputs too."""
    import torch

    rng = np.random.default_rng(15)
    base = sift_like(rng, 700)
    descs = []
    for i in range(5):
        keep = rng.permutation(700)[: 380 + 29 * i]
        d = np.maximum(base[keep] + 0.02 * rng.standard_normal((len(keep), 128)).astype(np.float32), 0)
        descs.append((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32))
    pp, ia, ib, met = fm.match_pairwise_csr(descs, 0.6, 1.5, True)
    order = fm.pair_order(5)
    assert len(order) == 10 and pp[-1] > 500
    dd = [torch.from_numpy(d).cuda() for d in descs]
    torch.cuda.synchronize()
    for world in (2, 3):
        for r in range(world):
            mine = [p for p in range(len(order)) if p % world == r]
            for inputs in (descs, dd):
                qp, qa, qb, qm = fm.match_pairs_csr(inputs, [order[p] for p in mine], 0.6, 1.5, True)
                for k, p in enumerate(mine):
                    s0, s1 = int(pp[p]), int(pp[p + 1])
                    t0, t1 = int(qp[k]), int(qp[k + 1])
                    assert np.array_equal(ia[s0:s1], qa[t0:t1]) and np.array_equal(ib[s0:s1], qb[t0:t1])
                    assert np.array_equal(met[s0:s1].view(np.uint32), qm[t0:t1].view(np.uint32))


def _boundary_sets(target, n=40000, seed=0):
    """A_i = (t_i, u_i, 0, ...) against B = {0, e2}: d1 = t^2 + u^2, d2 = d1 + 1 - 2u.  (t, u) are scattered so that
    d1 / d2 (ratio mode) or d1 (threshold mode, target = (None, thr)) lands within a few 1e-6 of the boundary: dense
    enough that some rows fall between the f64 boundary and its f32-rounded neighbour."""
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 128), np.float32)
    if target[0] is not None:
        r = target[0] * (1 + 4e-6 * rng.uniform(-1, 1, n))          # wanted d1/d2
        u = rng.uniform(0.0, 0.02, n)
        d1 = r * (1 - 2 * u) / (1 - r)                                # d1 = r (d1 + 1 - 2u)
        a[:, 0] = np.sqrt(np.maximum(d1 - u * u, 0)).astype(np.float32)
        a[:, 1] = u.astype(np.float32)
    else:
        d1 = target[1] * (1 + 4e-6 * rng.uniform(-1, 1, n))
        u = rng.uniform(0.0, 0.02, n)
        a[:, 0] = np.sqrt(d1 - u * u).astype(np.float32)
        a[:, 1] = u.astype(np.float32)
    b = np.zeros((2, 128), np.float32)
    b[1, 1] = 1.0
    return a, b


def test_ratio_test_uses_double_r2_on_the_boundary(fm):
    """matchFeaturesScratch.m:170-173 evaluates MaxRatio^2 in double: rows whose d1/d2 falls between 0.36 and
    (double)0.6f^2 must be dropped, exactly as the oracle does (a float MaxRatio kept them)."""
    a, b = _boundary_sets((0.36, None))
    _, _, d1, d2 = fm.nearest2SSDExhaustive(a, b)
    dd1, dd2 = d1.astype(np.float64), d2.astype(np.float64)
    keep64 = dd1 <= (0.6 * 0.6) * dd2
    keep32 = dd1 <= (float(np.float32(0.6)) ** 2) * dd2
    assert keep64.any() and (~keep64).any() and (keep64 != keep32).any(), "the fixture must straddle the boundary"
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=3.5, MaxRatio=0.6, Unique=False)
    om, omet = oracle.match_features(a, b, 0.6, 3.5, False, 0)
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))
    assert np.array_equal(m[:, 0], np.flatnonzero(keep64) + 1)


def test_match_threshold_is_compared_in_double(fm):
    """MatchThreshold = 0.1 is not representable in f32: d1 values between 0.1 and (double)0.1f are dropped."""
    a, b = _boundary_sets((None, 0.1))
    b[1, 1] = 1.9  # d2 far away: only the threshold decides (max|B| <= 2, so the reference's rule does not normalise)
    _, _, d1, _ = fm.nearest2SSDExhaustive(a, b)
    dd1 = d1.astype(np.float64)
    assert (dd1 <= 0.1).any() and (dd1 > 0.1).any()
    assert ((dd1 <= 0.1) != (dd1 <= float(np.float32(0.1)))).any(), "the fixture must straddle the boundary"
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=0.1, MaxRatio=1.0, Unique=False)
    om, omet = oracle.match_features(a, b, 1.0, 0.1, False, 0)
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))
    assert np.array_equal(m[:, 0], np.flatnonzero(dd1 <= 0.1) + 1)


def test_worker_threads_inherit_the_selected_device(gpu, monkeypatch):
    """aps_set_device also sets the process-wide default: a thread that never selected a device must not fall back
    to APS_DEVICE / device 0.  On a one-GPU box the fallback is made to fail by an invalid APS_DEVICE."""
    import threading

    capi = gpu._capi
    capi.check(capi.lib.aps_set_device(0))
    monkeypatch.setenv("APS_DEVICE", "63")
    out = {}

    def work():
        out["rc"] = capi.lib.aps_synchronize()
        out["err"] = capi.lib.aps_last_error()

    th = threading.Thread(target=work)
    th.start()
    th.join()
    assert out["rc"] == 0, out


def test_thread_device_binding_does_not_touch_the_process_default(gpu):
    """aps_set_thread_device binds the calling thread only (pool workers must name their device instead of
    inheriting whichever one another thread selected last); aps_get_device reports the binding; an out-of-range device
    is refused and leaves the binding alone; re-binding after the auxiliary streams of a matching call exist works
    (bind_device drops them with the old device's other resources)."""
    import threading

    capi = gpu._capi
    capi.check(capi.lib.aps_set_device(0))
    out = {}

    def work():
        out["bad"] = capi.lib.aps_set_thread_device(63)
        out["ok"] = capi.lib.aps_set_thread_device(0)
        out["dev"] = capi.lib.aps_get_device()
        rng = np.random.default_rng(3)
        a, b = sift_like(rng, 300), sift_like(rng, 280)
        fmod = import_module(gpu.__name__ + ".featureMatching")
        out["m1"] = fmod.match_pairs_csr([a, b], [(0, 1)], 0.9, 1.5)[0][-1]  # creates this thread's auxiliary streams
        out["again"] = capi.lib.aps_set_thread_device(0)
        out["m2"] = fmod.match_pairs_csr([a, b], [(0, 1)], 0.9, 1.5)[0][-1]

    th = threading.Thread(target=work)
    th.start()
    th.join()
    assert out["bad"] != 0 and out["ok"] == 0 and out["dev"] == 0 and out["again"] == 0 and out["m1"] == out["m2"] > 0, out
    assert capi.lib.aps_get_device() == 0


def test_filter_aware_dismissal_changes_no_match(fm, monkeypatch):
    """The candidate kernel dismisses rows whose screened values already prove that the ratio / threshold filter drops
    them (no exact evaluation, no fallback).  The match lists must equal the oracle's and those of a run with the
    dismissal off (APS_MATCH_NO_PRUNE=1): planted correspondences at every noise level from clear matches to clear
    non-matches, ratios scattered around the boundary, several MaxRatio / MatchThreshold settings."""
    rng = np.random.default_rng(31)
    a, b, ia, ib = planted_pair(rng, 3000, 3500, 1500, noise=0.02)
    # spread the planted pairs' noise so that their ratios cover 0.05 .. 1
    noise = rng.uniform(0.0, 0.25, len(ib))[:, None]
    pert = a[ia] + noise * rng.standard_normal((len(ia), 128)).astype(np.float32)
    pert = np.maximum(pert, 0)
    pert /= np.linalg.norm(pert, axis=1, keepdims=True) + 1e-12
    b[ib] = pert.astype(np.float32)
    for ratio, thr, unique in ((0.6, 1.5, True), (0.8, 1.5, True), (0.95, 0.4, False), (1.0, 3.5, True), (0.3, 0.05, True)):
        monkeypatch.delenv("APS_MATCH_NO_PRUNE", raising=False)
        m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
        monkeypatch.setenv("APS_MATCH_NO_PRUNE", "1")
        m0, met0 = fm.matchFeaturesScratch(a, b, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
        om, omet = oracle.match_features(a, b, ratio, thr, unique, 2)
        assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet)), (ratio, thr)
        assert np.array_equal(m0, om) and np.array_equal(bits(met0), bits(omet))
    monkeypatch.delenv("APS_MATCH_NO_PRUNE", raising=False)
    # the boundary fixtures (d1/d2 within a few 1e-6 of r^2): the dismissal must stay clear of them
    a2, b2 = _boundary_sets((0.36, None))
    m, met = fm.matchFeaturesScratch(a2, b2, MatchThreshold=3.5, MaxRatio=0.6, Unique=False)
    om, omet = oracle.match_features(a2, b2, 0.6, 3.5, False, 0)
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))


def _screen_ab(fm, monkeypatch, a, b, ratio, thr, unique, normalize=2):
    """match list with the int8 screen, without it, and the oracle's"""
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
    monkeypatch.setenv("APS_MATCH_NO_SCREEN", "1")
    m0, met0 = fm.matchFeaturesScratch(a, b, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    om, omet = oracle.match_features(a, b, ratio, thr, unique, normalize)
    assert np.array_equal(m0, om) and np.array_equal(bits(met0), bits(omet))
    assert np.array_equal(m, om) and np.array_equal(bits(met), bits(omet))
    return len(om)


@pytest.mark.parametrize("n2", [2, 3, 31, 33, 255, 256, 257, 300, 513, 1025])
def test_int8_screen_ragged_column_counts(fm, monkeypatch, n2):
    """The int8 pre-pass streams 256-column tiles of 32-column blocks; the columns past the end of the last tile are
    re-reads of the last row and must never count as a second neighbour.  Planted matches sit in the last columns."""
    rng = np.random.default_rng(100 + n2)
    a, b, ia, ib = planted_pair(rng, 700, n2, min(n2, 200), noise=0.03)
    b[-1] = a[5] + 0.01 * rng.standard_normal(128).astype(np.float32)  # a clear match in the very last column
    b[-1] = np.maximum(b[-1], 0) / np.linalg.norm(np.maximum(b[-1], 0))
    k = _screen_ab(fm, monkeypatch, a, b, 0.6, 3.5, True)
    assert k > 0
    _screen_ab(fm, monkeypatch, a, b, 0.9, 0.5, False)


def test_int8_screen_adversarial_sets(fm, monkeypatch):
    """Inputs that stress the screen's bounds: signed data, duplicated best columns (ratio exactly 1), near-duplicates
    whose ratio sits in the band the int8 error cannot resolve, all-zero rows, a huge dynamic range within a set
    (tiny rows quantise to zero), and unnormalised 0..255 descriptors."""
    rng = np.random.default_rng(7)
    # signed unit rows with planted matches
    a = rng.standard_normal((1500, 128)).astype(np.float32)
    b = rng.standard_normal((1800, 128)).astype(np.float32)
    b[:600] = a[:600] + rng.uniform(0.0, 0.4, 600)[:, None].astype(np.float32) * rng.standard_normal((600, 128)).astype(np.float32)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    for ratio, thr, unique in ((0.6, 3.5, True), (0.8, 1.0, False), (1.0, 3.5, True)):
        _screen_ab(fm, monkeypatch, a, b, ratio, thr, unique)
    # duplicated and nearly duplicated best columns
    a2, b2, ia, ib = planted_pair(rng, 1200, 1500, 500, noise=0.01)
    b2[1000:1100] = b2[ib[:100]]                                   # exact copies: d1 == d2
    b2[1100:1200] = b2[ib[100:200]] + 0.004 * rng.standard_normal((100, 128)).astype(np.float32)  # d2 barely above d1
    b2[1100:1200] = np.maximum(b2[1100:1200], 0)
    b2[1100:1200] /= np.linalg.norm(b2[1100:1200], axis=1, keepdims=True)
    b2[1200:1300] = b2[ib[200:300]] + rng.uniform(0.02, 0.08, 100)[:, None].astype(np.float32) * rng.standard_normal((100, 128)).astype(np.float32)
    b2[1200:1300] = np.maximum(b2[1200:1300], 0)
    b2[1200:1300] /= np.linalg.norm(b2[1200:1300], axis=1, keepdims=True)
    for ratio in (0.36, 0.6, 0.85, 1.0):
        _screen_ab(fm, monkeypatch, a2, b2, ratio, 3.5, True)
        _screen_ab(fm, monkeypatch, a2, b2, ratio, 0.3, False)
    # zero rows and a 1e6 dynamic range inside both sets (rows far below the set's quantisation step)
    a3, b3, _, _ = planted_pair(rng, 900, 1100, 400, noise=0.02)
    a3[::7] *= 1e-6
    b3[::5] *= 1e-6
    a3[3] = 0
    b3[4] = 0
    for ratio, thr in ((0.6, 3.5), (0.9, 1e-3)):
        monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
        pp, ii, jj, met = fm.match_pairwise_csr([a3, b3], ratio, thr, True, normalize=0)
        monkeypatch.setenv("APS_MATCH_NO_SCREEN", "1")
        pp0, ii0, jj0, met0 = fm.match_pairwise_csr([a3, b3], ratio, thr, True, normalize=0)
        monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
        om, omet = oracle.match_features(a3, b3, ratio, thr, True, 0)
        assert np.array_equal(np.stack([ii, jj], 1), om) and np.array_equal(bits(met), bits(omet))
        assert np.array_equal(np.stack([ii0, jj0], 1), om) and np.array_equal(bits(met0), bits(omet))
    # unnormalised 0..255 descriptors: the reference normalises them (max > 2), the screen sees the normalised copy
    a4, b4, _, _ = planted_pair(rng, 1000, 1300, 500, noise=0.03, unit=False)
    _screen_ab(fm, monkeypatch, a4, b4, 0.6, 3.5, True)


def _screen_share(gpu):
    import ctypes

    rows, surv = ctypes.c_int64(-1), ctypes.c_int64(-1)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
    return rows.value, surv.value


def _exact_jobs(gpu):
    import ctypes

    jobs, ex = ctypes.c_int64(-1), ctypes.c_int64(-1)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_exact_jobs(ctypes.byref(jobs), ctypes.byref(ex)))
    return jobs.value, ex.value


def test_int8_screen_exact_integer_codes(fm, gpu, monkeypatch):
    """Round 6: sets of integers 0 .. 255 (SIFT descriptors as OpenCV quantises them) that are matched in normalised form take
    EXACT int8 codes u - 128 on both sides, the column term of the centring entering as the MFMA's C operand.  The match lists
    must equal the oracle's, those of the general codes (APS_MATCH_NO_EXACT=1) and those without the screen; the exact codes
    must dismiss more rows; ragged column counts, zero rows, saturated entries, duplicated columns, one column, and a set
    that is not integer-valued (whose jobs must take the general codes)."""
    rng = np.random.default_rng(61)
    a, b, _, ib = planted_pair(rng, 3000, 3500, 1200, noise=0.03, unit=False)
    assert a.max() > 2 and np.array_equal(a, np.rint(a))

    def shares(x, y, ratio, thr, unique, want_exact=True):
        monkeypatch.delenv("APS_MATCH_NO_EXACT", raising=False)
        m, met = fm.matchFeaturesScratch(x, y, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
        rows, s_exact = _screen_share(gpu)
        assert want_exact is None or _exact_jobs(gpu) == (1, 1 if want_exact else 0)
        monkeypatch.setenv("APS_MATCH_NO_EXACT", "1")
        m1, met1 = fm.matchFeaturesScratch(x, y, MatchThreshold=thr, MaxRatio=ratio, Unique=unique)
        _, s_general = _screen_share(gpu)
        assert _exact_jobs(gpu) == (1, 0)
        monkeypatch.delenv("APS_MATCH_NO_EXACT", raising=False)
        assert np.array_equal(m, m1) and np.array_equal(bits(met), bits(met1)) and rows == len(x)
        return len(m), s_exact, s_general

    for ratio, thr, unique in ((0.6, 3.5, True), (0.8, 1.5, False), (1.0, 3.5, True), (0.36, 0.2, True)):
        _screen_ab(fm, monkeypatch, a, b, ratio, thr, unique)
        k, s_exact, s_general = shares(a, b, ratio, thr, unique)
        assert s_exact <= s_general, (ratio, s_exact, s_general)
        assert k <= s_exact, (k, s_exact)
    # planted pairs whose ratios scatter around the boundary: here the general codes' error band keeps rows the exact codes
    # dismiss (a guard against a silently disabled exact mode)
    ah, bh, ia, ibh = planted_pair(rng, 3000, 3500, 1500, noise=0.02)
    noise = rng.uniform(0.0, 0.25, len(ibh))[:, None]
    pert = np.maximum(ah[ia] + noise * rng.standard_normal((len(ia), 128)).astype(np.float32), 0)
    bh[ibh] = (pert / (np.linalg.norm(pert, axis=1, keepdims=True) + 1e-12)).astype(np.float32)
    ah, bh = np.round(ah * 512).clip(0, 255).astype(np.float32), np.round(bh * 512).clip(0, 255).astype(np.float32)
    _screen_ab(fm, monkeypatch, ah, bh, 0.6, 3.5, True)
    k, s_exact, s_general = shares(ah, bh, 0.6, 3.5, True)
    assert k <= s_exact <= s_general, (k, s_exact, s_general)
    # Outlier divisors: a few columns whose norm is far from the set's (a descriptor concentrated in one or two bins saturates at
    # 255 and has t = 255 .. 360 instead of ~512).  Such a column would inflate the bounds of the 250 ordinary columns it shares a
    # tile with beyond any real best similarity, so a column set whose divisors spread by more than 2 % keeps the rounded codes:
    # same lists, and the screen dismisses as many rows as it does without the exact codes.
    bo = bh.copy()
    for k, (i0, v0, i1, v1) in enumerate(((3, 255, 3, 255), (10, 255, 77, 255), (5, 200, 99, 120))):
        bo[40 + k] = 0
        bo[40 + k, i0] = v0
        bo[40 + k, i1] = v1
    _screen_ab(fm, monkeypatch, ah, bo, 0.6, 3.5, True)
    k_o, s_o, s_og = shares(ah, bo, 0.6, 3.5, True, want_exact=False)
    assert k_o <= s_o == s_og <= 1.1 * s_general + 16, (k_o, s_o, s_og, s_general)
    _screen_ab(fm, monkeypatch, bo, ah, 0.6, 3.5, True)   # (as the ROW side the same set is no obstacle)
    shares(bo, ah, 0.6, 3.5, True, want_exact=True)
    # degenerate rows: a duplicated best column (d1 == d2), all-zero rows (divisor eps), saturated and one-hot rows (an
    # all-255 row has no exact code within the tries: its set falls back to the general codes)
    b[100] = b[ib[0]]
    b[101] = 0
    a[7] = 0
    a[8] = 255
    b[102] = 255
    a[9, :] = 0
    a[9, 5] = 255
    for ratio, thr, unique in ((0.6, 3.5, True), (0.8, 1.5, False), (1.0, 3.5, True)):
        _screen_ab(fm, monkeypatch, a, b, ratio, thr, unique)
        shares(a, b, ratio, thr, unique, want_exact=None)
    a[8], b[102] = a[10], b[103]  # without the all-255 rows both sets have exact codes again (zero rows fit any divisor)
    shares(a, b, 0.6, 3.5, True)
    a, b, _, ib = planted_pair(rng, 3000, 3500, 1200, noise=0.03, unit=False)
    # the form the SIFT stage hands over: the same integers divided by their norm in f32 (matched without normalisation)
    au = (a / np.sqrt((a * a).sum(1, dtype=np.float32))[:, None].clip(1e-30)).astype(np.float32)
    bu = (b / np.sqrt((b * b).sum(1, dtype=np.float32))[:, None].clip(1e-30)).astype(np.float32)
    assert au.max() <= 1.0
    _screen_ab(fm, monkeypatch, au, bu, 0.6, 3.5, True)
    k, s_exact, s_general = shares(au, bu, 0.6, 3.5, True)
    assert k <= s_exact <= s_general, (k, s_exact, s_general)
    ahu = (ah / np.sqrt((ah * ah).sum(1, dtype=np.float32))[:, None].clip(1e-30)).astype(np.float32)
    bhu = (bh / np.sqrt((bh * bh).sum(1, dtype=np.float32))[:, None].clip(1e-30)).astype(np.float32)
    _screen_ab(fm, monkeypatch, ahu, bhu, 0.6, 3.5, True)
    k, s_exact, s_general = shares(ahu, bhu, 0.6, 3.5, True)
    assert k <= s_exact <= s_general, (k, s_exact, s_general)
    for n2 in (1, 2, 3, 31, 255, 256, 257, 513, 1025):
        a2, b2, _, _ = planted_pair(rng, 700, n2, min(n2, 200), noise=0.03, unit=False)
        b2[-1] = a2[5]
        _screen_ab(fm, monkeypatch, a2, b2, 0.6, 3.5, True)
        _screen_ab(fm, monkeypatch, a2, b2, 0.9, 0.5, False)
    # a batch of several sets of unequal sizes through the pooled list pass
    sets = [planted_pair(rng, 900 + 211 * k, 10, 5, unit=False)[0] for k in range(4)]
    sets[2][:300] = sets[0][:300]
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    pp, ii, jj, met = fm.match_pairwise_csr(sets, 0.6, 3.5, True)
    assert _exact_jobs(gpu) == (6, 6)
    p = 0
    for j in range(1, 4):
        for i in range(j):
            om, omet = oracle.match_features(sets[i], sets[j], 0.6, 3.5, True, 2)
            sl = slice(pp[p], pp[p + 1])
            assert np.array_equal(np.stack([ii[sl], jj[sl]], 1), om) and np.array_equal(bits(met[sl]), bits(omet)), (i, j)
            p += 1
    # one set with a single non-integer entry: general codes for the three jobs it takes part in, same lists
    sets[1] = sets[1].copy()
    sets[1][3, 3] += 0.3
    pp, ii, jj, met = fm.match_pairwise_csr(sets, 0.6, 3.5, True)
    assert _exact_jobs(gpu) == (6, 3)
    p = 0
    for j in range(1, 4):
        for i in range(j):
            om, omet = oracle.match_features(sets[i], sets[j], 0.6, 3.5, True, 2)
            sl = slice(pp[p], pp[p + 1])
            assert np.array_equal(np.stack([ii[sl], jj[sl]], 1), om) and np.array_equal(bits(met[sl]), bits(omet)), (i, j)
            p += 1


def test_int8_screen_is_off_for_non_finite_sets(fm, monkeypatch):
    """A NaN or an infinity in a set makes every bound of the screen meaningless: such jobs must take the f16 / f32 path
    for every row (the set's max ||x||^2 is not finite, which switches the screen off)."""
    rng = np.random.default_rng(11)
    a, b, _, _ = planted_pair(rng, 600, 800, 300, noise=0.02)
    for poison in (np.nan, np.inf):
        bb = b.copy()
        bb[17, 3] = poison
        monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
        pp, ii, jj, met = fm.match_pairwise_csr([a, bb], 0.6, 3.5, True, normalize=0)
        om, omet = oracle.match_features(a, bb, 0.6, 3.5, True, 0)
        assert np.array_equal(np.stack([ii, jj], 1), om) and np.array_equal(bits(met), bits(omet))
        aa = a.copy()
        aa[9, 100] = poison
        pp, ii, jj, met = fm.match_pairwise_csr([aa, b], 0.6, 3.5, True, normalize=0)
        om, omet = oracle.match_features(aa, b, 0.6, 3.5, True, 0)
        assert np.array_equal(np.stack([ii, jj], 1), om) and np.array_equal(bits(met), bits(omet))


def test_int8_screen_dismisses_most_rows_and_reports_it(fm, gpu, monkeypatch):
    """Guard against a silently disabled screen: on unit SIFT-like sets with 30 % planted matches it must dismiss well
    over half of the rows (aps_match_screen_stats), and report zero rows when it is switched off."""
    import ctypes

    capi = gpu._capi
    rng = np.random.default_rng(5)
    a, b, _, _ = planted_pair(rng, 4000, 5000, 1200, noise=0.02)
    rows, surv = ctypes.c_int64(-1), ctypes.c_int64(-1)
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    m, _ = fm.matchFeaturesScratch(a, b, MatchThreshold=3.5, MaxRatio=0.6, Unique=True)
    capi.check(capi.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
    assert rows.value == 4000 and len(m) <= surv.value < 0.45 * rows.value, (rows.value, surv.value, len(m))
    monkeypatch.setenv("APS_MATCH_NO_SCREEN", "1")
    fm.matchFeaturesScratch(a, b, MatchThreshold=3.5, MaxRatio=0.6, Unique=True)
    capi.check(capi.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
    assert rows.value == 0 and surv.value == 0


def test_pairwise_more_than_65535_pairs(fm):
    """BASELINE configs[4] has 500 images = 124750 pairs: more pairs than one grid dimension holds (65535).  370 small
    descriptor sets (68265 pairs) go through the batched matcher - screen, list pass, filter, chunked emit - and a
    sample of pairs, spread over the whole list including both sides of the 65535 boundary, is compared with the oracle."""
    rng = np.random.default_rng(17)
    n_img = 370
    base = sift_like(rng, 400)
    descs = []
    for i in range(n_img):
        k = int(rng.integers(20, 60))
        d = base[rng.permutation(400)[:k]] + 0.01 * rng.standard_normal((k, 128)).astype(np.float32)
        d = np.maximum(d, 0)
        descs.append((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32))
    pp, ii, jj, met = fm.match_pairwise_csr(descs, 0.8, 1.0, True)
    order = fm.pair_order(n_img)
    assert len(order) == n_img * (n_img - 1) // 2 > 65535 and len(pp) == len(order) + 1
    picks = sorted(set(rng.integers(0, len(order), 60).tolist()) | {0, 65534, 65535, 65536, len(order) - 1})
    for p in picks:
        i, j = order[p]
        om, omet = oracle.match_features(descs[i], descs[j], 0.8, 1.0, True, 2)
        s, e = int(pp[p]), int(pp[p + 1])
        got = np.stack([np.asarray(ii[s:e]), np.asarray(jj[s:e])], 1)
        assert np.array_equal(got, om), (p, i, j)
        assert np.array_equal(bits(np.asarray(met[s:e])), bits(omet))


def test_pooled_survivor_tiles_of_unequal_sets_equal_per_job_lists_and_oracle(fm, gpu, monkeypatch):
    """The f16 list pass packs the int8 screen's survivors of all jobs that share a B set into common 512-row tiles
    (match_cand_f16_kernel's row_job form, round 4): every A-side quantity is then per ROW.  Sets of very different sizes
    (a tile's anchor job is the group's first - here the 37-row set - while its rows come from sets of up to 2100 rows),
    unit and unnormalised descriptors, near-ties: the lists must equal those of one list per job (APS_MATCH_NO_POOL=1), of
    the all-f32 kernel and of the oracle, pair by pair."""
    import ctypes

    rng = np.random.default_rng(77)
    base = sift_like(rng, 2600)
    sizes = [37, 2100, 640, 1500, 5, 980]
    for unit in (True, False):
        descs = []
        for n in sizes:
            keep = rng.permutation(2600)[:n]
            noise = rng.uniform(0.0, 0.08, n)[:, None]
            d = np.maximum(base[keep] + noise * rng.standard_normal((n, 128)).astype(np.float32), 0)
            d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
            descs.append(d if unit else np.round(d * 512).clip(0, 255).astype(np.float32))
        descs[3][7] = descs[1][11]                     # an exact duplicate across sets
        descs[1][13] = descs[1][12] + np.float32(1e-4)  # two near-identical columns: the rows near them need the exact path
        order = fm.pair_order(len(descs))
        monkeypatch.delenv("APS_MATCH_NO_POOL", raising=False)
        monkeypatch.delenv("APS_MATCH_MODE", raising=False)
        pp, ia, ib, met = fm.match_pairs_csr(descs, order, 0.8, 1.5, True)
        rows, surv = ctypes.c_int64(0), ctypes.c_int64(0)
        gpu._capi.check(gpu.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
        assert rows.value == sum(len(descs[i]) for i, _ in order) and surv.value > 2000  # the list pass has real work
        monkeypatch.setenv("APS_MATCH_NO_POOL", "1")
        qp, qa, qb, qm = fm.match_pairs_csr(descs, order, 0.8, 1.5, True)
        monkeypatch.delenv("APS_MATCH_NO_POOL")
        monkeypatch.setenv("APS_MATCH_MODE", "f32")
        fp, fa, fb_, fmet = fm.match_pairs_csr(descs, order, 0.8, 1.5, True)
        monkeypatch.delenv("APS_MATCH_MODE")
        for other in ((qp, qa, qb, qm), (fp, fa, fb_, fmet)):
            assert np.array_equal(pp, other[0]) and np.array_equal(ia, other[1]) and np.array_equal(ib, other[2])
            assert np.array_equal(met.view(np.uint32), other[3].view(np.uint32))
        assert pp[-1] > 1500
        for p, (i, j) in enumerate(order):
            om, omet = oracle.match_features(descs[i], descs[j], 0.8, 1.5, True, 2)
            got = np.stack([ia[pp[p]:pp[p + 1]], ib[pp[p]:pp[p + 1]]], axis=1).astype(np.int64)
            assert np.array_equal(got, om.astype(np.int64)), (unit, i, j)
            assert np.array_equal(met[pp[p]:pp[p + 1]].view(np.uint32), np.asarray(omet, np.float32).view(np.uint32)), (unit, i, j)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 63, 64, 65, 127, 129, 1000, 19801])
@pytest.mark.parametrize("layout", ["row", "col"])
def test_set_statistics_see_every_row(gpu, n, layout):
    """The proofs of the screening passes rest on statistics of a whole descriptor set (largest and smallest ||x||^2, largest
    and smallest element, largest rounding losses).  They are maxima over all rows, folded from per-workgroup maxima
    (prep_stats_reduce): a fold that misses a workgroup - the last, ragged one above all - would make bounds unsound without
    changing any result on ordinary data.  The extreme rows are planted at the first row, the last row and the first row of
    the last 64-row workgroup in turn, and the words are compared with the same quantities computed on the host."""
    import ctypes as C

    capi = gpu._capi
    rng = np.random.default_rng(1000 + n)
    f32 = np.float32

    def seq_sq(x):  # canonical ||x||^2: k-ascending f32 sum of f32 squares
        s = np.zeros(x.shape[0], f32)
        for k in range(x.shape[1]):
            s = (s + x[:, k] * x[:, k]).astype(f32)
        return s

    for where in sorted({0, n - 1, (n - 1) // 64 * 64}):
        x = (rng.random((n, 128), dtype=f32) * 0.1 + 0.01).astype(f32)  # entries 0.01 .. 0.11: ||x||^2 ~ 0.6
        x[where] = 0.0
        x[where, 0:128:2] = f32(0.9173)         # the largest norm and the largest element ...
        x[where, 5] = f32(-0.7331)              # ... and the smallest element live in the planted row
        small = (where + 1) % n
        if n > 1:
            x[small] = f32(1e-3) * (1.0 + rng.random(128, dtype=f32))  # the smallest norm next to it (cyclically)
        xin = np.ascontiguousarray(x) if layout == "row" else np.asfortranarray(x)
        ld = 128 if layout == "row" else n
        out = np.zeros(8, f32)
        capi.check(capi.lib.aps_match_set_stats(capi.ptr(xin), n, ld, capi.APS_ROWMAJOR if layout == "row" else capi.APS_COLMAJOR,
                                                0, capi.ptr(out)))
        sq = seq_sq(x)
        assert out[0] == sq.max() and out[6] == sq.min(), (n, where, out, sq.max(), sq.min())
        assert out[4] == x.max() and out[7] == x.min(), (n, where, out)
        # rounding losses: at least the true f64 loss of the worst row, at most a few per cent above it
        h = x.astype(np.float16).astype(np.float64)
        h[np.abs(h) < 6.103515625e-05] = 0.0
        dn = np.sqrt(((x.astype(np.float64) - h) ** 2).sum(1)).max()
        assert dn <= out[1] <= dn * 1.01 + 1e-12, (n, where, out[1], dn)
        step = (float(x.max()) - float(x.min())) / 255.0
        cb = np.rint(np.float64(f32(x.min()) * f32(f32(255.0) / f32(x.max() - x.min())))) + 128.0
        qb = np.clip(np.rint(x.astype(np.float64) / step) - cb, -128, 127)
        db = np.sqrt(((x.astype(np.float64) - (qb + cb) * step) ** 2).sum(1)).max()
        assert 0.9 * db <= out[5] <= 1.1 * db + 1e-9, (n, where, out[5], db)
        assert 0.0 <= out[2] < 1e-6 and out[3] == 0.0  # ([2]: what three f16 pieces lose of b2/2 when a set's norms lie far apart)
    # normalised rows: every norm is 1 up to rounding, the extremes follow the rows
    xn = (rng.random((n, 128), dtype=f32) * 50.0).astype(f32)
    out = np.zeros(8, f32)
    capi.check(capi.lib.aps_match_set_stats(capi.ptr(xn), n, 128, capi.APS_ROWMAJOR, 1, capi.ptr(out)))
    assert abs(out[0] - 1.0) < 1e-5 and abs(out[6] - 1.0) < 1e-5 and 0.0 <= out[7] < out[4] < 1.0


def test_int8_screening_kernels_hold_the_whole_register_file_of_their_simds(gpu):
    """The co-residency rule as the LOADED code object declares it (hipFuncGetAttributes; tests/test_abi.py reads the same
    from the code object's metadata without a device): every int8-MFMA kernel - both shapes, the list form and the pooled
    matcher's bounds form - holds 256 registers per lane at a 512-thread workgroup bound, i.e. two waves fill a SIMD's 512
    registers and no other kernel's wave can sit beside them.  The library itself refuses to launch them otherwise."""
    import ctypes
    for shape in (16, 32):
        for bounds in (0, 1):
            regs, thr = ctypes.c_int(0), ctypes.c_int(0)
            gpu._capi.check(gpu.lib.aps_match_screen_kernel_regs(shape, bounds, ctypes.byref(regs), ctypes.byref(thr)))
            assert (regs.value + 7) // 8 * 8 >= 256, (shape, bounds, regs.value)
            assert thr.value == 512, (shape, bounds, thr.value)
    regs, thr = ctypes.c_int(0), ctypes.c_int(0)
    assert gpu.lib.aps_match_screen_kernel_regs(8, 0, ctypes.byref(regs), ctypes.byref(thr)) == gpu._capi.APS_E_ARG


def test_pca2nn_argument_errors(gpu):
    """aps_match_pca2nn refuses what nearest2ApproxFloatFast's `arguments` block and the library's layout contract refuse:
    other descriptor lengths, empty sets, a non-positive component count, leading dimensions below the row length."""
    import ctypes
    lib, capi = gpu.lib, gpu._capi
    A = np.zeros((4, 128), np.float32)
    idx, d1, d2 = np.zeros(4, np.uint32), np.zeros(4, np.float32), np.zeros(4, np.float32)
    p = lambda x: x.ctypes.data  # noqa: E731
    call = lambda n1, lda, n2, ldb, dim, k: lib.aps_match_pca2nn(p(A), n1, lda, p(A), n2, ldb, dim, capi.APS_ROWMAJOR, k, 1, p(idx), p(d1), p(d2), None, None)  # noqa: E731
    assert call(4, 128, 4, 128, 64, 48) == capi.APS_E_DIM
    assert call(0, 128, 4, 128, 128, 48) == capi.APS_E_ARG
    assert call(4, 128, 0, 128, 128, 48) == capi.APS_E_ARG
    assert call(4, 128, 4, 128, 128, 0) == capi.APS_E_ARG
    assert call(4, 64, 4, 128, 128, 48) == capi.APS_E_DIM
    assert call(4, 128, 4, 128, 128, 48) == capi.APS_OK


def test_pairwise_operator_honours_the_matcher_switches(gpu):
    """featureMatchingPairwise.m:103-117 (getMatches): Matchingmethod = 'Approximate' sends every pair through
    matchFeaturesScratch with ApproxFloatNNMethod (pca2nn: the device PCA path, equal to the oracle's filtered result per pair);
    'Exhaustive' is the batched all-pairs call; useMATLABFeatureMatch = 1 (toolbox matchFeatures) has no device form."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    rng = np.random.default_rng(21)
    base = sift_like(rng, 900)
    descs = []
    for i in range(3):
        keep = rng.permutation(900)[: 500 + 40 * i]
        d = np.maximum(base[keep] + 0.02 * rng.standard_normal((len(keep), 128)).astype(np.float32), 0)
        descs.append((d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32))
    inp = {"Matchingthreshold": 1.5, "Ratiothreshold": 0.6, "useMATLABFeatureMatch": 0}
    ex = fm.featureMatchingPairwise(dict(inp, Matchingmethod="Exhaustive"), descs, 3)
    ap = fm.featureMatchingPairwise(dict(inp, Matchingmethod="Approximate", ApproxFloatNNMethod="pca2nn"), descs, 3)
    for (i, j) in fm.pair_order(3):
        assert ex[i][j].dtype == np.float64 and ap[i][j].dtype == np.float64 and ap[i][j].shape[1] == 2
        oi, od1, od2 = oracle.pca2nn(descs[i], descs[j], 48, True)
        want, _ = fm.filter_matches(oi, od1, od2, len(descs[j]), 0.6, 1.5, True)
        assert np.array_equal(ap[i][j], want.astype(np.float64)), (i, j)
        se, sa = {tuple(r) for r in ex[i][j].tolist()}, {tuple(r) for r in ap[i][j].tolist()}
        assert len(se & sa) >= 0.85 * len(se) > 50
    assert ap[1][0] is None and ap[0][0] is None
    with pytest.raises(NotImplementedError):
        fm.featureMatchingPairwise(dict(inp, useMATLABFeatureMatch=1), descs, 3)


def test_thread_stream_priority_recreates_the_stream_and_changes_no_result(gpu):
    """aps_set_thread_stream_priority (round 6: the main thread of a set-after-set loop runs on the device's highest priority
    level): every level gives a working stream, also after the thread's auxiliary streams exist, and the same match lists."""
    import threading

    capi = gpu._capi
    out = {}

    def work():
        capi.check(capi.lib.aps_set_thread_device(0))
        rng = np.random.default_rng(5)
        a, b = sift_like(rng, 400), sift_like(rng, 380)
        fmod = import_module(gpu.__name__ + ".featureMatching")
        res = []
        for level in (None, 1, 0, -1, 1):
            if level is not None:
                out.setdefault("rc", []).append(capi.lib.aps_set_thread_stream_priority(level))
            pp, ia, ib, _ = fmod.match_pairs_csr([a, b], [(0, 1)], 0.9, 1.5)
            res.append((int(pp[-1]), np.asarray(ia).tobytes(), np.asarray(ib).tobytes()))
        out["same"] = all(r == res[0] for r in res) and res[0][0] > 0

    th = threading.Thread(target=work)  # (a thread of its own: the suite's main thread keeps its stream)
    th.start()
    th.join()
    assert out.get("rc") == [0, 0, 0, 0] and out.get("same"), out
