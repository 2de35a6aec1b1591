"""GPU parity for the global matcher's device side (exact kNN, per-query filter) and the Hamming 2-NN."""
from importlib import import_module

import numpy as np
import pytest

import oracle
from util import bits, sift_like

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm(gpu):
    return import_module(gpu.__name__ + ".featureMatching")


@pytest.mark.parametrize("ft,fq,k", [(300, 300, 4), (1000, 257, 4), (64, 500, 8), (5, 40, 4), (3000, 3000, 2)])
def test_knn_bit_exact(fm, ft, fq, k):
    rng = np.random.default_rng(ft + fq + k)
    train = sift_like(rng, ft)
    query = train if ft == fq else sift_like(rng, fq)
    train[ft // 2] = train[1]  # an exact duplicate: ties resolve to the lower index
    idx, dist = fm.flann_knn_win(train, query, k)
    oi, od = oracle.knn(train, query, k)
    assert np.array_equal(idx, oi)
    assert np.array_equal(bits(dist), bits(od))
    if ft == fq:
        assert np.array_equal(idx[:, 0][np.arange(ft) != ft // 2], (np.arange(ft) + 1)[np.arange(ft) != ft // 2])


def test_knn_fewer_train_rows_than_k_and_errors(fm, gpu):
    rng = np.random.default_rng(1)
    t, q = sift_like(rng, 2), sift_like(rng, 7)
    idx, dist = fm.flann_knn_win(t, q, 4)
    assert np.all(idx[:, 2:] == 0) and np.all(np.isinf(dist[:, 2:])) and np.all(idx[:, :2] > 0)
    with pytest.raises(ValueError):
        fm.flann_knn_win(t, q, 0)
    with pytest.raises(gpu.ApsError):
        fm.flann_knn_win(t, q, 9)


def test_global_matcher_equals_oracle_chain(fm):
    rng = np.random.default_rng(2)
    base = sift_like(rng, 700)
    descs = []
    for i in range(4):
        keep = rng.permutation(700)[: 350 + 40 * i]
        d = base[keep] + 0.02 * rng.standard_normal((len(keep), 128)).astype(np.float32)
        descs.append(np.maximum(d, 0).astype(np.float32))
    inp = {"k": 4, "Ratiothreshold": 0.6}
    cells = fm.featureMatchingGlobal(inp, descs, 4)
    # oracle chain: same pooling/normalisation, exact kNN, sequential filter
    pool = np.concatenate(descs)
    sq = np.zeros(len(pool), np.float32)
    for kk in range(128):
        sq = sq + pool[:, kk] * pool[:, kk]
    pool = (pool / np.sqrt(sq + np.float32(np.finfo(np.float32).eps))[:, None]).astype(np.float32)
    counts = [len(d) for d in descs]
    img = np.repeat(np.arange(1, 5, dtype=np.uint32), counts)
    loc = np.concatenate([np.arange(1, c + 1, dtype=np.uint32) for c in counts])
    ni, nd = oracle.knn(pool, pool, 4)
    rows = oracle.global_filter(ni, nd, img, loc, 0.6)
    total = 0
    for i in range(4):
        for j in range(i + 1, 4):
            exp = rows[(rows[:, 0] == i + 1) & (rows[:, 1] == j + 1)][:, 2:]
            got = cells[i][j]
            if len(exp) == 0:
                assert got is None
            else:
                assert np.array_equal(got, exp.astype(np.float64)), (i, j)
            total += len(exp)
    assert total > 300


def test_hamming_equals_oracle_including_edges(fm):
    rng = np.random.default_rng(3)
    for nb in (32, 64, 17):
        A = rng.integers(0, 256, (300, nb), dtype=np.uint8)
        B = rng.integers(0, 256, (777, nb), dtype=np.uint8)
        B[5] = A[0]
        B[600] = A[0]          # equal best distances: first wins (strict <), later one becomes second via <=
        B[10] = B[11]
        got = fm.nearest2HammingExhaustiveMEX(A, B)
        exp = oracle.hamming_2nn(A, B)
        for g, e in zip(got, exp):
            assert np.array_equal(g, e)
    one = fm.nearest2HammingExhaustiveMEX(A, B[:1])
    assert np.all(one[2] == 17 * 8) and np.all(one[0] == 1)
    none = fm.nearest2HammingExhaustiveMEX(A, B[:0])
    assert np.all(none[0] == 0) and np.all(np.isnan(none[1])) and np.all(np.isnan(none[2]))
    with pytest.raises(ValueError):
        fm.nearest2HammingExhaustiveMEX(A, B[:, :8])
    with pytest.raises(TypeError):
        fm.nearest2HammingExhaustiveMEX(A.astype(np.float32), B)


@pytest.mark.parametrize("nb,k", [(32, 4), (64, 4), (32, 1), (32, 8), (20, 3)])
def test_hamming_knn_equals_oracle_and_known_answers(gpu, nb, k):
    """flann_knn_win on uint8 descriptors ('bf' = BFMatcher knnMatch and 'flann' = LSH in flann_knn.cpp:199-240): exact
    Hamming k-NN, ascending, ties -> lower index, index 0 / Inf where the train set has fewer than k rows."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    rng = np.random.default_rng(nb * 10 + k)
    train = rng.integers(0, 256, (700, nb), dtype=np.uint8)
    query = rng.integers(0, 256, (333, nb), dtype=np.uint8)
    train[600:610] = train[100:110]                        # duplicated train rows: ties -> the lower index wins
    query[:50] = train[rng.integers(0, 700, 50)]          # exact copies: distance 0 first
    query[50:60] = train[100:110]
    for method in ("bf", "flann"):
        idx, dist = fm.flann_knn_win(train, query, k, method)
        oi, od = oracle.knn_hamming(train, query, k)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert np.all(dist[:50, 0] == 0) and np.all(np.diff(dist, axis=1) >= 0)
    if k >= 2:
        assert np.array_equal(idx[50:60, 0], np.arange(101, 111)) and np.array_equal(idx[50:60, 1], np.arange(601, 611))
    # brute force in numpy on a few rows
    for i in (0, 77, 332):
        d = np.unpackbits(train ^ query[i], axis=1).sum(1)
        order = np.lexsort((np.arange(len(d)), d))[:k]
        assert np.array_equal(idx[i].astype(np.int64) - 1, order) and np.array_equal(dist[i], d[order].astype(np.float32))
    # fewer train rows than k
    idx, dist = fm.flann_knn_win(train[:2], query[:5], 4, "bf")
    assert np.all(idx[:, 2:] == 0) and np.all(np.isinf(dist[:, 2:])) and np.all(idx[:, :2] > 0)
    with pytest.raises(ValueError):
        fm.flann_knn_win(train, query[:, :nb - 1], k, "bf")
    with pytest.raises(ValueError):
        fm.flann_knn_win(train.astype(np.float32), query.astype(np.float32), k, "bf")


def test_blocked_screened_knn_equals_f32_path_and_oracle(gpu, monkeypatch):
    """aps_knn_global on a pool against itself (featureMatchingGlobal's call) takes the blocked path: exact own-block lists
    + f16-screened top-3 of every other block + merge.  Indices and distance bits must equal the all-f32 kernel
    (APS_KNN_MODE=f32) and the oracle - with planted near-duplicates across blocks (three copies of a row in ANOTHER block
    defeat the three-per-block lists and must come back through the whole-pool fallback) and exact duplicates (ties)."""
    fm = import_module(gpu.__name__ + ".featureMatching")
    rng = np.random.default_rng(77)
    x = sift_like(rng, 9000)
    # near-duplicates in other blocks (block size 2048 below): row 100's copies at 3000, 5000, 7000; three in ONE block
    for dst, eps in ((3000, 1e-3), (5000, 2e-3), (7000, 3e-3), (4100, 1e-3), (4200, 2e-3), (4300, 3e-3)):
        src = 100 if dst in (3000, 5000, 7000) else 200
        v = x[src] + eps * rng.standard_normal(128).astype(np.float32)
        x[dst] = (np.maximum(v, 0) / np.linalg.norm(np.maximum(v, 0))).astype(np.float32)
    x[8000] = x[10]      # exact duplicates: tie broken by index
    x[8001] = x[10]
    monkeypatch.setenv("APS_KNN_BLOCK", "2048")
    monkeypatch.delenv("APS_KNN_MODE", raising=False)
    idx, dist = fm.flann_knn_win(x, x, 4)
    monkeypatch.setenv("APS_KNN_MODE", "f32")
    idx0, dist0 = fm.flann_knn_win(x, x, 4)
    monkeypatch.delenv("APS_KNN_MODE")
    assert np.array_equal(idx, idx0) and np.array_equal(bits(dist), bits(dist0))
    oi, od = oracle.knn(x, x, 4)
    assert np.array_equal(idx, oi) and np.array_equal(bits(dist), bits(od))
    assert set(idx[100, :4].tolist()) == {101, 3001, 5001, 7001}
    assert set(idx[200, :4].tolist()) == {201, 4101, 4201, 4301}
    assert idx[10, 0] == 11 and idx[10, 1] == 8001 and idx[10, 2] == 8002 and dist[10, 1] == dist[10, 0]
    # k < 4 and the default block size on a larger pool (two blocks of 20480): against the f32 path
    monkeypatch.delenv("APS_KNN_BLOCK")
    y = sift_like(rng, 24000)
    for k in (2, 4):
        monkeypatch.delenv("APS_KNN_MODE", raising=False)
        a_i, a_d = fm.flann_knn_win(y, y, k)
        monkeypatch.setenv("APS_KNN_MODE", "f32")
        b_i, b_d = fm.flann_knn_win(y, y, k)
        assert np.array_equal(a_i, b_i) and np.array_equal(bits(a_d), bits(b_d))
    monkeypatch.delenv("APS_KNN_MODE")


def test_screened_pooled_matcher_equals_plain_search_and_oracle(gpu, monkeypatch):
    """featureMatchingGlobal through aps_knn_global_screened: the int8 screen dismisses the queries whose two nearest
    cross-image rows provably fail the ratio test, the others get their exact four nearest.  The CSR lists must equal
    those of the plain exact search (APS_KNN_MODE=f32) and of the oracle chain on a pool that has clear matches, matches
    near the ratio boundary, near-duplicates INSIDE an image (they crowd the four nearest: fewer than two cross-image
    neighbours remain), exact duplicates across images (ties by index) and an empty image."""
    import ctypes

    fm = import_module(gpu.__name__ + ".featureMatching")
    rng = np.random.default_rng(123)
    base = sift_like(rng, 2600)
    descs = []
    for i in range(6):
        keep = rng.permutation(2600)[: 1500 + 60 * i]
        noise = rng.uniform(0.0, 0.12, len(keep))[:, None]   # ratios from clear matches to clear non-matches
        d = np.maximum(base[keep] + noise * rng.standard_normal((len(keep), 128)).astype(np.float32), 0)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        own = sift_like(rng, 300)                            # rows only this image has
        descs.append(np.concatenate([d, own]).astype(np.float32))
    descs[2][5] = descs[2][4] + np.float32(1e-3) * rng.standard_normal(128).astype(np.float32)   # crowding inside image 2
    descs[2][6] = descs[2][4] + np.float32(2e-3) * rng.standard_normal(128).astype(np.float32)
    descs[2][7] = descs[2][4] + np.float32(3e-3) * rng.standard_normal(128).astype(np.float32)
    descs[3][9] = descs[1][11]                               # exact duplicates across images
    descs[4][9] = descs[1][11]
    descs.insert(3, np.zeros((0, 128), np.float32))          # an image without features
    n_img = len(descs)
    assert sum(len(d) for d in descs) > 8192                 # the screened path, not the small-pool shortcut
    monkeypatch.delenv("APS_KNN_MODE", raising=False)
    pp, oi, oj = fm.match_global_csr(descs, 0.6, 4)
    rows, surv = ctypes.c_int64(0), ctypes.c_int64(0)
    gpu._capi.check(gpu.lib.aps_knn_global_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
    assert rows.value == sum(len(d) for d in descs) and 0 < surv.value < 0.7 * rows.value, (rows.value, surv.value)
    monkeypatch.setenv("APS_KNN_MODE", "f32")
    pp0, oi0, oj0 = fm.match_global_csr(descs, 0.6, 4)
    monkeypatch.delenv("APS_KNN_MODE")
    assert np.array_equal(pp, pp0) and np.array_equal(oi, oi0) and np.array_equal(oj, oj0) and pp[-1] > 1500
    monkeypatch.setenv("APS_MATCH_NO_SCREEN", "1")           # the f16 path on every row: same lists again
    pp1, oi1, oj1 = fm.match_global_csr(descs, 0.6, 4)
    monkeypatch.delenv("APS_MATCH_NO_SCREEN")
    assert np.array_equal(pp, pp1) and np.array_equal(oi, oi1) and np.array_equal(oj, oj1)
    # the oracle chain
    pool = np.concatenate(descs)
    sq = np.zeros(len(pool), np.float32)
    for kk in range(128):
        sq = sq + pool[:, kk] * pool[:, kk]
    pool = (pool / np.sqrt(sq + np.float32(np.finfo(np.float32).eps))[:, None]).astype(np.float32)
    counts = [len(d) for d in descs]
    img = np.repeat(np.arange(1, n_img + 1, dtype=np.uint32), counts)
    loc = np.concatenate([np.arange(1, c + 1, dtype=np.uint32) for c in counts])
    ni, nd = oracle.knn(pool, pool, 4)
    want = oracle.global_filter(ni, nd, img, loc, 0.6)
    for p, (i, j) in enumerate(fm.pair_order(n_img)):
        exp = want[(want[:, 0] == i + 1) & (want[:, 1] == j + 1)][:, 2:]
        got = np.stack([oi[pp[p]:pp[p + 1]], oj[pp[p]:pp[p + 1]]], axis=1)
        assert np.array_equal(got.astype(np.int64), exp.astype(np.int64)), (i, j)
    # a pool whose (row, image) table exceeds 2^31 slots (BASELINE configs[4]) is searched in passes over ranges of query
    # images; APS_KNN_SLOT_CAP lowers the budget so that this pool takes that path too: three passes, two, one image per pass
    for cap in (7 * 12000, 7 * 6000, 1):
        monkeypatch.setenv("APS_KNN_SLOT_CAP", str(cap))
        ppc, oic, ojc = fm.match_global_csr(descs, 0.6, 4)
        gpu._capi.check(gpu.lib.aps_knn_global_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
        monkeypatch.delenv("APS_KNN_SLOT_CAP")
        assert np.array_equal(pp, ppc) and np.array_equal(oi, oic) and np.array_equal(oj, ojc), cap
        assert rows.value == sum(len(d) for d in descs) and 0 < surv.value < 0.7 * rows.value
    # other ratios: a tight one (few queries pass) and one above 1 (nothing may be dismissed wrongly)
    for ratio in (0.3, 0.95):
        a = fm.match_global_csr(descs, ratio, 4)
        monkeypatch.setenv("APS_KNN_MODE", "f32")
        b = fm.match_global_csr(descs, ratio, 4)
        monkeypatch.delenv("APS_KNN_MODE")
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), ratio
