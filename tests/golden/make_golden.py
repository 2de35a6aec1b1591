"""Generates tests/golden/golden_v1.npz: seeded inputs and the ORACLE's outputs for every stage of the hot path.

The reference ships no fixtures of its own and cannot be executed here (MATLAB / OpenCV
absent), so these vectors do not pin the oracle to the reference; they pin the oracle AND the HIP path to each
other across machines, compilers and rounds: tests/test_golden.py replays them on the CPU (oracle) and on the
GPU (product) and demands bit equality.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import oracle  # noqa: E402
from util import planted_pair  # noqa: E402


def textured(rng, h, w):
    base = rng.random((h // 6 + 2, w // 6 + 2, 3))
    img = np.kron(base, np.ones((6, 6, 1)))[:h, :w] + 0.15 * rng.random((h, w, 3))
    for _ in range(2):  # cheap separable smoothing without scipy
        img = (np.roll(img, 1, 0) + 2 * img + np.roll(img, -1, 0)) / 4
        img = (np.roll(img, 1, 1) + 2 * img + np.roll(img, -1, 1)) / 4
    img = (img - img.min()) / (img.max() - img.min())
    return (img * 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20261003)
    out = {}
    a, b, _, _ = planted_pair(rng, 300, 400, 150)
    out["match_a"], out["match_b"] = a, b
    out["match_idx"], out["match_d1"], out["match_d2"] = oracle.match_2nn_ssd(a, b)
    m, met = oracle.match_features(a, b, 0.6, 1.5, True, 2)
    out["match_pairs"], out["match_metric"] = m, met
    H = np.array([[1.02, 0.03, 120.0], [-0.02, 0.98, -60.0], [2e-6, -1e-6, 1.0]])
    p1 = np.stack([rng.uniform(0, 4000, 500), rng.uniform(0, 2000, 500)], 1)
    q = np.c_[p1, np.ones(500)] @ H.T
    p2 = q[:, :2] / q[:, 2:3] + 0.3 * rng.standard_normal((500, 2))
    bad = rng.permutation(500)[:200]
    p2[bad] = np.stack([rng.uniform(0, 4000, 200), rng.uniform(0, 2000, 200)], 1)
    samples = np.stack([rng.permutation(500)[:4] + 1 for _ in range(564)]).astype(np.uint32)
    model, mask, found, trials = oracle.ransac_homography(p1, p2, samples, 5.5, 99.9, 500)
    out.update(ransac_p1=p1, ransac_p2=p2, ransac_samples=samples, ransac_model=model, ransac_mask=mask,
               ransac_found=np.array([found]), ransac_trials=np.array([trials]))
    img = textured(rng, 96, 128)
    d, l, aux = oracle.sift(img)
    out.update(sift_img=img, sift_desc=d, sift_loc=l, sift_aux=aux)
    C = rng.random((3, 40, 56, 3), dtype=np.float32)
    W = rng.random((3, 40, 56), dtype=np.float32)
    W[0, :, 28:] = 0
    out.update(mb_C=C, mb_W=W, mb_F=oracle.multiband_blend(C, W, 4, 1.0), lin_F=oracle.linear_blend(C, W))
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("wrote golden_v1.npz:", {k: v.shape for k, v in out.items()})


def main_v2():
    """golden_v2.npz: the rows of SURVEY section 8(c) that v1 does not hold -- Hamming 2-NN (with ties and an
    all-equal row), exact global kNN + the per-query filter on a 3-image toy pool, MLESAC, the planar image warp."""
    rng = np.random.default_rng(20261004)
    out = {}
    A = rng.integers(0, 256, (90, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (130, 32), dtype=np.uint8)
    B[7] = A[3]            # exact duplicate: d1 = 0
    B[20] = B[21] = A[5]   # tie: the first index wins (nearest2HammingExhaustiveMEX.cpp:63-68)
    A[11] = 0
    out["ham_A"], out["ham_B"] = A, B
    out["ham_idx"], out["ham_d1"], out["ham_d2"] = oracle.hamming_2nn(A, B)
    base = rng.gamma(0.6, 1.0, (260, 128)).astype(np.float32)
    descs = [np.maximum(base[rng.permutation(260)[:n]] + 0.02 * rng.standard_normal((n, 128)).astype(np.float32), 0)
             for n in (120, 140, 100)]
    pool = np.concatenate(descs).astype(np.float32)
    sq = np.zeros(len(pool), np.float32)
    for kk in range(128):
        sq = sq + pool[:, kk] * pool[:, kk]
    pool = (pool / np.sqrt(sq + np.float32(np.finfo(np.float32).eps))[:, None]).astype(np.float32)
    img = np.repeat(np.arange(1, 4, dtype=np.uint32), [120, 140, 100])
    loc = np.concatenate([np.arange(1, c + 1, dtype=np.uint32) for c in (120, 140, 100)])
    ni, nd = oracle.knn(pool, pool, 4)
    out.update(knn_pool=pool, knn_idx=ni, knn_dist=nd, gf_img=img, gf_loc=loc,
               gf_rows=oracle.global_filter(ni, nd, img, loc, 0.6))
    Ht = np.array([[1.02, 0.03, 12.0], [-0.02, 0.98, -7.0], [1e-5, -2e-5, 1.0]])
    p1 = rng.uniform(0, 500, (300, 2))
    q = np.c_[p1, np.ones(300)] @ Ht.T
    p2 = q[:, :2] / q[:, 2:]
    p2[:80] += rng.uniform(-60, 60, (80, 2))
    p2 += rng.normal(0, 0.3, p2.shape)
    samples = np.stack([rng.choice(300, 4, replace=False) + 1 for _ in range(1064)]).astype(np.uint32)
    model, mask, found, trials = oracle.mlesac_homography(p1, p2, samples, 2.0, 99.9, 1000)
    out.update(ml_p1=p1, ml_p2=p2, ml_samples=samples, ml_model=model, ml_mask=mask, ml_found=np.array([found]),
               ml_trials=np.array([trials]))
    imgw = rng.integers(0, 256, (70, 100, 3), dtype=np.uint8)
    Hw = np.array([[1.01, 0.02, 7.5], [-0.015, 0.99, -3.25], [1e-5, -2e-5, 1.0]])
    out.update(iw_img=imgw, iw_H=Hw, iw_view=np.array([80.0, 120.0, -5.5, -4.5, 1.0, 1.0]),
               iw_u8=oracle.image_warp_h(imgw, Hw, 80, 120, -5.5, -4.5, 1.0, 1.0, 9),
               iw_f32=oracle.image_warp_h(imgw[..., 0].astype(np.float32) / 255, Hw, 80, 120, -5.5, -4.5, 1.0, 1.0, 0.0))
    np.savez_compressed(os.path.join(HERE, "golden_v2.npz"), **out)
    print("wrote golden_v2.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.exists(os.path.join(HERE, "golden_v1.npz")) or "--v1" in sys.argv:
        main()
    main_v2()
