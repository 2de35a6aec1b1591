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


if __name__ == "__main__":
    main()
