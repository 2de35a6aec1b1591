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


def main_v3():
    """golden_v3.npz: the rows of SURVEY section 8(f) -- imresize (bicubic shrink + bilinear size form), the bundle
    adjustment's per-pair normal-equation blocks (both directions, Huber outliers, a degenerate depth), and the crop
    rectangle (holes, a bay, content up to the last column, white canvas)."""
    rng = np.random.default_rng(20261004)
    out = {}
    out["rs_img"] = textured(rng, 97, 131)
    out["rs_bicubic_037"] = oracle.imresize_u8(out["rs_img"], 0.37, "bicubic")
    out["rs_bilinear_40x150"] = oracle.imresize_u8(out["rs_img"], (40, 150), "bilinear")

    def rot(scale):
        w = rng.normal(0, scale, 3)
        a = np.linalg.norm(w)
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / a
        return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)

    packs, Uis, Ujs, ptr = [], [], [], [0]
    for m in (5, 70, 0, 129):
        cams = [{"f": float(rng.uniform(500, 900)), "cx": 320.0, "cy": 240.0, "R": rot(0.2)} for _ in range(2)]
        cams += [dict(c, f=c["f"] + rng.normal(0, 2), R=rot(0.01) @ c["R"]) for c in cams]
        X = rng.normal(0, 1, (max(m, 1), 3)) + np.array([0, 0, 4.0])
        pts = []
        for c in cams[:2]:
            K = np.array([[c["f"], 0, c["cx"]], [0, c["f"], c["cy"]], [0, 0, 1.0]])
            p = (K @ c["R"] @ X.T).T
            pts.append((p[:, :2] / p[:, 2:3] + rng.normal(0, 1.5, (max(m, 1), 2)))[:m])
        if m > 4:
            pts[0][1] += 150.0  # beyond sigmaHuber
        packs.append(np.stack([np.concatenate([[c["f"], c["cx"], c["cy"]], c["R"].ravel(order="F")]) for c in cams]))
        Uis.append(pts[0])
        Ujs.append(pts[1])
        ptr.append(ptr[-1] + m)
    out["ba_Ui"], out["ba_Uj"] = np.concatenate(Uis), np.concatenate(Ujs)
    out["ba_ptr"], out["ba_cams"] = np.array(ptr, np.int64), np.stack(packs)
    out["ba_both"] = oracle.ba_pair_blocks(out["ba_Ui"], out["ba_Uj"], ptr, out["ba_cams"], 2.0, True)
    out["ba_one"] = oracle.ba_pair_blocks(out["ba_Ui"], out["ba_Uj"], ptr, out["ba_cams"], 2.0, False)

    h, w = 120, 190
    yy, xx = np.mgrid[0:h, 0:w]
    m = ((yy - 60) / 50.0) ** 2 + ((xx - 100) / 92.0) ** 2 < 1
    m &= ~(((yy - 50) ** 2 + (xx - 60) ** 2) < 64)        # enclosed hole
    m &= ~((yy < 35) & (np.abs(xx - 120) < 6))              # bay open to the top
    m[40:80, 150:190] = True                                # content up to the last column
    crop = np.zeros((h, w, 3), np.uint8)
    crop[m] = rng.integers(1, 256, (int(m.sum()), 3), dtype=np.uint8)
    out["crop_img"] = crop
    r, ok, dbg = oracle.crop_rect(crop, False, 0)
    out["crop_black"] = np.array(list(r) + [int(ok)] + list(dbg), np.int64)
    white = 255 - crop
    r, ok, dbg = oracle.crop_rect(white, True, 250)
    out["crop_white"] = np.array(list(r) + [int(ok)] + list(dbg), np.int64)
    np.savez_compressed(os.path.join(HERE, "golden_v3.npz"), **out)
    print("wrote golden_v3.npz:", {k: v.shape for k, v in out.items()})


def main_v4():
    """golden_v4.npz (round 3): imageWarp 'nearest' / 'bicubic' on uint8 and single images, the pooled matcher's CSR lists
    on a pool large enough for the screened search, and the two-stage 'fit' resize of resizeImagesToLimits."""
    rng = np.random.default_rng(20261004)
    out = {}
    img = textured(rng, 70, 96)
    Hm = np.array([[0.96, 0.06, 5.5], [-0.05, 1.03, -2.25], [3e-5, -2e-5, 1.0]])
    out["w_img"], out["w_H"] = img, Hm
    out["w_view"] = np.array([80.0, 110.0, -4.5, -3.5, 1.0, 1.0])  # rows, cols, x0, y0, sx, sy
    f32 = (img[..., 1].astype(np.float32) / 255.0)
    for m in ("nearest", "bicubic"):
        out[f"w_u8_{m}"] = oracle.image_warp_h(img, Hm, 80, 110, -4.5, -3.5, 1.0, 1.0, 9, method=m)
        out[f"w_f32_{m}"] = oracle.image_warp_h(f32, Hm, 80, 110, -4.5, -3.5, 1.0, 1.0, 0.25, method=m)
    # pooled matcher: 5 images of ~1800 descriptors sharing planted correspondences (pool > 8192 rows)
    from util import sift_like

    base = sift_like(rng, 2400)
    counts = []
    pool = []
    for i in range(5):
        keep = rng.permutation(2400)[: 1500 + 50 * i]
        noise = rng.uniform(0.0, 0.1, len(keep))[:, None]
        d = np.maximum(base[keep] + noise * rng.standard_normal((len(keep), 128)).astype(np.float32), 0)
        d = np.concatenate([(d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32), sift_like(rng, 250)])
        pool.append(d.astype(np.float32))
        counts.append(len(d))
    allp = np.concatenate(pool)
    sq = np.zeros(len(allp), np.float32)
    for kk in range(128):
        sq = sq + allp[:, kk] * allp[:, kk]
    normed = (allp / np.sqrt(sq + np.float32(np.finfo(np.float32).eps))[:, None]).astype(np.float32)
    imgi = np.repeat(np.arange(1, 6, dtype=np.uint32), counts)
    loc = np.concatenate([np.arange(1, c + 1, dtype=np.uint32) for c in counts])
    ni, nd = oracle.knn(normed, normed, 4)
    out["gm_pool"] = allp.astype(np.float16).astype(np.float32)  # (stored at half precision: the fixture stays small)
    # recompute on the stored values so that the fixture is self-consistent
    allp = out["gm_pool"]
    sq = np.zeros(len(allp), np.float32)
    for kk in range(128):
        sq = sq + allp[:, kk] * allp[:, kk]
    normed = (allp / np.sqrt(sq + np.float32(np.finfo(np.float32).eps))[:, None]).astype(np.float32)
    ni, nd = oracle.knn(normed, normed, 4)
    out["gm_counts"] = np.asarray(counts, np.int64)
    out["gm_rows"] = oracle.global_filter(ni, nd, imgi, loc, 0.6)
    # resizeImagesToLimits 'fit': shrink by the scalar form, then to the common largest size
    big = textured(rng, 120, 200)
    out["fit_img"] = big
    s1 = oracle.imresize_u8(big, 0.4, "bicubic")
    out["fit_stage1"] = s1
    out["fit_stage2"] = oracle.imresize_u8(s1, (64, 80), "bicubic")
    np.savez_compressed(os.path.join(HERE, "golden_v4.npz"), **out)
    print("wrote golden_v4.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    if not os.path.exists(os.path.join(HERE, "golden_v1.npz")) or "--v1" in sys.argv:
        main()
    if not os.path.exists(os.path.join(HERE, "golden_v2.npz")) or "--v2" in sys.argv:
        main_v2()
    main_v3()
    main_v4()
