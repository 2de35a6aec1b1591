"""Round 6 (review weak 10): is `cpu_baseline.modelled_64_views` a fair model?  The bench times the oracle chain on a 2 x 2 block
of the 4K views and scales its pieces - SIFT by views, the exhaustive matcher by pairs, RANSAC by candidate pairs, the render
by canvas area.  Here the same chain is TIMED on a 3 x 3 block (9 views, 36 pairs) and compared with what the 2 x 2 sample
predicts for it with the same rules.  Oracle = test infrastructure, C + OpenMP on this box's cores.
usage: python scripts/validate_cpu_model.py  (on a GPU box: the views come from the device-side scene generator)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import apsamd, oracle
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
rp = import_module(apsamd.__name__ + ".renderPanorama")
W, H, F, OVERLAP, FINEST = 3840, 2160, 8000.0, 0.4, 16.0
inp = pl.default_input(bands=5)


def chain(nx, ny):
    imgs, cams = synth.make_scene(nx, ny, W, H, F, OVERLAP, device="cuda", finest_px=FINEST)
    imgs = [i.cpu().numpy() for i in imgs]
    n = len(imgs)
    t0 = time.perf_counter()
    feats = [oracle.sift(im, inp["Sigma"], inp["NumLayersInOctave"], inp["ContrastThreshold"], inp["EdgeThreshold"]) for im in imgs]
    t_sift = time.perf_counter() - t0
    t0 = time.perf_counter()
    matches = {(i, j): oracle.match_features(feats[i][0], feats[j][0], inp["Ratiothreshold"], inp["Matchingthreshold"], True, 2)[0]
               for j in range(1, n) for i in range(j)}
    t_match = time.perf_counter() - t0
    t0 = time.perf_counter()
    rng = np.random.default_rng(0)
    n_cand = 0
    for (i, j), m in matches.items():
        if len(m) >= 4 * 6:  # (candidate pairs as imageMatching.m selects them: enough matches to be worth a RANSAC)
            n_cand += 1
            smp = np.stack([rng.permutation(len(m))[:4] + 1 for _ in range(564)]).astype(np.uint32)
            oracle.ransac_homography(feats[j][1][m[:, 1] - 1], feats[i][1][m[:, 0] - 1], smp, inp["maxDistance"], inp["inliersConfidence"], inp["maxIter"])
    t_ransac = time.perf_counter() - t0
    geo = rp.canvas_geometry(cams, [(H, W, 3)] * n, "spherical", 0, rp.default_opts({"anglePower": 2}, cams, 0))
    t0 = time.perf_counter()
    oracle.render(imgs, cams, geo, (2048, 2048), 2.0, "multiband", 5, 1.0)
    t_render = time.perf_counter() - t0
    return {"views": n, "pairs": n * (n - 1) // 2, "cand": n_cand, "area": float(geo["W"] * geo["H"]), "sift": t_sift, "match": t_match,
            "ransac": t_ransac, "render": t_render, "total": t_sift + t_match + t_ransac + t_render}


a = chain(2, 2)
b = chain(3, 3)
pred = {"sift": a["sift"] * b["views"] / a["views"], "match": a["match"] * b["pairs"] / a["pairs"],
        "ransac": a["ransac"] * b["cand"] / max(a["cand"], 1), "render": a["render"] * b["area"] / a["area"]}
pred["total"] = sum(pred.values())
print(f"oracle on {oracle.NUM_THREADS} threads, 3840x2160 views of the bench scene")
print("2 x 2 sample :", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in a.items()})
print("3 x 3 timed  :", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in b.items()})
print("3 x 3 modelled from the sample (SIFT x views, match x pairs, RANSAC x candidate pairs, render x canvas area):", {k: round(v, 2) for k, v in pred.items()})
for k in ("sift", "match", "ransac", "render", "total"):
    print(f"  {k:7s}: timed {b[k]:7.2f} s, modelled {pred[k]:7.2f} s, ratio {b[k] / max(pred[k], 1e-9):.3f}")
