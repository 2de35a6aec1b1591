"""Timing of the bundle-adjustment pair blocks at the 64-view scene's scale (384 edges x ~3.7 k matches).
(The oracle needs 176 ms on one core for this batch.)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import apsamd
from importlib import import_module


def _rot(rng, scale=0.2):
    w = rng.normal(0, scale, 3)
    a = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / a
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def _batch(rng, sizes):
    """Random two-view scenes: per pair two cameras (+ slightly incremented copies) and noisy projections of m points."""
    packs, Uis, Ujs, ptr = [], [], [], [0]
    for m in sizes:
        cams = [{"f": float(rng.uniform(500, 900)), "cx": 320.0, "cy": 240.0, "R": _rot(rng)} for _ in range(2)]
        cams += [dict(c, f=c["f"] + rng.normal(0, 2), R=_rot(rng, 0.01) @ c["R"]) for c in cams]
        X = rng.normal(0, 1, (m, 3)) + np.array([0, 0, 4.0])
        pts = []
        for c in cams[:2]:
            K = np.array([[c["f"], 0, c["cx"]], [0, c["f"], c["cy"]], [0, 0, 1.0]])
            p = (K @ c["R"] @ X.T).T
            pts.append(p[:, :2] / p[:, 2:3] + rng.normal(0, 1.5, (m, 2)))
        packs.append(np.stack([np.concatenate([[c["f"], c["cx"], c["cy"]], c["R"].ravel(order="F")]) for c in cams]))
        Uis.append(pts[0])
        Ujs.append(pts[1])
        ptr.append(ptr[-1] + m)
    return np.concatenate(Uis), np.concatenate(Ujs), ptr, np.stack(packs)


ba = import_module(apsamd.__name__ + ".bundleAdjustment")
capi = apsamd._capi
rng = np.random.default_rng(13)
Ui, Uj, ptr, cams = _batch(rng, [int(v) for v in rng.integers(3000, 4500, 384)])
capi.profile_enable(True)
for it in range(3):
    capi.profile_reset()
    t0 = time.perf_counter()
    got = ba.ba_pair_blocks(Ui, Uj, ptr, cams, 2.0, True)
    dt = time.perf_counter() - t0
    prof = capi.profile_all()
print(f"{ptr[-1]} matches in 384 pairs: device kernel {prof['ba_pair_blocks'][0]:.3f} ms ({dt*1e3:.1f} ms with host staging of "
      f"{(Ui.nbytes + Uj.nbytes) / 1e6:.0f} MB); parity with the oracle: tests/test_ba_gpu.py::test_bench_scale_batch")
