"""Ad-hoc timing probe of the SIFT stage on synthetic 4K views (not the bench)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

fm = import_module(apsamd.__name__ + ".featureMatching")
synth = import_module(apsamd.__name__ + ".synth")
W, H, f = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160, 8000.0)
t0 = time.perf_counter()
imgs, cams = synth.make_scene(2, 1, W, H, f, device="cuda")
torch.cuda.synchronize()
print(f"scene gen {time.perf_counter()-t0:.2f}s")
inp = {"detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6}
for it in range(3):
    t0 = time.perf_counter()
    d, p = fm.sift_extract(inp, imgs[0], device_out=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"iter {it}: {dt*1e3:.1f} ms, {len(p)} features, {W*H/1e6/dt:.0f} MPix/s")
d2, p2 = fm.sift_extract(inp, imgs[1], device_out=True)
m, met = fm.matchFeaturesScratch(d, d2, MatchThreshold=1.5, MaxRatio=0.6)
K = cams[0]["K"]
p0 = p[m[:, 0] - 1]; p1 = p2[m[:, 1] - 1]
rays = np.linalg.solve(K, np.c_[p0, np.ones(len(p0))].T)
q = K @ (cams[1]["R"] @ cams[0]["R"].T @ rays)
err = np.linalg.norm((q[:2] / q[2]).T - p1, axis=1)
print(f"matches {len(m)}, median reproj err {np.median(err):.3f}px, <3px: {(err<3).mean():.3f}")
