"""Round 6: the distribution of the row divisors t = ||u|| of the bench scene's SIFT descriptors (integers u over their norm):
the exact screen's only slack is the spread [tmin, tmax] of a column set."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs[:6])]
for d in descs:
    x = d.cpu().numpy()
    xm = np.where(x > 0, x, np.inf).min(1)
    ok = np.isfinite(xm)
    t = np.zeros(len(x))
    for m in range(1, 25):
        u = np.rint(x * (m / xm)[:, None])
        good = (np.abs(x * (m / xm)[:, None] - u) < 1e-3 * np.maximum(u, 1)).all(1) & (t == 0) & ok
        t[good] = np.sqrt((u[good] ** 2).sum(1))
    q = np.percentile(t, [0, 0.01, 0.1, 1, 50, 99, 99.9, 99.99, 100])
    print(len(x), "rows; t percentiles 0/0.01/0.1/1/50/99/99.9/99.99/100:", np.round(q, 2), " tmin/tmax %.4f; 0.1-99.9%%: %.4f" % (q[0] / q[-1], q[2] / q[6]),
          " rows below 0.99 median: %d, above 1.01 median: %d" % ((t < 0.99 * q[4]).sum(), (t > 1.01 * q[4]).sum()))
