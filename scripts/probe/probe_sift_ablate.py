"""Round 6: what the SIFT stage of the 64 x 4K scene (ten worker streams) costs when parts of the per-view chain are cut off
(timing build, APS_SIFT_ABLATE: 2 = the Gaussian pyramid only, 1 = pyramid + extrema sweep, unset / 0 = everything).  One setting
per process (the switch is read once)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
capi = apsamd._capi
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
def timed(views, reps=5):
    ts = []
    for _ in range(reps):
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            pl.sift_many(inp, views)
        except Exception as e:
            pass
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), float(np.median(ts))
timed(imgs, 2)
print("APS_SIFT_ABLATE =", os.environ.get("APS_SIFT_ABLATE"), " 64 views: min %.2f median %.2f ms;  8 views: min %.2f median %.2f ms" % (timed(imgs) + timed(imgs[::8])))
