"""Round 6 A/B on one box: the SIFT stage of the 64 x 4K scene (ten worker streams) and of 8 views with the descriptor kernel's
row-interval sweep (default) - this process - against the whole-square sweep (APS_DESCR_PLAIN=1, read once per process: run twice)."""
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
        pl.sift_many(inp, views)
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), float(np.median(ts))
timed(imgs, 2)
print("APS_DESCR_PLAIN =", os.environ.get("APS_DESCR_PLAIN"), " 64 views: min %.2f median %.2f ms;  8 views: min %.2f median %.2f ms" % (timed(imgs) + timed(imgs[::8])))
