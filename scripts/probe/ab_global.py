"""A/B of the pooled matcher (featureMatchingGlobal) on the bench scene: default against APS_MATCH_NO_POOL=1, same process."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
imgs, _ = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
del imgs
for rep in range(3):
    for mode in ("pooled", "per-job"):
        if mode == "per-job":
            os.environ["APS_MATCH_NO_POOL"] = "1"
        else:
            os.environ.pop("APS_MATCH_NO_POOL", None)
        capi.profile_enable(2); capi.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pp, _, _ = fm.match_global_csr(descs, 0.6, 4, device_out=True)
        capi.check(capi.lib.aps_synchronize()); dt = time.perf_counter() - t0
        prof = capi.profile_all(); capi.profile_enable(False)
        print(f"{mode:8s} rep {rep}: {1e3 * dt:7.2f} ms  matches {int(pp[-1])}  kernels " +
              ", ".join(f"{k} {v[0]:.1f}" for k, v in prof.items() if v[0] > 0.5), flush=True)
