import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
rp = import_module(apsamd.__name__ + ".renderPanorama")
pl = import_module(apsamd.__name__ + ".pipeline")
capi = apsamd._capi
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
sizes = [(H, W, 3)] * len(imgs)
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": True}
def run():
    pano, _ = rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 32, opts, device_out=True)
    capi.check(apsamd.lib.aps_synchronize()); torch.cuda.synchronize()
    return pano
run(); run()
for prof in (False, True):
    capi.profile_enable(prof)
    for r in range(3):
        capi.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter(); p = run(); dt = (time.perf_counter() - t0) * 1e3
        print(f"prof={prof} wall {dt:.2f} ms", tuple(p.shape), flush=True)
    if prof:
        print(", ".join(f"{k}={v[0]:.3f}ms/{v[1]}" for k, v in capi.profile_all().items()))
