"""Where the host time of renderPanorama goes (cProfile of three calls on the bench scene)."""
import cProfile, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
rp = import_module(apsamd.__name__ + ".renderPanorama")
pl = import_module(apsamd.__name__ + ".pipeline")
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
sizes = [(H, W, 3)] * len(imgs)
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": True}
def run():
    pano, _ = rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 32, opts, device_out=True)
    apsamd._capi.check(apsamd.lib.aps_synchronize()); torch.cuda.synchronize()
run(); run()
t0 = time.perf_counter(); run(); print("wall ms", (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); run(); run(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
