import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
rp = import_module(apsamd.__name__ + ".renderPanorama")
capi = apsamd._capi
W, H, f, world = 3840, 2160, 8000.0, 8
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
sizes = [(H, W, 3)] * 64
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}
def sync():
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
full_, _, _, geo = rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 27, opts, device_out=True, return_covered=True)
ranges = par.tile_ranges(int(geo["H"]), int(geo["W"]), (2048, 2048), world)
sub = ("range",) + ranges[3]
for _ in range(3):
    rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 27, opts, device_out=True, tile_subset=sub, geo=geo); sync()
capi.profile_enable(1); capi.profile_reset()
import cProfile, pstats
pr = cProfile.Profile()
ts = []
for _ in range(5):
    sync(); t0 = time.perf_counter()
    pr.enable()
    rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 27, opts, device_out=True, tile_subset=sub, geo=geo)
    pr.disable()
    t1 = time.perf_counter()
    sync(); ts.append((t1 - t0, time.perf_counter() - t0))
print("call returns / synced (ms):", [(round(1e3*a,2), round(1e3*b,2)) for a, b in ts])
print({k: round(v[0] / 5, 3) for k, v in capi.profile_all().items() if v[0] > 0.01})
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
