"""Would the render of step k fit beside the feature extraction of step k + 1?  (A pipelined loop could hide it there: the render's
kernels are latency / vector-issue bound, the extraction's bandwidth bound, and neither touches the int8 matrix pipe.)
Measures, on the bench scene: the extraction alone, the render alone, and both started together from two host threads."""
import sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
rp = import_module(apsamd.__name__ + ".renderPanorama")
capi = apsamd._capi
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
sizes = [(H, W, 3)] * 64
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}


def sync():
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()


def sift():
    out = [fu.result() for fu in pl.sift_submit(inp, imgs, points_device=True)]
    capi.check(capi.lib.aps_synchronize())
    return out


def render(res):
    t0 = time.perf_counter()
    rp.renderPanorama(inp, imgs, sizes, cams, "spherical", 27, opts, device_out=True)
    capi.check(capi.lib.aps_synchronize())
    res["render"] = 1e3 * (time.perf_counter() - t0)


for _ in range(2):
    sift(); render({}); sync()
for rep in range(4):
    sync(); t0 = time.perf_counter(); sift(); sync(); ts = 1e3 * (time.perf_counter() - t0)
    r = {}; sync(); render(r); sync()
    sync(); t0 = time.perf_counter()
    r2 = {}
    th = threading.Thread(target=render, args=(r2,)); th.start()
    sift(); th.join(); sync()
    tb = 1e3 * (time.perf_counter() - t0)
    print(f"extraction alone {ts:.1f} ms, render alone {r['render']:.1f} ms, both together {tb:.1f} ms (render inside: {r2['render']:.1f} ms); serial sum {ts + r['render']:.1f}", flush=True)
