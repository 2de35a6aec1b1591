"""Full-scale check of the render culls (footprints, block-level image cull, fused levels): the 64 x 4K bench scene with
ground-truth cameras is rendered with every cull on and with APS_RENDER_NO_CULL=1 / APS_RENDER_NO_FUSE=1; the two
panoramas must be identical in every byte."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

import bench

synth = import_module(apsamd.__name__ + ".synth")
rp = import_module(apsamd.__name__ + ".renderPanorama")
W, H, f = bench.W, bench.H, bench.FOCAL
cams = synth.grid_cameras(8, 8, W, H, f, 2 * np.arctan(W / (2 * f)) * (1 - bench.OVERLAP), 2 * np.arctan(H / (2 * f)) * (1 - bench.OVERLAP), 1.0, 12345)
imgs = [synth.render_view(c, H, W, 12345, "cuda", finest_px=bench.FINEST_PX) for c in cams]
torch.cuda.synchronize()
opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}
sizes = [(H, W, 3)] * 64
outs = []
for env in ({}, {"APS_RENDER_NO_CULL": "1", "APS_RENDER_NO_FUSE": "1"}):
    for k in ("APS_RENDER_NO_CULL", "APS_RENDER_NO_FUSE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t0 = time.perf_counter()
    pano, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 27, opts, device_out=True)
    apsamd._capi.check(apsamd.lib.aps_synchronize())
    print(env or "all culls on", f"{(time.perf_counter() - t0) * 1e3:.1f} ms", tuple(pano.shape), int(pano.sum()))
    outs.append(pano)
same = torch.equal(outs[0], outs[1])
print("identical:", same)
sys.exit(0 if same else 1)
