"""Render stage alone on the bench scene (64 x 4K views, ground-truth cameras): wall time per render and, under
rocprofv3 --kernel-trace --stats, the per-kernel breakdown.  usage: probe_render.py [reps] [NXxNY] [blending] [first/step of the tiles]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
rp = import_module(apsamd.__name__ + ".renderPanorama")
pl = import_module(apsamd.__name__ + ".pipeline")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nx, ny = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8x8").split("x"))
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(nx, ny, W, H, f, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
sizes = [(H, W, 3)] * len(imgs)
blending = sys.argv[3] if len(sys.argv) > 3 else "multiband"
subset = tuple(int(v) for v in sys.argv[4].split("/")) if len(sys.argv) > 4 else None
opts = {"anglePower": 2, "blending": blending, "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}
for r in range(reps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pano, _ = rp.renderPanorama(inp, imgs, sizes, cams, "spherical", len(imgs) // 2, opts, device_out=True, tile_subset=subset)
    apsamd._capi.check(apsamd.lib.aps_synchronize()); torch.cuda.synchronize()
    print(f"render {r} ({blending}): {(time.perf_counter()-t0)*1e3:.1f} ms, pano {tuple(pano.shape)}, mean {pano.float().mean().item():.3f}", flush=True)
import zlib
print("crc32 of the panorama bytes:", hex(zlib.crc32(pano.cpu().numpy().tobytes())), flush=True)
