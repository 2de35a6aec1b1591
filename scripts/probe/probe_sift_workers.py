"""SIFT stage alone on the bench scene for several worker counts: is it GPU-bound or launch-bound?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(8, 8, W, H, f, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
for workers in [int(v) for v in sys.argv[1:]] or [8]:
    pl._SIFT_POOL = None
    for rep in range(3):
        t0 = time.perf_counter()
        out = pl.sift_many(inp, imgs, workers=workers)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"workers {workers}: {dt*1e3:.1f} ms, features/view {np.mean([len(p) for _, p in out]):.0f}", flush=True)
