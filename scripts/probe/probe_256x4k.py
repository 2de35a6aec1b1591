"""BASELINE configs[3]'s image set on one GPU (256 4K views, one world): stage times of a cold and two warm passes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
views, cams = synth.make_scene(16, 16, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
Ks = [c["K"] for c in cams]
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pano, info = par.stitch_distributed(inp, dict(enumerate(views)), len(views), Ks, (2048, 2048), 0, None, pano_root=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {it}: {dt:.3f} s  pano {tuple(pano.shape)} verified {info['n_pairs_verified']}  " +
          ", ".join(f"{k}={v*1e3:.1f}" for k, v in info["times"].items()), flush=True)
    del pano, info
