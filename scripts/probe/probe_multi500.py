"""BASELINE configs[4] image count on one GPU (500 mixed 2K views in 25 worlds): stage times and library brackets of a
warm run.  usage: probe_multi500.py [n_worlds]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
capi = apsamd._capi
W, H, f, nx, ny = 2048, 1536, 2400.0, 5, 4
n_worlds = int(sys.argv[1]) if len(sys.argv) > 1 else 25
views, Ks = [], []
for wi in range(n_worlds):
    imgs, cams = synth.make_scene(nx, ny, W, H, f, 0.4, seed=1000 + 17 * wi, device="cuda", finest_px=10.0)
    views += imgs; Ks += [c["K"] for c in cams]
perm = np.random.default_rng(9).permutation(len(views))
views, Ks = [views[k] for k in perm], [Ks[k] for k in perm]
torch.cuda.synchronize()
n = len(views)
inp = pl.default_input(bands=5)
for it in range(3):
    if it == 2:
        capi.profile_enable(True); capi.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, Ks, (2048, 2048), 0, None, pano_root=0)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"run {it}: {dt:.3f} s  comps {info['n_components']} feats/view {np.mean(info['n_features']):.0f} verified {info['n_pairs_verified']}  " +
          ", ".join(f"{k}={v*1e3:.1f}" for k, v in info["times"].items()), flush=True)
    del pano, info
print(", ".join(f"{k}={v[0]:.2f}ms/{v[1]}" for k, v in capi.profile_all().items()))
