"""Round 6: how much of the extrema sweep could be skipped?  Per octave and DoG layer of one bench view: the share of 64 x 32 tiles (the blur's
tile) whose max |DoG| exceeds the sweep's threshold thr = floor(0.5 * contrast / nl * 255) - a tile at or below it cannot hold a candidate of
that layer.  Approximate pyramid (torch, f32 separable Gaussians; not the library's bits) - a statistic, not a parity check."""
import sys, math
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
nl, sigma0, contrast = 4, 1.6, 0.04
for key in ("NumLayersInOctave", "n_layers"):
    if key in inp: nl = int(inp[key])
print("input keys:", {k: v for k, v in inp.items() if "ayer" in k or "igma" in k or "ontrast" in k})
thr = math.floor(0.5 * contrast / nl * 255.0)
def blur(x, s):
    r = max(1, int(round(s * 4)))
    k = torch.exp(-0.5 * (torch.arange(-r, r + 1, device=x.device, dtype=torch.float32) / s) ** 2); k /= k.sum()
    x = F.conv2d(F.pad(x, (r, r, 0, 0), mode="reflect"), k.view(1, 1, 1, -1))
    return F.conv2d(F.pad(x, (0, 0, r, r), mode="reflect"), k.view(1, 1, -1, 1))
for vi in (0, 27, 63):
    im = imgs[vi].float()
    g = (0.298936021293775 * im[..., 0] + 0.587043074451121 * im[..., 1] + 0.114020904255103 * im[..., 2] + 0.5).floor()[None, None]
    g = F.interpolate(g, scale_factor=2, mode="bilinear", align_corners=False)
    base = blur(g, math.sqrt(max(sigma0 ** 2 - 1.0, 0.01)))
    kf = 2.0 ** (1.0 / nl)
    tot_px = tot_live = 0
    for o in range(6):
        G = [base]
        for i in range(1, nl + 3):
            sp = sigma0 * kf ** (i - 1)
            G.append(blur(G[-1], math.sqrt((sp * kf) ** 2 - sp ** 2)))
        D = [G[i + 1] - G[i] for i in range(nl + 2)]
        h, w = D[0].shape[-2:]
        shares = []
        live_any = None
        for d in D:
            t = F.max_pool2d(d.abs(), kernel_size=(32, 64), ceil_mode=True) > thr
            shares.append(float(t.float().mean()))
        # a strip region is needed when any of the nl centre layers is live there (its neighbours are then read too)
        centre = torch.stack([F.max_pool2d(D[s].abs(), kernel_size=(32, 64), ceil_mode=True) > thr for s in range(1, nl + 1)]).any(0)
        # grow by one tile (the 3 x 3 window reaches into the neighbouring tile)
        grown = F.max_pool2d(centre.float(), 3, 1, 1) > 0
        print(f"view {vi} octave {o} ({w} x {h}): tiles with max|DoG| > {thr} per layer " + " ".join(f"{s:.2f}" for s in shares) +
              f" | tiles needed by the sweep {float(centre.float().mean()):.2f} (grown by one tile {float(grown.float().mean()):.2f})")
        tot_px += h * w; tot_live += h * w * float(grown.float().mean())
        base = G[nl][..., ::2, ::2]
    print(f"view {vi}: pixel-weighted share of the sweep that is needed: {tot_live / tot_px:.2f}")
