"""Gap statistics of exact 2-NN distances on bench-like SIFT descriptors: how often is d(K+1) - d(2) < eps?
Decides how many candidates a lower-precision screening product needs.  python scripts/probe/probe_gaps.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
NX, NY, W, H, F, OV = 8, 8, 3840, 2160, 8000.0, 0.4
cams = synth.grid_cameras(NX, NY, W, H, F, 2 * np.arctan(W / (2 * F)) * (1 - OV), 2 * np.arctan(H / (2 * F)) * (1 - OV), 1.0, 12345)
ids = [0, 1, 9, 40]
imgs = [synth.render_view(cams[i], H, W, 12345, "cuda", finest_px=16.0) for i in ids]
input_ = pl.default_input(bands=5)
torch.cuda.synchronize()
descs, _ = pl.extract_features(input_, imgs)
descs = [torch.as_tensor(d, device="cuda").float() for d in descs]
torch.cuda.synchronize()
descs = [d / (d.norm(dim=1, keepdim=True) + 1.1920929e-07) for d in descs]
for d in descs:
    h = d.half().float()
    dn = (d - h).norm(dim=1)
    b = d.bfloat16().float()
    print("n", d.shape[0], "f16 delta-norm mean %.3e max %.3e | bf16 %.3e max %.3e | min nonzero %.3e" % (
        dn.mean(), dn.max(), (d - b).norm(dim=1).mean(), (d - b).norm(dim=1).max(), d[d > 0].min()))
for (a, b) in [(0, 1), (0, 2), (0, 3), (1, 2)]:
    A, B = descs[a], descs[b]
    D = (A * A).sum(1, keepdim=True) + (B * B).sum(1)[None] - 2 * A @ B.T
    v, _ = torch.topk(D, 10, dim=1, largest=False)
    v = v.double()
    line = f"pair {ids[a]}-{ids[b]} d1 med {v[:,0].median():.3f} d2 med {v[:,1].median():.3f}:"
    for eps in (2.4e-4, 6e-4, 1.0e-3, 1.4e-3, 2.8e-3, 5.6e-3, 1.6e-2):
        fr = [float(((v[:, K] - v[:, 1]) < eps).double().mean()) for K in (3, 4, 5, 6, 8)]
        line += "\n   eps %.1e  K=3,4,5,6,8 fallback frac: " % eps + " ".join("%.4f" % x for x in fr)
    print(line)
