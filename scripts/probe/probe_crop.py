"""Timing of the crop rectangle on a bench-sized canvas (21123 x 11632).  (The same shape at 60 MPix is checked against
the oracle in tests/test_fullsize_gpu.py; the oracle needs 1.16 s on one core for this canvas.)"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

ip = import_module(apsamd.__name__ + ".imageProcessing")
capi = apsamd._capi
H, W = 11632, 21123
yy = torch.arange(H, device="cuda").view(-1, 1).float()
xx = torch.arange(W, device="cuda").view(1, -1).float()
# a panorama-like outline: a wide ellipse with a wavy rim, a few enclosed holes, a bay at the top
rim = 0.46 + 0.02 * torch.sin(xx / 900.0) + 0.015 * torch.cos(yy / 700.0)
m = ((yy - H / 2) / (H * rim)) ** 2 + ((xx - W / 2) / (W * 0.49)) ** 2 < 1
for cy, cx, r in [(3000, 5000, 150), (8000, 15000, 90), (6000, 10000, 40)]:
    m &= ~(((yy - cy) ** 2 + (xx - cx) ** 2) < r * r)
m &= ~((yy < 2500) & ((xx - 12000).abs() < 300))
img = (m.unsqueeze(-1) * torch.tensor([180, 150, 120], device="cuda")).to(torch.uint8).contiguous()
torch.cuda.synchronize()
capi.profile_enable(True)
for it in range(3):
    capi.profile_reset()
    t0 = time.perf_counter()
    rect, ok = ip.cropRectangle(img)
    dt = time.perf_counter() - t0
    prof = capi.profile_all()
print(f"device: rect {rect} valid {ok}; {dt*1e3:.1f} ms wall; " + ", ".join(f"{k}={v[0]:.2f}ms" for k, v in prof.items() if k.startswith("crop")))
