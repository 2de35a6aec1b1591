"""Serial timing probe of the SIFT stage on one synthetic 4K view with the per-kernel breakdown."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
fm = import_module(apsamd.__name__ + ".featureMatching")
synth = import_module(apsamd.__name__ + ".synth")
capi = apsamd._capi
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(2, 1, W, H, f, device="cuda", finest_px=16.0)
inp = {"detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6}
fm.sift_extract(inp, imgs[0], device_out=True)
capi.profile_enable(True)
for it in range(2):
    capi.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d, p = fm.sift_extract(inp, imgs[0], device_out=True)
    capi.check(capi.lib.aps_synchronize()); dt = time.perf_counter() - t0
prof = capi.profile_all()
print(f"{dt*1e3:.2f} ms wall, {len(p)} features; " + ", ".join(f"{k}={v[0]:.3f}ms/{v[1]}" for k, v in prof.items()))
