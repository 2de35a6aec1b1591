"""The matching stage of the bench scene alone: features of the 64 x 4K views once, then match_pairs_csr over all 2016 pairs
four times - wall time per call, the library's APS_TRACE phases for the last call and its per-kernel HIP-event times.  Shows
what of the stage is not kernel time (host phases, read-backs)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
order = fm.pair_order(len(imgs))
for r in range(4):
    last = r == 3
    if last:
        os.environ["APS_TRACE"] = "1"
        capi.profile_enable(1)
        capi.profile_reset()
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
    print(f"call {r}: {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
prof = capi.profile_all()
print("kernels (HIP events): " + ", ".join(f"{k} {v[0]:.3f}" for k, v in prof.items() if v[0] > 0.005), "; sum %.2f ms" % sum(v[0] for v in prof.values()))
