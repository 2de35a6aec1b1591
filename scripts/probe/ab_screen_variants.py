"""A/B of experiment switches of the screening kernel (APS_SCR_VARIANT) on the bench scene's 2016 pairs, same process."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
imgs, _ = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
del imgs
order = fm.pair_order(len(descs))
variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 4, 3, 10, 0]
ref = None
for v in variants:
    os.environ["APS_SCR_VARIANT"] = str(v)
    ts = []
    for rep in range(3):
        capi.profile_enable(2); capi.profile_reset()
        out = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
        capi.check(capi.lib.aps_synchronize())
        ts.append(capi.profile_all()["match_screen_i8"][0]); capi.profile_enable(False)
    n = int(out[0][-1])
    ref = n if ref is None else ref
    print(f"variant {v:2d}: screen {min(ts):.2f} ms (median {sorted(ts)[1]:.2f}), matches {n} {'ok' if n == ref else 'DIFFERENT'}", flush=True)
