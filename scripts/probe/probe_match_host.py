"""Round 6: where the matching stage's wall time goes outside its kernels (Python wrapper, C entry, read-backs): the call
timed from Python against the library's own APS_TRACE phases and a cProfile of the wrapper."""
import os, sys, time, cProfile, pstats
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
order = fm.pair_order_array(len(imgs))
for r in range(3):
    fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
os.environ["APS_TRACE"] = "1"
pr = cProfile.Profile()
for r in range(3):
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    out = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    pr.disable()
    t1 = time.perf_counter()
    capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
    print(f"call {r}: returns after {(t1 - t0) * 1e3:.2f} ms, synced {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
