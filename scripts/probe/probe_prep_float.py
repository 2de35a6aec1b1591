"""Round 6: what the exact-code search of q8_desc_rows costs on data that has NO exact code (ordinary float descriptors): the
preparation of 64 sets of 20 k rows, SIFT-like integers over their norm against the same rows with a 1e-3 perturbation."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import apsamd
from importlib import import_module
from util import sift_like
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
rng = np.random.default_rng(3)
ints = [sift_like(rng, 20000, unit=False) for _ in range(16)]
unit = [(a / np.sqrt((a * a).sum(1, dtype=np.float32))[:, None].clip(1e-30)).astype(np.float32) for a in ints]
flt = [u * (1.0 + 1e-3 * rng.standard_normal(u.shape).astype(np.float32)) for u in unit]
for name, sets in (("integers over their norm", unit), ("perturbed floats", flt)):
    dev = [torch.from_numpy(s).cuda() for s in sets]
    order = fm.pair_order_array(len(dev))
    for r in range(3):
        capi.profile_enable(1); capi.profile_reset()
        fm.match_pairs_csr(dev, order, 0.6, 1.5, True, device_out=True)
        capi.check(capi.lib.aps_synchronize())
        prof = {k: round(v[0], 3) for k, v in capi.profile_all().items() if v[0] > 0.005}
    import ctypes
    j, e = ctypes.c_int64(), ctypes.c_int64(); capi.lib.aps_match_screen_exact_jobs(ctypes.byref(j), ctypes.byref(e))
    print(f"{name}: exact jobs {e.value} of {j.value}; {prof}")
