"""Round 6 A/B on one box: the survivors' pass as the exact int8 list kernel + rescoring (default) against the f16 candidate
kernel (APS_MATCH_NO_LIST_I8=1), same descriptors, same process; kernel times by HIP events, candidate statistics."""
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
def run(tag, reps=3):
    best = None
    for r in range(reps):
        capi.profile_enable(1); capi.profile_reset()
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True, device_out=True)
        capi.check(capi.lib.aps_synchronize()); torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        prof = {k: round(v[0], 3) for k, v in capi.profile_all().items() if v[0] > 0.005}
        best = (wall, prof) if best is None or wall < best[0] else best
    print(f"{tag:>10}: wall {best[0]:.2f} ms  {best[1]}  matches {int(out[0][-1])}", flush=True)
    return out
for k in range(2):
    a = run("list i8")
    os.environ["APS_MATCH_NO_LIST_I8"] = "1"
    b = run("cand f16")
    del os.environ["APS_MATCH_NO_LIST_I8"]
assert np.array_equal(a[0], b[0]) and bool(torch.equal(a[1], b[1])) and bool(torch.equal(a[2], b[2])) and bool(torch.equal(a[3].view(torch.int32), b[3].view(torch.int32)))
print("lists identical")
