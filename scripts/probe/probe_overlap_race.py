"""Does a matching call on the main thread disturb SIFT running on the worker threads (and vice versa)?
Reference: SIFT of 16 views with nothing else running, the match lists of the first 12 views with nothing else running.
Then both at once, several times; any difference is reported."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
fm = import_module(apsamd.__name__ + ".featureMatching")
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(4, 4, W, H, f, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
ref = pl.sift_many(inp, imgs)
torch.cuda.synchronize()
sig = lambda out: [(int(d.shape[0]), float(d.double().sum())) for d, _ in out]
ref_sig = sig(ref)
descs = [d for d, _ in ref[:12]]
order = fm.pair_order(12)
mref = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
msig = lambda m: (int(m[0][-1]), int(m[1].long().sum()), int(m[2].long().sum()))
mref_sig = msig(mref)
print("reference:", sum(s[0] for s in ref_sig), mref_sig, flush=True)
bad = 0
apsamd._capi.profile_enable(1)
apsamd._capi.profile_reset()
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    futs = pl.sift_submit(inp, imgs)
    ms = []
    while not all(fu.done() for fu in futs):
        try:
            ms.append(msig(fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)))
        except Exception as e:  # ablated kernels hand garbage to the filter
            ms.append(("exc", str(e)[:40]))
    out = [fu.result() for fu in futs]
    torch.cuda.synchronize()
    s = sig(out)
    for i in range(len(s)):
        if s[i] != ref_sig[i]:
            pa = {tuple(np.round(r, 4)) for r in ref[i][1]}
            pb = {tuple(np.round(r, 4)) for r in out[i][1]}
            da, db = ref[i][0].cpu().numpy(), out[i][0].cpu().numpy()
            nd = -1
            if da.shape == db.shape:
                rows = np.flatnonzero((da != db).any(axis=1))
                nd = len(rows)
                ex = [(int(r), ref[i][1][r].round(2).tolist(), float(np.abs(da[r] - db[r]).max())) for r in rows[:3]]
            else:
                ex = []
            print(f"   image {i}: {len(pa)} -> {len(pb)} keypoints, only-ref {sorted(pa - pb)[:4]}, only-run {sorted(pb - pa)[:4]}, "
                  f"descriptor rows differing {nd} {ex}", flush=True)
            break
    sift_ok = s == ref_sig
    match_ok = all(m == mref_sig for m in ms)
    bad += (not sift_ok) + (not match_ok)
    print(f"rep {rep}: sift {'same' if sift_ok else 'DIFFERENT ' + str([i for i in range(len(s)) if s[i] != ref_sig[i]])}, "
          f"{len(ms)} match calls {'same' if match_ok else 'DIFFERENT ' + str([m for m in ms if m != mref_sig][:2])}", flush=True)
prof = apsamd._capi.profile_all()
print({k: (round(v[0], 2), v[1]) for k, v in prof.items() if k.startswith("match")})
print("RESULT", "clean" if bad == 0 else f"{bad} disturbed runs")
