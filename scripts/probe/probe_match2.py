"""Throughput probe of the match stage on SIFT-like planted descriptors with the per-kernel breakdown."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import apsamd
from importlib import import_module
from util import sift_like
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kf = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(0)
base = sift_like(rng, kf * 2)
descs = []
for i in range(n_img):
    keep = rng.permutation(kf * 2)[:kf]
    d = base[keep] + 0.02 * rng.standard_normal((kf, 128)).astype(np.float32)
    d = np.maximum(d, 0); d /= np.linalg.norm(d, axis=1, keepdims=True)
    descs.append(torch.from_numpy(d.astype(np.float32)).cuda())
fm.match_pairwise_csr(descs, 0.6, 1.5, True)
capi.profile_enable(True)
for it in range(2):
    capi.profile_reset()
    t0 = time.perf_counter()
    pp, ii, jj, met = fm.match_pairwise_csr(descs, 0.6, 1.5, True)
    capi.check(capi.lib.aps_synchronize())
    dt = time.perf_counter() - t0
npairs = n_img * (n_img - 1) // 2
flops = 2 * 128 * npairs * kf * kf
prof = capi.profile_all()
print(f"total {dt*1e3:.1f} ms, matches={len(ii)}; " + ", ".join(f"{k}={v[0]:.2f}ms" for k, v in prof.items()))
k = "match_cand_f16" if "match_cand_f16" in prof else "match2nn"
print(f"{k}: {flops/prof[k][0]/1e9:.1f} TFLOP/s algorithmic")
