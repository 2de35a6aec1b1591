"""Kernel-time probe of the split-precision candidate kernel (profile API), for A/B runs:
   python scripts/probe/probe_cand.py [n_img] [features]   (APS_MATCH_ABLATE=<bits> is honoured by builds made with
   make EXTRA=-DAPS_MATCH_TIMING only: 1 no selection, 2 no DMA, 4 screen never fires, 8 print phase timings, 32 no rescoring)"""
import sys

import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 12
kf = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
g = torch.Generator(device="cuda").manual_seed(0)
descs = []
for i in range(n_img):
    d = torch.rand(kf, 128, device="cuda", generator=g) ** 3
    d = d / d.norm(dim=1, keepdim=True)
    descs.append(d.contiguous())
torch.cuda.synchronize()
capi.profile_enable(True)
for it in range(3):
    capi.profile_reset()
    fm.match_pairwise_csr(descs, 0.6, 1.5, True)
    capi.check(apsamd.lib.aps_synchronize())
    p = capi.profile_all()
npairs = n_img * (n_img - 1) // 2
ms = p["match_cand_f16"][0]
fl = 2 * 128 * npairs * kf * kf
print(f"cand {ms:.2f} ms  ({ms/npairs*1e3:.1f} us/pair)  algorithmic {fl/ms/1e9:.1f} TF  pipe(9/8x) {1.125*fl/ms/1e9:.0f} TF;"
      f" fallback {p.get('match2nn_fallback',(0,0))[0]:.2f}")
