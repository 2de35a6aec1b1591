"""Ad-hoc throughput probe of the descriptor-distance kernel (not the bench)."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import apsamd
from importlib import import_module

fm = import_module(apsamd.__name__ + ".featureMatching")
lib = apsamd.lib
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kf = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
g = torch.Generator(device="cuda").manual_seed(0)
descs = []
for i in range(n_img):
    d = torch.rand(kf, 128, device="cuda", generator=g) ** 3
    d = d / d.norm(dim=1, keepdim=True)
    descs.append(d.contiguous())
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    pp, ii, jj, met = fm.match_pairwise_csr(descs, 0.6, 1.5, True)
    apsamd._capi.check(lib.aps_synchronize())
    dt = time.perf_counter() - t0
    npairs = n_img * (n_img - 1) // 2
    flops = 2 * 128 * npairs * kf * kf
    print(f"iter {it}: {dt*1e3:.1f} ms  pairs={npairs} kf={kf}  {flops/dt/1e12:.1f} TFLOP/s (f32 MFMA peak 157.3) matches={len(ii)}")
