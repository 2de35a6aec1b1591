"""Timing of the bundle-adjustment pair blocks at the 64-view scene's scale (384 edges x ~3.7 k matches).
(The oracle needs 176 ms on one core for this batch.)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import apsamd
from importlib import import_module
from test_ba_gpu import _batch

ba = import_module(apsamd.__name__ + ".bundleAdjustment")
capi = apsamd._capi
rng = np.random.default_rng(13)
Ui, Uj, ptr, cams = _batch(rng, [int(v) for v in rng.integers(3000, 4500, 384)])
capi.profile_enable(True)
for it in range(3):
    capi.profile_reset()
    t0 = time.perf_counter()
    got = ba.ba_pair_blocks(Ui, Uj, ptr, cams, 2.0, True)
    dt = time.perf_counter() - t0
    prof = capi.profile_all()
print(f"{ptr[-1]} matches in 384 pairs: device kernel {prof['ba_pair_blocks'][0]:.3f} ms ({dt*1e3:.1f} ms with host staging of "
      f"{(Ui.nbytes + Uj.nbytes) / 1e6:.0f} MB); parity with the oracle: tests/test_ba_gpu.py::test_bench_scale_batch")
