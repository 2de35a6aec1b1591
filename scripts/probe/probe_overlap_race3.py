"""Which property of a co-running workgroup disturbs SIFT?  (debug build: make EXTRA=-DAPS_DBG)
modes (bits): 1 LDS-DMA, 2 int8 MFMA 32x32x32 (VGPR accumulators), 4 LDS traffic, 8 f16 MFMA, 32 VALU, 64 int8 MFMA 16x16x64
(VGPR accumulators), 128 / 256 = 2 / 64 with the accumulators in AGPRs (csrc/dbg_agpr.hip)"""
import sys, ctypes
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
lib = apsamd._capi.lib
lib.aps_dbg_corun.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
W, H, f = 3840, 2160, 8000.0
imgs, cams = synth.make_scene(4, 4, W, H, f, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
sig = lambda out: [(int(d.shape[0]), float(d.double().sum())) for d, _ in out]
ref_sig = sig(pl.sift_many(inp, imgs))
for mode in [int(v) for v in sys.argv[1:]] or [0, 4, 1, 2]:
    res = []
    for rep in range(4):
        futs = pl.sift_submit(inp, imgs)
        k = 0
        while not all(fu.done() for fu in futs):
            apsamd._capi.check(lib.aps_dbg_corun(mode, 4096, 400))
            k += 1
        s = sig([fu.result() for fu in futs])
        res.append("same" if s == ref_sig else "DIFF" + str(sum(1 for i in range(len(s)) if s[i] != ref_sig[i])))
    print("mode", mode, res, "launches", k, flush=True)
