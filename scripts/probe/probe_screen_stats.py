"""Bench-scene matching (64 x 4K views, 2016 pairs): rows, int8-screen survivors and final matches, split by whether the
pair overlaps (has matches at all).  Shows how much of the f16 row-list pass is spent on rows that end up matched."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
fm = import_module(apsamd.__name__ + ".featureMatching")
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
imgs, _ = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
inp = pl.default_input(bands=5)
descs = [d for d, _ in pl.sift_many(inp, imgs)]
del imgs
order = fm.pair_order(len(descs))
pp, ia, ib, met = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
rows, surv = ctypes.c_int64(0), ctypes.c_int64(0)
apsamd._capi.check(apsamd.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
cnt = np.diff(pp)
nrows = np.array([descs[i].shape[0] for i, _ in order])
print(f"rows {rows.value}  survivors {surv.value} ({surv.value / rows.value:.4f})  matches after unique {int(pp[-1])} ({int(pp[-1]) / rows.value:.4f})")
print(f"pairs with >= 50 matches: {(cnt >= 50).sum()}, their rows {nrows[cnt >= 50].sum()} their matches {cnt[cnt >= 50].sum()}")
pp2, ia2, ib2, met2 = fm.match_pairs_csr(descs, order, 0.6, 1.5, False, device_out=True)
print(f"matches before unique {int(pp2[-1])} ({int(pp2[-1]) / rows.value:.4f})")
