"""Round 6: which jobs of small matching calls run on exact int8 codes (aps_match_screen_exact_jobs): integer descriptors matched in
normalised form, the same integers over their f32 norm, and sub-sets of both."""
import sys, ctypes, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import apsamd
from importlib import import_module
from util import planted_pair
fm = import_module(apsamd.__name__ + ".featureMatching")
capi = apsamd._capi
def ej():
    j, e = ctypes.c_int64(), ctypes.c_int64(); capi.lib.aps_match_screen_exact_jobs(ctypes.byref(j), ctypes.byref(e)); return j.value, e.value
rng = np.random.default_rng(61)
a, b, _, ib = planted_pair(rng, 3000, 3500, 1200, noise=0.03, unit=False)
m, _ = fm.matchFeaturesScratch(a, b, MatchThreshold=3.5, MaxRatio=0.6, Unique=True); print("int sets:", len(m), ej())
au = (a / np.sqrt((a * a).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
bu = (b / np.sqrt((b * b).sum(1, dtype=np.float32))[:, None]).astype(np.float32)
m, _ = fm.matchFeaturesScratch(au, bu, MatchThreshold=3.5, MaxRatio=0.6, Unique=True); print("unit sets:", len(m), ej())
for n in (64, 100, 1000):
    m, _ = fm.matchFeaturesScratch(a[:n], b[:n], MatchThreshold=3.5, MaxRatio=0.6, Unique=True); print("int sets n", n, len(m), ej())
    m, _ = fm.matchFeaturesScratch(au[:n], bu[:n], MatchThreshold=3.5, MaxRatio=0.6, Unique=True); print("unit sets n", n, len(m), ej())
