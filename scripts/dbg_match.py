import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import apsamd, oracle
from importlib import import_module
from util import planted_pair, bits
fm = import_module(apsamd.__name__ + ".featureMatching")
rng = np.random.default_rng(11)
a, b, ia, ib = planted_pair(rng, 1500, 1700, 600, noise=0.03, unit=False)
an, bn = oracle.normalize_rows(a), oracle.normalize_rows(b)
# 2nn on pre-normalised input (no device normalisation)
_, idx, d1, d2 = fm.nearest2SSDExhaustive(an, bn)
oi, o1, o2 = oracle.match_2nn_ssd(an, bn)
print("prenorm 2nn equal:", np.array_equal(idx, oi), np.array_equal(bits(d1), bits(o1)), np.array_equal(bits(d2), bits(o2)))
for uq in (True, False):
    m, met = fm.matchFeaturesScratch(a, b, MatchThreshold=1.5, MaxRatio=0.6, Unique=uq)
    om, omet = oracle.match_features(a, b, 0.6, 1.5, uq, 2)
    print("unique", uq, m.shape, om.shape)
    if m.shape == om.shape:
        bad = np.where((m != om).any(1))[0]
        print(" differing rows:", bad[:10], len(bad))
        for r in bad[:5]:
            print("  ", r, m[r], met[r], om[r], omet[r], bits(met[r:r+1]), bits(omet[r:r+1]))
    m2, met2 = fm.matchFeaturesScratch(an, bn, MatchThreshold=1.5, MaxRatio=0.6, Unique=uq)
    print(" prenorm path equal to oracle:", np.array_equal(m2, om), np.array_equal(bits(met2), bits(omet)))
