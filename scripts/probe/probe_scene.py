import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
fm = import_module(apsamd.__name__ + ".featureMatching")
synth = import_module(apsamd.__name__ + ".synth")
inp = {"detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6}
W, H, f = 3840, 2160, 8000.0
for fp in [float(a) for a in sys.argv[1:]]:
    imgs, cams = synth.make_scene(2, 1, W, H, f, device="cuda", finest_px=fp)
    d, p = fm.sift_extract(inp, imgs[0], device_out=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d, p, aux = fm.sift_extract(inp, imgs[0], device_out=True, want_aux=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    d2, p2 = fm.sift_extract(inp, imgs[1], device_out=True)
    m, met = fm.matchFeaturesScratch(d, d2, MatchThreshold=1.5, MaxRatio=0.6)
    octs = (aux[:, 3].astype(int) & 255)
    print(f"finest_px={fp}: {len(p)} features ({dt*1e3:.1f} ms), per octave {np.bincount(octs).tolist()}, matches {len(m)}, std {imgs[0].float().std().item():.1f}")
