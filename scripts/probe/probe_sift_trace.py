"""One 4K view through SIFT a few times (single stream): under rocprofv3 --kernel-trace the per-dispatch durations show
what each octave's blur / extrema launch costs stand-alone."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
fm = import_module(apsamd.__name__ + ".featureMatching")
synth = import_module(apsamd.__name__ + ".synth")
import os
H = int(os.environ.get("APS_TRACE_ROWS", "2160"))  # (round 6: a band of a view - do planes that fit the Infinity Cache blur faster?)
imgs, cams = synth.make_scene(1, 1, 3840, 2160, 8000.0, device="cuda", finest_px=16.0)
imgs = [imgs[0][:H].contiguous()]
inp = {"detector": "SIFT"}
for _ in range(3):
    d, p = fm.sift_extract(inp, imgs[0], device_out=True)
torch.cuda.synchronize()
print(len(p))
