"""The feature-extraction stage of the bench alone (64 x 4K views, the resident form: sift_submit with the keypoints left on the
device, 10 worker streams), repeated: median / min over the repetitions - the A/B harness for SIFT kernel changes (the bench's
own stage time moves by +-1.5 ms from box to box)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ts = []
for rep in range(reps + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = [f.result() for f in pl.sift_submit(inp, imgs, points_device=True)]
    apsamd._capi.check(apsamd.lib.aps_synchronize())
    torch.cuda.synchronize()
    if rep >= 2:
        ts.append(1e3 * (time.perf_counter() - t0))
    del out
ts = np.array(ts)
print(f"features stage: median {np.median(ts):.2f} ms  min {ts.min():.2f}  max {ts.max():.2f}  ({reps} repetitions)")
