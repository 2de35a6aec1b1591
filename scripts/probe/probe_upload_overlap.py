"""End-to-end input path: how long do the 64 uploads take alone, and what do they cost the SIFT stage they overlap with?"""
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
host = [torch.empty(i.shape, dtype=torch.uint8, pin_memory=True).copy_(i) for i in imgs]
cs = torch.cuda.Stream()
def upload():
    up, evs = [], []
    with torch.cuda.stream(cs):
        for h in host:
            up.append(h.to("cuda", non_blocking=True))
            e = torch.cuda.Event(); e.record(cs); evs.append(e)
    return up, evs
def t(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0), r
for rep in range(3):
    a, _ = t(lambda: pl.sift_many(inp, imgs))
    b, _ = t(upload)
    def both():
        up, evs = upload()
        return pl.sift_many(inp, up, ready=evs)
    c, _ = t(both)
    def serial():
        up, evs = upload(); torch.cuda.synchronize()
        return pl.sift_many(inp, up)
    d, _ = t(serial)
    print(f"SIFT resident {a:.1f} ms | 64 uploads alone {b:.1f} ms ({1.59e3 / b:.1f} GB/s) | overlapped {c:.1f} ms | uploads then SIFT {d:.1f} ms", flush=True)
