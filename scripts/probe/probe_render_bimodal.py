"""Why is the render stage sometimes 42 ms instead of 25?  Per step: stage times, torch device-allocation count."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
W, H, f = 3840, 2160, 8000.0
cams = synth.grid_cameras(8, 8, W, H, f, 2 * np.arctan(W / (2 * f)) * 0.6, 2 * np.arctan(H / (2 * f)) * 0.6, 1.0, 12345)
local = {i: synth.render_view(cams[i], H, W, 12345, "cuda", finest_px=16.0) for i in range(64)}
torch.cuda.synchronize()
inp = pl.default_input(bands=5)
Ks = [c["K"] for c in cams]
for step in range(8):
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    f0 = torch.cuda.memory_stats().get("num_device_free", 0)
    t0 = time.perf_counter()
    pano, info = par.stitch_distributed(inp, local, 64, Ks, (2048, 2048), 0, None, pano_root=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.cuda.memory_stats()
    print(f"step {step}: {dt*1e3:.1f} ms render {info['times']['render']*1e3:.1f} features {info['times']['features']*1e3:.1f} "
          f"torch mallocs +{st.get('num_device_alloc', 0) - n0} frees +{st.get('num_device_free', 0) - f0} reserved {st['reserved_bytes.all.current'] / 2**30:.1f} GiB", flush=True)
    del pano
