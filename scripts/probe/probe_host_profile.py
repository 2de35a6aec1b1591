"""Host-side profile (cProfile) of one resident stitch of the bench scene: where the Python / ctypes time of a step goes."""
import sys, os, time, cProfile, pstats, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import apsamd
from importlib import import_module
synth = import_module(apsamd.__name__ + ".synth"); pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")
W, H, FOCAL, OVERLAP = 3840, 2160, 8000.0, 0.40
nx = ny = 8; n = 64
cams = synth.grid_cameras(nx, ny, W, H, FOCAL, 2*np.arctan(W/(2*FOCAL))*(1-OVERLAP), 2*np.arctan(H/(2*FOCAL))*(1-OVERLAP), 1.0, 12345)
imgs = {i: synth.render_view(cams[i], H, W, 12345, "cuda", finest_px=16.0) for i in range(n)}
torch.cuda.synchronize()
if os.environ.get("APS_PARALLEL_FORCE_COLLECTIVES") == "1":  # the N > 1 driver with one rank over RCCL (parallel._multi)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29561")
    os.dup2(2, 1)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
inp = pl.default_input(bands=5)
Ks = [c["K"] for c in cams]
for _ in range(3):
    pano, info = par.stitch_distributed(inp, imgs, n, Ks, (2048, 2048), 0, None, pano_root=0)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
pano, info = par.stitch_distributed(inp, imgs, n, Ks, (2048, 2048), 0, None, pano_root=0)
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumtime").print_stats(45)
print(info.get("times"))
