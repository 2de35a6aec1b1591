"""Run by tests/test_nccl_path_gpu.py in a child process: the N > 1 code path of parallel.stitch_distributed on the REAL
backend ("nccl" = RCCL) with a one-rank process group and APS_PARALLEL_FORCE_COLLECTIVES=1 (RCCL refuses two ranks on one
GPU).  Every collective, asynchronous handle, stream hand-over and tile / panorama gather of the multi-rank driver runs;
the results must equal the plain single-process run byte for byte.  Prints OK <summary> on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import apsamd  # noqa: E402
from importlib import import_module  # noqa: E402
from test_components_gpu import _worlds, W, H  # noqa: E402

synth = import_module(apsamd.__name__ + ".synth")
pl = import_module(apsamd.__name__ + ".pipeline")
par = import_module(apsamd.__name__ + ".parallel")


def run_all(tag):
    out = {}
    views, cams, _ = _worlds(synth)
    torch.cuda.synchronize()
    n = len(views)
    Ks = [c["K"] for c in cams]
    for name, extra in (("pairwise", {}), ("global", {"matchFeaturesPairwise": 0, "k": 4}),
                        ("second_pass", {"resizeImage": 1, "resizeImagePanoramaCluster": 1})):
        inp = pl.default_input(bands=3)
        inp.update(extra)
        pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, Ks, (512, 512), 0, None, pano_root=0)
        torch.cuda.synchronize()
        out[name] = ([p.cpu().numpy() for p in info["panoramas"]], info["n_components"], list(info["n_features"]),
                     info["n_pairs_verified"])
    # one big component of 3 x 3 views with ground-truth cameras: tiles sharded, gathered to the root
    imgs, gcams = synth.make_scene(3, 3, W, H, 900.0, overlap=0.45, seed=9, device="cuda", finest_px=6.0)
    torch.cuda.synchronize()
    pano, info = par.stitch_distributed(pl.default_input(bands=3), dict(enumerate(imgs)), 9, [c["K"] for c in gcams], (256, 256),
                                        0, gcams, pano_root=0)
    torch.cuda.synchronize()
    out["one_component"] = ([pano.cpu().numpy()], info["n_components"], list(info["n_features"]), info["n_pairs_verified"])
    return out


def main():
    plain = run_all("plain")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29541"),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", APS_PARALLEL_FORCE_COLLECTIVES="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert par._multi(1)
    forced = run_all("forced")
    dist.barrier()
    dist.destroy_process_group()
    for name in plain:
        pa, fa = plain[name], forced[name]
        assert pa[1:] == fa[1:], (name, pa[1:], fa[1:])
        assert len(pa[0]) == len(fa[0]) and len(pa[0]) >= 1, name
        for x, y in zip(pa[0], fa[0]):
            assert x.shape == y.shape and np.array_equal(x, y), (name, x.shape, y.shape)
    print("OK", {k: (len(v[0]), v[1], v[3]) for k, v in forced.items()})


if __name__ == "__main__":
    try:
        main()
    except BaseException:  # the parent shows the child's output only in part: put the reason where it is seen
        import traceback

        print("RUNNER FAILED\n" + traceback.format_exc()[-3000:], flush=True)
        raise
