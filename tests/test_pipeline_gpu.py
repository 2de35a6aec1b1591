"""GPU end to end on a small synthetic scene: the sharded driver (parallel.stitch_distributed) on one rank, with the
match lists resident on the device and with the multi-rank host-list code path, must agree in every verified pair,
every model bit and every panorama byte; and the panorama must show the scene (all views verified, canvas covered)."""
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(gpu, monkeypatch, host_lists):
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    if host_lists:
        monkeypatch.setenv("APS_PARALLEL_HOST_LISTS", "1")
    else:
        monkeypatch.delenv("APS_PARALLEL_HOST_LISTS", raising=False)
    nx, ny, w, h, f = 3, 2, 640, 480, 900.0
    cams = synth.grid_cameras(nx, ny, w, h, f, 2 * np.arctan(w / (2 * f)) * 0.6, 2 * np.arctan(h / (2 * f)) * 0.6, 1.0, 7)
    views = {i: synth.render_view(cams[i], h, w, 7, "cuda", finest_px=6.0) for i in range(nx * ny)}
    torch.cuda.synchronize()
    input_ = pl.default_input(bands=3)
    pano, info = par.stitch_distributed(input_, views, nx * ny, [c["K"] for c in cams], (512, 512), 0, None, pano_root=0)
    torch.cuda.synchronize()
    return pano.cpu().numpy(), info


def test_resident_and_host_list_paths_agree(gpu, monkeypatch):
    pa, ia = _run(gpu, monkeypatch, False)
    pb, ib = _run(gpu, monkeypatch, True)
    assert ia["pairs"] == ib["pairs"] and len(ia["pairs"]) >= 5
    for ma, mb in zip(ia["models"], ib["models"]):
        assert np.array_equal(np.asarray(ma).view(np.uint64), np.asarray(mb).view(np.uint64))
    assert pa.shape == pb.shape and np.array_equal(pa, pb)
    assert ia["n_components"] == 1 and len(ia["members"]) == 6
    assert (pa.max(axis=2) > 0).mean() > 0.5  # the canvas is mostly covered
