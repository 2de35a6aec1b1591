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


def test_extraction_beside_matching_changes_no_bit(gpu):
    """Feature extraction on the worker streams while the main thread runs the batched matcher (int8 screening kernel,
    f16 list pass, filter) - the overlap of scripts/probe/probe_overlap_race.py as a regression test.  Round 2 found
    SIFT's refine / orientation / descriptor kernels returning different results when they shared a SIMD with int8-MFMA
    waves; the screening kernel therefore keeps its SIMDs to itself (match_screen_i8_kernel claims the whole register
    file).  Whatever co-runs, every descriptor, location and match must equal the quiet run's."""
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    fm = import_module(gpu.__name__ + ".featureMatching")
    W, H, f = 1920, 1080, 4000.0
    imgs, _ = synth.make_scene(4, 3, W, H, f, 0.4, seed=5, device="cuda", finest_px=8.0)
    torch.cuda.synchronize()
    inp = pl.default_input()
    quiet = pl.sift_many(inp, imgs)
    descs = [d for d, _ in quiet[:8]]
    order = fm.pair_order(len(descs))
    mq = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True)
    assert min(int(d.shape[0]) for d in descs) > 3000 and int(mq[0][-1]) > 500
    sig = lambda m: (m[0].tolist(), m[1].tolist(), m[2].tolist(), np.asarray(m[3]).view(np.uint32).tolist())  # noqa: E731
    n_match_calls = 0
    for _ in range(3):
        futs = pl.sift_submit(inp, imgs)
        while not all(fu.done() for fu in futs):
            m = fm.match_pairs_csr(descs, order, inp["Ratiothreshold"], inp["Matchingthreshold"], True)
            assert sig(m) == sig(mq), "a match list changed while features were being extracted"
            n_match_calls += 1
        for (d, p), (dq, pq) in zip([fu.result() for fu in futs], quiet):
            assert d.shape == dq.shape and bool(torch.equal(d.view(torch.int32), dq.view(torch.int32)))
            assert np.array_equal(p.view(np.uint64), pq.view(np.uint64))
    assert n_match_calls >= 1
