"""Size-independent properties at the sizes BASELINE.json quotes (the oracle would need minutes to hours here):
~20k-feature all-pairs matching, a 4K SIFT, a multi-tile 4K render.  These complement the bit-exact parity tests,
which run at sizes the oracle finishes in seconds."""
import os
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(gpu):
    g = gpu.__name__
    return {k: import_module(g + "." + k) for k in ("featureMatching", "synth", "renderPanorama", "imageProcessing")}


def test_matching_20k_permutation_and_mode_equality(mods, monkeypatch):
    """A x permuted(A) at Kf = 19828: every row's nearest neighbour is its own copy at distance 0 (to rounding), the one-to-one
    filter keeps all rows, and the certified split-bf16 path returns the same bits as the all-f32 path."""
    import torch

    fm = mods["featureMatching"]
    n = 19828
    g = torch.Generator(device="cuda").manual_seed(5)
    A = torch.rand(n, 128, device="cuda", generator=g) ** 3
    A = (A / A.norm(dim=1, keepdim=True)).contiguous()
    perm = torch.randperm(n, device="cuda", generator=g)
    B = A[perm].contiguous()
    torch.cuda.synchronize()
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(n, device="cuda")
    res = {}
    for mode in ("split", "f32"):
        monkeypatch.setenv("APS_MATCH_MODE", mode)
        _, idx, d1, d2 = fm.nearest2SSDExhaustive(A, B)
        res[mode] = (idx, d1, d2)
    monkeypatch.delenv("APS_MATCH_MODE")
    idx, d1, d2 = res["split"]
    assert np.array_equal(idx.astype(np.int64) - 1, inv.cpu().numpy())
    # a2 and b2 come from a mul/add chain, the dot product from an fma chain: a copy is at distance 0 up to a few ulp of 2
    assert np.all(np.abs(d1) <= 1e-6) and np.all(d2 > 1e-3)
    for a, b in zip(res["split"], res["f32"]):
        assert np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    m, met = fm.matchFeaturesScratch(A, B, MatchThreshold=1.5, MaxRatio=0.6)
    assert len(m) == n and np.array_equal(m[:, 1].astype(np.int64) - 1, inv.cpu().numpy()[m[:, 0].astype(np.int64) - 1])
    assert np.all(np.abs(met) <= 1e-6)


def test_sift_4k_is_deterministic_and_well_formed(mods):
    fm, synth = mods["featureMatching"], mods["synth"]
    imgs, _ = synth.make_scene(1, 1, 3840, 2160, 8000.0, device="cuda", finest_px=16.0)
    inp = {"detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6}
    f1, p1 = fm.sift_extract(inp, imgs[0])
    f2, p2 = fm.sift_extract(inp, imgs[0])
    assert f1.shape == f2.shape and np.array_equal(f1.view(np.uint32), f2.view(np.uint32)) and np.array_equal(p1, p2)
    assert f1.shape[0] > 10000 and f1.shape[1] == 128 and f1.dtype == np.float32 and p1.dtype == np.float64
    assert np.allclose(np.linalg.norm(f1, axis=1), 1.0, atol=1e-5) and f1.min() >= 0
    assert p1[:, 0].min() >= 1 and p1[:, 0].max() <= 3840 and p1[:, 1].min() >= 1 and p1[:, 1].max() <= 2160


def test_render_4k_multitile_batched_equals_per_tile_path(mods, monkeypatch):
    """16 4K views (4 x 4 grid), 2048^2 tiles, 5 bands: the batched level-major path (render_batch.hip, warp in its
    exact mode) against the per-tile path with its footprint culls, block-level image cull and fused levels all off,
    byte for byte; then the default fast warp against those within the stated tolerance."""
    import torch

    synth, rp = mods["synth"], mods["renderPanorama"]
    W, H, f = 3840, 2160, 8000.0
    cams = synth.grid_cameras(4, 4, W, H, f, 2 * np.arctan(W / (2 * f)) * 0.6, 2 * np.arctan(H / (2 * f)) * 0.6, 1.0, 12345)
    imgs = [synth.render_view(c, H, W, 12345, "cuda", finest_px=16.0) for c in cams]
    torch.cuda.synchronize()
    opts = {"anglePower": 2, "blending": "multiband", "pyrLevels": 5, "pyrSigma": 1.0, "tile": (2048, 2048), "cropBorder": False}
    sizes = [(H, W, 3)] * 16
    outs = []
    keys = ("APS_RENDER_LEGACY", "APS_RENDER_NO_CULL", "APS_RENDER_NO_FUSE", "APS_RENDER_CHECK_RECTS", "APS_WARP_EXACT")
    for env in ({"APS_RENDER_CHECK_RECTS": "1", "APS_WARP_EXACT": "1"}, {"APS_RENDER_LEGACY": "1"},
                {"APS_RENDER_LEGACY": "1", "APS_RENDER_NO_CULL": "1", "APS_RENDER_NO_FUSE": "1"}, {}):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pano, _ = rp.renderPanorama({}, imgs, sizes, cams, "spherical", 5, opts, device_out=True)
        outs.append(pano)
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    assert outs[0].shape[0] > 4096 and outs[0].shape[1] > 8192 and int((outs[0] > 0).sum()) > 5e7
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    # the default (fast-warp) render against the exact one: within one grey level on >= 99.95 %, never more than two
    dd = (outs[3].to(torch.int16) - outs[0].to(torch.int16)).abs()
    assert int(dd.max()) <= 2 and float((dd <= 1).float().mean()) >= 0.9995
    assert float((dd == 0).float().mean()) >= 0.97


def test_crop_rectangle_on_a_large_canvas(mods):
    """A 60 MPix canvas with a wavy outline, holes and a bay: the device rectangle must equal the oracle's, and the
    round count of the fill must stay small on a panorama-like shape (checked through its time share)."""
    import oracle

    ip = mods["imageProcessing"]
    H, W = 6100, 9900
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    rim = 0.45 + 0.03 * np.sin(xx / 400.0)
    m = ((yy - H / 2) / (H * rim)) ** 2 + ((xx - W / 2) / (W * 0.48)) ** 2 < 1
    m &= ~(((yy - 2000) ** 2 + (xx - 3000) ** 2) < 120 ** 2)
    m &= ~((yy < 1500) & (np.abs(xx - 6000) < 200))
    img = np.zeros((H, W, 3), np.uint8)
    img[m] = (170, 140, 90)
    want_rect, want_ok, _ = oracle.crop_rect(img)
    got_rect, got_ok = ip.cropRectangle(img)
    assert got_rect == want_rect and got_ok == want_ok and want_ok


def test_bench_scale_matching_screen_on_equals_screen_off(gpu, mods, monkeypatch):
    """BASELINE configs[2]'s matching workload in full: the 64 x 4K bench scene's descriptors (~19.8 k per view), all 2016
    pairs.  The int8 screening pre-pass + row-list f16 pass must return exactly the lists of the f16 path on every row
    (APS_MATCH_NO_SCREEN=1) - pair offsets, both index lists and the metric's bits - and the lists must be one-to-one per
    pair.  (The oracle needs ~20 minutes for this; the small-size tests pin both paths to it.)"""
    import ctypes
    import torch

    fm, synth = mods["featureMatching"], mods["synth"]
    pl = import_module(gpu.__name__ + ".pipeline")
    imgs, _ = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
    torch.cuda.synchronize()
    inp = pl.default_input(bands=5)
    descs = [d for d, _ in pl.sift_many(inp, imgs)]
    del imgs
    torch.cuda.synchronize()
    assert 15000 < np.mean([d.shape[0] for d in descs]) < 25000
    order = fm.pair_order(len(descs))
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    pp, ia, ib, met = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    rows, surv = ctypes.c_int64(0), ctypes.c_int64(0)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_stats(ctypes.byref(rows), ctypes.byref(surv)))
    assert rows.value == sum(descs[i].shape[0] for i, _ in order) and int(pp[-1]) <= surv.value < 0.25 * rows.value
    monkeypatch.setenv("APS_MATCH_NO_SCREEN", "1")
    pp0, ia0, ib0, met0 = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    monkeypatch.delenv("APS_MATCH_NO_SCREEN", raising=False)
    assert np.array_equal(pp, pp0) and int(pp[-1]) > 100000
    assert bool(torch.equal(ia, ia0)) and bool(torch.equal(ib, ib0)) and bool(torch.equal(met.view(torch.int32), met0.view(torch.int32)))
    del pp0, ia0, ib0, met0
    # round 6, the last hop of the chain at this scale: the all-f32 kernel (v_mfma_f32_32x32x2_f32 = the k-ascending fma chain of
    # the contract, the kernel the small-size tests pin to the oracle) on EVERY row of all 2016 pairs - same pair offsets, both
    # index lists, the metric's bits
    monkeypatch.setenv("APS_MATCH_MODE", "f32")
    pp3, ia3, ib3, met3 = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    monkeypatch.delenv("APS_MATCH_MODE", raising=False)
    assert np.array_equal(pp, pp3)
    assert bool(torch.equal(ia, ia3)) and bool(torch.equal(ib, ib3)) and bool(torch.equal(met.view(torch.int32), met3.view(torch.int32)))
    del pp3, ia3, ib3, met3
    # round 6: the SIFT stage's descriptors are integers over their norm, so every job ran on exact int8 codes; with the
    # general codes (APS_MATCH_NO_EXACT=1) more rows survive the screen and the lists are the same
    jobs, exact = ctypes.c_int64(0), ctypes.c_int64(0)
    monkeypatch.setenv("APS_MATCH_NO_EXACT", "1")
    pp4, ia4, ib4, met4 = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    monkeypatch.delenv("APS_MATCH_NO_EXACT", raising=False)
    rows4, surv4 = ctypes.c_int64(0), ctypes.c_int64(0)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_stats(ctypes.byref(rows4), ctypes.byref(surv4)))
    gpu._capi.check(gpu._capi.lib.aps_match_screen_exact_jobs(ctypes.byref(jobs), ctypes.byref(exact)))
    assert (jobs.value, exact.value) == (len(order), 0) and surv.value < 0.7 * surv4.value
    assert np.array_equal(pp, pp4) and bool(torch.equal(ia, ia4)) and bool(torch.equal(ib, ib4))
    assert bool(torch.equal(met.view(torch.int32), met4.view(torch.int32)))
    del pp4, ia4, ib4, met4
    fm.match_pairs_csr(descs[:8], fm.pair_order(8), 0.6, 1.5, True, device_out=True)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_exact_jobs(ctypes.byref(jobs), ctypes.byref(exact)))
    assert (jobs.value, exact.value) == (28, 28)
    # the screening pass on the other MFMA shape (v_mfma_i32_32x32x32_i8, rounds 2-3; the default is 16x16x64): its groups of
    # columns differ, so its survivor set may differ by a few rows - the lists may not
    monkeypatch.setenv("APS_SCREEN_SHAPE", "32")
    pp2, ia2, ib2, met2 = fm.match_pairs_csr(descs, order, 0.6, 1.5, True, device_out=True)
    monkeypatch.delenv("APS_SCREEN_SHAPE", raising=False)
    rows2, surv2 = ctypes.c_int64(0), ctypes.c_int64(0)
    gpu._capi.check(gpu._capi.lib.aps_match_screen_stats(ctypes.byref(rows2), ctypes.byref(surv2)))
    assert rows2.value == rows.value and abs(surv2.value - surv4.value) < 0.01 * surv4.value  # (this shape runs the general codes)
    assert np.array_equal(pp, pp2) and bool(torch.equal(ia, ia2)) and bool(torch.equal(ib, ib2))
    assert bool(torch.equal(met.view(torch.int32), met2.view(torch.int32)))
    # one-to-one per pair: within a pair's segment no column index repeats
    seg = torch.repeat_interleave(torch.arange(len(order), device="cuda"), torch.from_numpy(np.diff(pp)).to("cuda"))
    key = seg.to(torch.int64) * (1 << 32) + ib.to(torch.int64)
    assert int(torch.unique(key).numel()) == int(key.numel())


def _configs4(gpu, mods, **switches):
    """BASELINE configs[4] as SURVEY 8(d) cfg5 specifies it, on one GPU: 500 mixed 2048 x 1080 views drawn from SIX
    independent worlds of 60 - 110 views each (different seeds, grids and overlaps, one of them a full 360 degree ring),
    shuffled -> matched -> connected components -> six equirectangular panoramas.  Properties only (the oracle would need
    hours): every component is exactly one world, every panorama is rendered, covered and of a plausible size."""
    import time
    import torch

    synth = mods["synth"]
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    W, H, f = 2048, 1080, 2400.0
    fov_x = 2 * np.arctan(W / (2 * f))
    worlds = [(10, 6, 0.40), (10, 7, 0.40), (10, 8, 0.40), (10, 9, 0.45), (19, 5, 1.0 - (2 * np.pi / 19) / fov_x), (15, 7, 0.45)]
    assert sum(nx * ny for nx, ny, _ in worlds) == 500 and all(60 <= nx * ny <= 110 for nx, ny, _ in worlds)
    views, Ks, world_of = [], [], []
    for wi, (nx, ny, ov) in enumerate(worlds):
        imgs, cams = synth.make_scene(nx, ny, W, H, f, ov, seed=1000 + 17 * wi, device="cuda", finest_px=10.0)
        views += imgs
        Ks += [c["K"] for c in cams]
        world_of += [wi] * len(imgs)
    perm = np.random.default_rng(9).permutation(len(views))
    views, Ks, world_of = [views[k] for k in perm], [Ks[k] for k in perm], [world_of[k] for k in perm]
    torch.cuda.synchronize()
    n = len(views)
    assert n == 500
    inp = pl.default_input(bands=5, panorama2DisplaynSave="equirectangular", **switches)
    t0 = time.perf_counter()
    pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, Ks, (2048, 2048), 0, None, pano_root=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"configs[4] (6 worlds, 500 views of 2048 x 1080, {switches or 'pairwise matcher'}) on one GPU: {dt:.2f} s, stages {info['times']}")
    assert info["n_components"] == len(worlds) and len(info["panoramas"]) == len(worlds)
    sizes = sorted(nx * ny for nx, ny, _ in worlds)
    assert sorted(len(c["members"]) for c in info["components"]) == sizes
    for c, p in zip(info["components"], info["panoramas"]):
        assert len({world_of[k] for k in c["members"]}) == 1
        assert p.dtype == torch.uint8 and p.shape[2] == 3 and p.shape[1] > 3 * W and p.shape[0] > 2 * H
        assert (p[::2, ::2].amax(dim=2) > 0).float().mean().item() > 0.5
    # the ring world closes: its panorama spans (nearly) the full 2 pi f of an equirectangular canvas
    ring = next(p for c, p in zip(info["components"], info["panoramas"]) if len(c["members"]) == 95)
    assert ring.shape[1] > 0.9 * 2 * np.pi * f
    assert any(pano is p for p in info["panoramas"])
    return info


def test_multi_panorama_recognition_at_configs4_image_count(gpu, mods):
    """configs[4] with every pair matched exhaustively (matchFeaturesPairwise = 1, 124 750 pairs)."""
    _configs4(gpu, mods)


def test_configs4_with_the_reference_default_pooled_matcher(gpu, mods):
    """configs[4] with the reference's DEFAULT matcher switch (inputs.m:46 matchFeaturesPairwise = 0: featureMatchingGlobal.m:69-161,
    the "global matcher" of BASELINE.md row 5): one pool of ~2.7 M descriptors, exact 4-NN of the pool against itself
    through the int8 proof pass, per-query filter.  Same properties: six components, each exactly one world."""
    import ctypes

    info = _configs4(gpu, mods, matchFeaturesPairwise=0, k=4)
    rows, searched = ctypes.c_int64(0), ctypes.c_int64(0)
    gpu._capi.check(gpu._capi.lib.aps_knn_global_screen_stats(ctypes.byref(rows), ctypes.byref(searched)))
    print(f"pooled matcher at configs[4] size: {sum(info['n_features'])} pool rows, (row, image) slots {rows.value}, searched {searched.value}")


def test_256_views_4k_at_configs3_image_count(gpu, mods):
    """BASELINE configs[3]'s image set as SURVEY 8(d) cfg4 specifies it, on one GPU (the 8-GPU sharding itself needs the
    node): 256 views of 3840 x 2160, f = 8000 px, on a 32 x 8 yaw/pitch grid with a yaw step of 11.25 degrees - a FULL
    360 degree ring (58 % overlap), theta running from -pi to pi across the canvas - and a pitch step of 9.2 degrees; all
    32 640 pairs matched, one component, one spherical panorama of about 2 pi f x 1.4 f pixels.  Properties only."""
    import time
    import torch

    synth = mods["synth"]
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    W, H, f, nx, ny = 3840, 2160, 8000.0, 32, 8
    cams = synth.grid_cameras(nx, ny, W, H, f, np.radians(11.25), np.radians(9.2), 1.0, 12345)
    views = [synth.render_view(c, H, W, 12345, "cuda", finest_px=16.0) for c in cams]
    torch.cuda.synchronize()
    n = len(views)
    assert n == 256
    inp = pl.default_input(bands=5)
    t0 = time.perf_counter()
    pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, [c["K"] for c in cams], (2048, 2048), 0, None, pano_root=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"configs[3] image set (32 x 8 ring) on one GPU: {dt:.2f} s, panorama {tuple(pano.shape)}, verified pairs "
          f"{info['n_pairs_verified']}, stages {info['times']}")
    assert info["n_components"] == 1 and len(info["panoramas"]) == 1 and sorted(info["members"]) == list(range(n))
    # the 4-neighbour pairs of the grid verify, the ring closure (column 31 next to column 0) included
    pairs = {tuple(sorted(p)) for p in info["pairs"]}
    ring_ok = sum(tuple(sorted((iy * nx + ix, iy * nx + (ix + 1) % nx))) in pairs for iy in range(ny) for ix in range(nx))
    col_ok = sum((iy * nx + ix, (iy + 1) * nx + ix) in pairs for iy in range(ny - 1) for ix in range(nx))
    assert ring_ok >= 0.95 * nx * ny and col_ok >= 0.95 * nx * (ny - 1), (ring_ok, col_ok)
    assert any(tuple(sorted((iy * nx, iy * nx + nx - 1))) in pairs for iy in range(ny)), "the ring does not close"
    # a full-circle canvas: ~2 pi f wide (theta wraps at the canvas edge), ~8 rows of 9.2 degrees + one field of view high
    assert pano.dtype == torch.uint8 and pano.shape[2] == 3
    assert pano.shape[1] > 0.95 * 2 * np.pi * f and pano.shape[0] > 1.2 * f
    assert (pano[::4, ::4].amax(dim=2) > 0).float().mean().item() > 0.6


def test_bench_scale_set_after_set_equals_the_sequential_stitch(gpu, mods):
    """The 64 x 4K job twice, the second stitch's extraction started from the first one's after_matching hook (bench.py's pipelined
    steps): verified pairs, feature counts and every byte of both panoramas equal one plain stitch of the same views."""
    import torch

    synth = mods["synth"]
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    imgs, cams = synth.make_scene(8, 8, 3840, 2160, 8000.0, 0.4, device="cuda", finest_px=16.0)
    local = dict(enumerate(imgs))
    Ks = [c["K"] for c in cams]
    inp = pl.default_input(bands=5)
    torch.cuda.synchronize()
    pano0, info0 = par.stitch_distributed(inp, local, len(imgs), Ks, (2048, 2048), 0, None, pano_root=0)
    ref = (pano0.cpu(), info0["n_pairs_verified"], list(info0["n_features"]))
    del pano0
    nxt = [None]

    def start_next():
        nxt[0] = par.submit_features(inp, local)

    pano1, info1 = par.stitch_distributed(inp, local, len(imgs), Ks, (2048, 2048), 0, None, pano_root=0, after_matching=start_next)
    assert nxt[0] is not None and len(nxt[0]["futures"]) == 64
    p1 = pano1.cpu()
    del pano1
    pano2, info2 = par.stitch_distributed(inp, local, len(imgs), Ks, (2048, 2048), 0, None, pano_root=0, features=nxt[0])
    for p, info in ((p1, info1), (pano2.cpu(), info2)):
        assert info["n_pairs_verified"] == ref[1] and list(info["n_features"]) == ref[2]
        assert p.shape == ref[0].shape and torch.equal(p, ref[0])
    pl.release_device_memory()
