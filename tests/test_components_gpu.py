"""GPU: every connected component becomes a panorama (displayPanorama.m:88-116, recognizePanoramas.m:70-113), in the
single-process driver and in the sharded one, and the 2-rank run of the sharded driver equals the 1-rank run."""
import os
import socket
import sys
from importlib import import_module

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, F = 640, 480, 900.0


def _worlds(synth, device="cuda"):
    """Three independent procedural worlds (different seeds) of 2x2, 3x2 and 2x2 views, shuffled into one set."""
    views, cams, world_of = [], [], []
    for wi, (nx, ny, seed) in enumerate([(2, 2, 11), (3, 2, 23), (2, 2, 37)]):
        cs = synth.grid_cameras(nx, ny, W, H, F, 2 * np.arctan(W / (2 * F)) * 0.6, 2 * np.arctan(H / (2 * F)) * 0.6, 1.0, seed)
        for c in cs:
            views.append(synth.render_view(c, H, W, seed, device, finest_px=6.0))
            cams.append(c)
            world_of.append(wi)
    perm = np.random.default_rng(5).permutation(len(views))
    return [views[k] for k in perm], [cams[k] for k in perm], [world_of[k] for k in perm]


def test_three_worlds_give_three_panoramas_equal_to_stitching_each_alone(gpu):
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    views, cams, world_of = _worlds(synth)
    torch.cuda.synchronize()
    n = len(views)
    inp = pl.default_input(bands=3)
    Ks = [c["K"] for c in cams]
    # sharded driver, one rank, ground-truth cameras: the panorama bytes then depend on the component structure only
    pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, Ks, (512, 512), 0, cams, pano_root=0)
    assert info["n_components"] == 3 and len(info["panoramas"]) == 3
    for c in info["components"]:
        assert len({world_of[k] for k in c["members"]}) == 1, "a component mixes worlds"
    sizes = sorted(len(c["members"]) for c in info["components"])
    assert sizes == [4, 4, 6]
    for c, p in zip(info["components"], info["panoramas"]):
        members = c["members"]
        sub_views = {q: views[k] for q, k in enumerate(members)}
        alone, ia = par.stitch_distributed(inp, sub_views, len(members), [Ks[k] for k in members], (512, 512), 0,
                                           [cams[k] for k in members], pano_root=0)
        assert ia["n_components"] == 1 and len(ia["panoramas"]) == 1
        assert tuple(alone.shape) == tuple(p.shape) and bool(torch.equal(alone, p)), "panorama differs from the world alone"
        assert (p.amax(dim=2) > 0).float().mean().item() > 0.5
    # the main panorama is the component of the best-connected image
    assert any(pano is p for p in info["panoramas"])
    # single-process driver with estimated cameras: the same three panoramas' member sets
    panos, i2 = pl.stitch(inp, views, Ks=Ks, tile=(512, 512))
    assert i2["n_components"] == 3 and len(panos) == 3
    assert [c["members"] for c in i2["components"]] == [c["members"] for c in info["components"]]
    for c in i2["components"]:
        assert 0 <= c["ref"] < len(c["members"]) and all(cam is not None for cam in c["cameras"])


def test_single_images_and_unmatched_sets_render_nothing(gpu):
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    cams = synth.grid_cameras(2, 1, W, H, F, 2.5, 0.0, 0.0, 3)  # two views that do not overlap at all
    views = [synth.render_view(c, H, W, 3 + k, "cuda", finest_px=6.0) for k, c in enumerate(cams)]
    torch.cuda.synchronize()
    pano, info = par.stitch_distributed(pl.default_input(), dict(enumerate(views)), 2, [c["K"] for c in cams], (512, 512))
    assert info["n_pairs_verified"] == 0 and info["panoramas"] == [] and pano.numel() == 0


def test_set_after_set_with_the_next_extraction_started_early_gives_the_same_panoramas(gpu):
    """parallel.submit_features from stitch_distributed's after_matching hook (round 6: the next set's loadImages beside the current
    set's RANSAC / cameras / render, as bench.py's pipelined steps run it): three sets stitched one after the other with the next
    set's extraction under way equal, byte for byte, the same sets stitched strictly one after the other."""
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    sets = []
    for seed in (11, 23, 37):
        cs = synth.grid_cameras(3, 2, W, H, F, 2 * np.arctan(W / (2 * F)) * 0.6, 2 * np.arctan(H / (2 * F)) * 0.6, 1.0, seed)
        sets.append((dict(enumerate(synth.render_view(c, H, W, seed, "cuda", finest_px=6.0) for c in cs)), [c["K"] for c in cs]))
    torch.cuda.synchronize()
    inp = pl.default_input(bands=3)
    plain = []
    for views, Ks in sets:
        pano, info = par.stitch_distributed(inp, views, len(views), Ks, (512, 512), 0, None, pano_root=0)
        plain.append((pano.cpu().numpy(), info["n_pairs_verified"], info["n_features"]))
    ahead = [None]
    hooks = []
    for k, (views, Ks) in enumerate(sets):
        feat, ahead[0] = ahead[0], None

        def start_next(k=k):
            hooks.append(k)
            ahead[0] = par.submit_features(inp, sets[k + 1][0])

        pano, info = par.stitch_distributed(inp, views, len(views), Ks, (512, 512), 0, None, pano_root=0, features=feat,
                                            after_matching=start_next if k + 1 < len(sets) else None)
        assert (k == 0) == (feat is None)
        assert info["n_pairs_verified"] == plain[k][1] and info["n_features"] == plain[k][2]
        assert np.array_equal(pano.cpu().numpy(), plain[k][0])
    assert hooks == [0, 1] and ahead[0] is None
    # the same with the images uploaded on a side stream (one event per image) and the extraction started in two parts
    # (submit_features(first=) from after_matching, submit_features_rest from after_ransac)
    side = torch.cuda.Stream()
    host = [{i: v.cpu().pin_memory() for i, v in views.items()} for views, _ in sets]

    def upload(k):
        up, evs = {}, {}
        with torch.cuda.stream(side):
            for i, h in host[k].items():
                up[i] = h.to("cuda", non_blocking=True)
                evs[i] = torch.cuda.Event()
                evs[i].record(side)
        return up, evs

    nxt = [None]
    for k, (_, Ks) in enumerate(sets):
        if nxt[0] is None:
            up, evs = upload(k)
            feat = None
        else:
            up, evs, feat = nxt[0]
            nxt[0] = None

        def first_part(k=k):
            u, e = upload(k + 1)
            nxt[0] = (u, e, par.submit_features(inp, u, e, first=2))
            assert len(nxt[0][2]["futures"]) == 2

        def second_part():
            par.submit_features_rest(nxt[0][2])
            assert len(nxt[0][2]["futures"]) == 6 and nxt[0][2]["rest"] is None

        last = k + 1 == len(sets)
        pano, info = par.stitch_distributed(inp, up, len(up), Ks, (512, 512), 0, None, pano_root=0, image_events=evs, features=feat,
                                            after_matching=None if last else first_part, after_ransac=None if last else second_part)
        assert info["n_pairs_verified"] == plain[k][1] and np.array_equal(pano.cpu().numpy(), plain[k][0])
    # a handle for other images is refused
    wrong = par.submit_features(inp, {0: sets[0][0][0], 1: sets[0][0][1]})
    with pytest.raises(AssertionError):
        par.stitch_distributed(inp, sets[0][0], 6, sets[0][1], (512, 512), 0, None, pano_root=0, features=wrong)
    for f in wrong["futures"]:
        f.result()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q, mode):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["APS_DEVICE"] = "0"
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(0)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        import apsamd

        synth = import_module(apsamd.__name__ + ".synth")
        pl = import_module(apsamd.__name__ + ".pipeline")
        par = import_module(apsamd.__name__ + ".parallel")
        if mode == "worlds":
            views, cams, _ = _worlds(synth)
            gt = None
        else:
            cams = synth.grid_cameras(3, 2, W, H, F, 2 * np.arctan(W / (2 * F)) * 0.6, 2 * np.arctan(H / (2 * F)) * 0.6, 1.0, 7)
            views = [synth.render_view(c, H, W, 7, "cuda", finest_px=6.0) for c in cams]
            gt = None
        torch.cuda.synchronize()
        n = len(views)
        local = {i: views[i] for i in par.shard_indices(n, world, rank)}
        pano, info = par.stitch_distributed(pl.default_input(bands=3), local, n, [c["K"] for c in cams], (256, 256), 0, gt,
                                            pano_root=0)
        torch.cuda.synchronize()
        out = None
        if rank == 0:
            out = {"pairs": info["pairs"], "models": [np.asarray(m).copy() for m in info["models"]],
                   "panos": [p.cpu().numpy() for p in info["panoramas"]], "ncomp": info["n_components"],
                   "members": [c["members"] for c in info["components"]]}
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        q.put((rank, "ok", out))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc(), None))
        raise


def _spawn(world, mode):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res if r[1] != "ok"]
    return next(r[2] for r in res if r[0] == 0)


@pytest.mark.parametrize("mode", ["grid", "worlds"])
def test_two_ranks_equal_one_rank(gpu, mode):
    """stitch_distributed with 2 ranks (both on this GPU, gloo with host-staged collectives) against the 1-rank run:
    verified pairs, model bits and every panorama byte.  'grid': one panorama, tiles sharded; 'worlds': three
    panoramas, components sharded (3 components >= 2 ranks)."""
    one = _spawn(1, mode)
    two = _spawn(2, mode)
    assert one["pairs"] == two["pairs"] and len(one["pairs"]) >= 5
    for a, b in zip(one["models"], two["models"]):
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    assert one["ncomp"] == two["ncomp"] and one["members"] == two["members"]
    assert len(one["panos"]) == len(two["panos"]) == (3 if mode == "worlds" else 1)
    for a, b in zip(one["panos"], two["panos"]):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_overlapped_extraction_and_matching_changes_nothing(gpu, monkeypatch):
    """APS_MATCH_OVERLAP_CHUNK: the pairs of every finished chunk of images are matched while the worker streams extract
    the next chunk (parallel._match_pass).  Verified pairs, models and every panorama byte must equal the sequential
    run's, and two overlapped runs must equal each other (the extraction must not be disturbed by the concurrent
    matching kernels: the int8 screening kernel keeps its SIMDs to itself for that reason)."""
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    views, cams, _ = _worlds(synth)
    torch.cuda.synchronize()
    n = len(views)
    inp = pl.default_input(bands=3)
    Ks = [c["K"] for c in cams]

    def run():
        pano, info = par.stitch_distributed(inp, dict(enumerate(views)), n, Ks, (512, 512), 0, None, pano_root=0)
        return info

    monkeypatch.delenv("APS_MATCH_OVERLAP_CHUNK", raising=False)
    ref = run()
    for chunk in ("4", "5", "4"):
        monkeypatch.setenv("APS_MATCH_OVERLAP_CHUNK", chunk)
        got = run()
        assert got["n_features"] == ref["n_features"]
        assert got["n_pairs_verified"] == ref["n_pairs_verified"] and got["n_components"] == ref["n_components"]
        assert len(got["panoramas"]) == len(ref["panoramas"])
        for a, b in zip(got["panoramas"], ref["panoramas"]):
            assert tuple(a.shape) == tuple(b.shape) and bool(torch.equal(a, b))
    monkeypatch.delenv("APS_MATCH_OVERLAP_CHUNK", raising=False)


# ---- the second matching pass (imageMatchingPanoramaConComps.m:48-91) ----------------------------------------------------
def _worlds_of_three_sizes(synth):
    """Three worlds whose ORIGINALS differ in size and aspect: a global 'fit' brings all of them to one common size
    (anisotropically), the per-component 'fit' of the second pass keeps each world's own aspect."""
    spec = [(2, 2, 11, 800, 600, 1125.0), (3, 2, 23, 640, 400, 900.0), (2, 2, 37, 720, 720, 1010.0)]
    views, Ks, world_of = [], [], []
    for wi, (nx, ny, seed, w, h, f) in enumerate(spec):
        cs = synth.grid_cameras(nx, ny, w, h, f, 2 * np.arctan(w / (2 * f)) * 0.6, 2 * np.arctan(h / (2 * f)) * 0.6, 1.0, seed)
        for c in cs:
            views.append(synth.render_view(c, h, w, seed, "cuda", finest_px=6.0))
            Ks.append(c["K"])
            world_of.append(wi)
    perm = np.random.default_rng(9).permutation(len(views))
    return [views[k] for k in perm], [Ks[k] for k in perm], [world_of[k] for k in perm]


@pytest.mark.parametrize("driver", ["single", "sharded"])
def test_second_pass_on_images_resized_per_component(gpu, driver):
    """resizeImage = 1 and resizeImagePanoramaCluster = 1 with more than one component: the images are resized per
    component from the ORIGINALS, features are re-extracted and everything is re-matched and re-verified.  The run
    must (a) take the branch, (b) see different feature counts than the first pass, (c) keep every resized image on the
    device, and (d) produce exactly the panoramas of a plain run on images resized that way beforehand."""
    import torch

    synth = import_module(gpu.__name__ + ".synth")
    pl = import_module(gpu.__name__ + ".pipeline")
    par = import_module(gpu.__name__ + ".parallel")
    ip = import_module(gpu.__name__ + ".imageProcessing")
    originals, Ks_o, world_of = _worlds_of_three_sizes(synth)
    torch.cuda.synchronize()
    n = len(originals)
    hw_o = [(int(v.shape[0]), int(v.shape[1])) for v in originals]
    inp = pl.default_input(bands=3, resizeImage=1, resizeImagePanoramaCluster=1, heightLimit=480, widthLimit=480)
    # loadImages.m:66-68: the first pass runs on resizeImagesToLimits(ALL images, 'fit') = one component holding everything
    first = pl.resize_per_component(inp, dict(enumerate(originals)), np.zeros(n, np.int64), hw_o)
    first = [first[k] for k in range(n)]
    assert all(v.is_cuda for v in first) and len({tuple(v.shape) for v in first}) == 1
    # the device resize is the host resize
    assert np.array_equal(first[0].cpu().numpy(), ip.imresize(ip.imresize(originals[0].cpu().numpy(),
                          pl.fit_size(*hw_o[0], 480, 480)[2], "bicubic"), tuple(first[0].shape[:2]), "bicubic"))
    Ks_1 = [pl.rescale_K(K, hw_o[k], first[k].shape[:2]) for k, K in enumerate(Ks_o)]
    if driver == "single":
        panos, info = pl.stitch(inp, first, Ks=Ks_1, tile=(256, 256), images_original=originals)
    else:
        _, info = par.stitch_distributed(inp, dict(enumerate(first)), n, Ks_1, (256, 256), 0, None, pano_root=0,
                                         local_originals=dict(enumerate(originals)))
        panos = info["panoramas"]
    assert info["second_pass"] is True and info["n_components"] == 3 and len(panos) == 3
    assert list(info["n_features"]) != list(info["n_features_first_pass"])
    for c in info["components"]:
        assert len({world_of[k] for k in c["members"]}) == 1
    # the same images prepared beforehand, plain run without the flags
    labels = np.asarray(info["labels"])
    pre = pl.resize_per_component(inp, dict(enumerate(originals)), labels, hw_o)
    pre = [pre[k] for k in range(n)]
    assert len({tuple(v.shape) for v in pre}) == 3  # every world kept its own aspect
    Ks_2 = [pl.rescale_K(K, first[k].shape[:2], pre[k].shape[:2]) for k, K in enumerate(Ks_1)]
    plain = pl.default_input(bands=3)
    if driver == "single":
        panos_b, info_b = pl.stitch(plain, pre, Ks=Ks_2, tile=(256, 256))
    else:
        _, info_b = par.stitch_distributed(plain, dict(enumerate(pre)), n, Ks_2, (256, 256), 0, None, pano_root=0)
        panos_b = info_b["panoramas"]
    assert info_b["second_pass"] is False and list(info_b["n_features"]) == list(info["n_features"])
    assert [c["members"] for c in info_b["components"]] == [c["members"] for c in info["components"]]
    assert len(panos_b) == 3
    for a, b in zip(panos, panos_b):
        assert tuple(a.shape) == tuple(b.shape) and bool(torch.equal(a, b))
        assert (a.amax(dim=2) > 0).float().mean().item() > 0.4
