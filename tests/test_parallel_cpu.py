"""CPU (gloo, world_size 2) tests of the multi-GPU sharding logic in parallel.py: partitioning, the ragged
all-gather and the item exchange.  The compute kernels need a GPU; the collective plumbing does not."""
import os
import socket
import sys
from importlib import import_module

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import apsamd

        par = import_module(apsamd.__name__ + ".parallel")
        dev = torch.device("cpu")
        assert par.world() == (world, rank)
        # ragged all-gather: rank r contributes r+2 rows
        local = torch.arange((rank + 2) * 3, dtype=torch.float32).reshape(rank + 2, 3) + 100 * rank
        parts = par.allgather_ragged(local)
        assert [p.shape[0] for p in parts] == [2, 3]
        assert torch.equal(parts[rank], local) and parts[1 - rank][0, 0] == 100 * (1 - rank)
        # item exchange: 7 items owned i % 2, item i has i rows filled with i (item 0 is empty)
        n = 7
        ids = par.shard_indices(n, world, rank)
        arrays = [np.full((i, 2), i, np.int64) for i in ids]
        allit = par.exchange_items(ids, arrays, n, lambda i: i % world, 2, np.int64, dev)
        for i in range(n):
            assert allit[i].shape == (i, 2) and (allit[i] == i).all(), i
        # uneven ownership through a weighted partition
        w = [5.0, 1.0, 1.0, 1.0, 4.0, 1.0]
        own = par.partition_weighted(w, world)
        mine = [k for k in range(len(w)) if own[k] == rank]
        arrays = [np.full((k + 1, 1), float(k)) for k in mine]
        allit = par.exchange_items(mine, arrays, len(w), lambda k: int(own[k]), 1, np.float64, dev)
        assert all(a.shape == (k + 1, 1) and (a == k).all() for k, a in enumerate(allit))
        # the early image all-gather: 5 images of 6x4x3, image i on rank i % world
        imgs = {i: torch.full((6, 4, 3), 10 * i + 1, dtype=torch.uint8) for i in par.shard_indices(5, world, rank)}
        ig = par.ImageGather(imgs, 5)
        got = ig.wait()
        assert len(got) == 5 and all(g.shape == (6, 4, 3) and int(g[0, 0, 0]) == 10 * i + 1 and bool((g == g[0, 0, 0]).all())
                                     for i, g in enumerate(got))
        # ragged sizes and a rank that owns nothing (n < world): 1 image of 3x5x3 on rank 0 only
        one = {0: torch.full((3, 5, 3), 9, dtype=torch.uint8)} if rank == 0 else {}
        got1 = par.ImageGather(one, 1, dev).wait()
        assert len(got1) == 1 and got1[0].shape == (3, 5, 3) and bool((got1[0] == 9).all())
        mixed = {i: torch.full((2 + i, 3, 3), i + 1, dtype=torch.uint8) for i in par.shard_indices(3, world, rank)}
        gotm = par.ImageGather(mixed, 3, dev).wait()
        assert [tuple(g.shape) for g in gotm] == [(2, 3, 3), (3, 3, 3), (4, 3, 3)]
        assert all(bool((g == i + 1).all()) for i, g in enumerate(gotm))
        # the chunked feature exchange: 7 images, image i on rank i % world with 3 + 2 i descriptors; the local futures
        # resolve late and out of order (a worker pool), every rank runs the same rounds
        from concurrent.futures import ThreadPoolExecutor
        import time as _time

        nimg = 7

        def extract(i):
            _time.sleep(0.01 * ((i * 5) % 3))
            k = 3 + 2 * i
            d = torch.full((k, 128), float(i), dtype=torch.float32) + torch.arange(k, dtype=torch.float32)[:, None]
            return d, np.stack([np.arange(k, dtype=np.float64) + 0.25 * i, np.full(k, float(i))], axis=1)

        with ThreadPoolExecutor(3) as pool:
            futs = {i: pool.submit(extract, i) for i in par.shard_indices(nimg, world, rank)}
            for rounds in (1, 2, 4):
                ex = par.FeatureExchange(futs, nimg, dev, rounds=rounds).run()
                assert ex.rounds == min(rounds, (nimg + world - 1) // world)
                descs, kps = ex.wait()
                for i in range(nimg):
                    d, p = extract(i)
                    assert torch.equal(descs[i], d) and np.array_equal(kps[i].numpy(), p), (rounds, i)
        # tiles to the root: canvas 7x10x3 with 3x4 tiles (ragged edge tiles), tile t painted with t+1 by rank t % world
        H, W = 7, 10
        rects = par.tile_rects(H, W, (3, 4))
        assert len(rects) == 9 and rects[0] == (0, 0, 3, 4) and rects[-1] == (6, 8, 1, 2)
        pano = torch.zeros((H, W, 3), dtype=torch.uint8)
        for t, (r0, c0, ht, wt) in enumerate(rects):
            if t % world == rank:
                pano[r0:r0 + ht, c0:c0 + wt] = t + 1
        full = par.gather_tiles_to_root(pano.clone(), (3, 4), root=0)
        if rank == 0:
            for t, (r0, c0, ht, wt) in enumerate(rects):
                assert bool((full[r0:r0 + ht, c0:c0 + wt] == t + 1).all()), t
        red = pano.clone()
        dist.all_reduce(red, op=dist.ReduceOp.MAX)  # the all-ranks form gives the same canvas
        if rank == 0:
            assert torch.equal(red, full)
        # the same with contiguous, area-balanced tile runs (tile_ranges: what the sharded render deals since round 5)
        ranges = par.tile_ranges(H, W, (3, 4), world)
        pano = torch.zeros((H, W, 3), dtype=torch.uint8)
        for t, (r0, c0, ht, wt) in enumerate(rects):
            if ranges[rank][0] <= t < ranges[rank][1]:
                pano[r0:r0 + ht, c0:c0 + wt] = t + 1
        full = par.gather_tiles_to_root(pano.clone(), (3, 4), root=0, ranges=ranges)
        if rank == 0:
            for t, (r0, c0, ht, wt) in enumerate(rects):
                assert bool((full[r0:r0 + ht, c0:c0 + wt] == t + 1).all()), t
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise e


def test_tile_ranges_are_contiguous_cover_the_canvas_and_balance_area():
    sys.path.insert(0, ROOT)
    import apsamd

    par = import_module(apsamd.__name__ + ".parallel")
    for (H, W, tile, ws) in [(11400, 20200, (2048, 2048), 8), (11400, 20200, (2048, 2048), 2), (7, 10, (3, 4), 2), (100, 100, (64, 64), 8),
                             (5000, 50300, (2048, 2048), 8)]:
        rects = par.tile_rects(H, W, tile)
        rg = par.tile_ranges(H, W, tile, ws)
        assert len(rg) == ws and rg[0][0] == 0 and rg[-1][1] == len(rects)
        assert all(rg[r][1] == rg[r + 1][0] for r in range(ws - 1)) and all(b <= e for b, e in rg)
        area = [sum(rects[t][2] * rects[t][3] for t in range(b, e)) for b, e in rg]
        if len(rects) >= 4 * ws:  # enough tiles to balance: no rank above 1.35x the mean (one tile of slack)
            assert max(area) <= 1.35 * (H * W / ws), (H, W, area)
    # the bench canvas on 8 ranks: a rank's tiles lie in at most two tile rows
    rects = par.tile_rects(11400, 20200, (2048, 2048))
    for b, e in par.tile_ranges(11400, 20200, (2048, 2048), 8):
        assert len({rects[t][0] for t in range(b, e)}) <= 2


def test_partition_helpers():
    sys.path.insert(0, ROOT)
    import apsamd

    par = import_module(apsamd.__name__ + ".parallel")
    assert par.shard_indices(10, 4, 1) == [1, 5, 9]
    w = np.array([9, 7, 6, 5, 4, 3, 2, 1.0])
    own = par.partition_weighted(w, 3)
    loads = [w[own == r].sum() for r in range(3)]
    assert max(loads) - min(loads) <= 2 and sorted(set(own.tolist())) == [0, 1, 2]
    assert par.world() == (1, 0)
    t = torch.arange(6).reshape(3, 2)
    assert torch.equal(par.allgather_ragged(t)[0], t)  # world 1: identity


def test_sharding_collectives_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_blocked_pair_partition_is_balanced_and_touches_fewer_images():
    import importlib

    import apsamd

    par = importlib.import_module(apsamd.__name__ + ".parallel")
    fm = importlib.import_module(apsamd.__name__ + ".featureMatching")
    rng = np.random.default_rng(0)
    for n in (9, 64, 130):
        order = fm.pair_order(n)
        cnt = rng.integers(15000, 21000, n)
        w = [float(cnt[i]) * float(cnt[j]) for (i, j) in order]
        for ws in (1, 2, 3, 4, 8):
            own = par.partition_pairs_blocked(order, w, n, ws)
            assert own.shape == (len(order),) and own.min() >= 0 and own.max() < ws
            assert np.array_equal(own, par.partition_pairs_blocked(order, w, n, ws))  # deterministic: every rank agrees
            loads = np.bincount(own, weights=w, minlength=ws)
            if ws > 1 and len(order) >= 20 * ws:
                assert loads.max() <= 1.02 * loads.sum() / ws, (n, ws, loads)
            if n == 64 and ws == 8:
                touched = [len({x for p in np.nonzero(own == r)[0] for x in order[p]}) for r in range(ws)]
                assert max(touched) <= 44 and sum(touched) / ws <= 36, touched  # all 64 with a pair-by-pair deal
    assert par.partition_pairs_blocked([], [], 4, 8).size == 0
    assert np.array_equal(par.partition_pairs_blocked([(0, 1)], [0.0], 2, 4), [0])


# ---- bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run around it) --------------------------
def _bench(*argv, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=e, timeout=300)


def test_bench_gpus_flag_starts_that_many_ranks_and_relays_one_line():
    """--gpus 2 with no WORLD_SIZE: bench.py starts two ranks as children (torch.distributed.run, 127.0.0.1), every rank sees
    WORLD_SIZE = 2 and the flags, rank 0's single JSON line comes back on stdout and nothing else does.  The ranks stop at
    the rank probe (gloo), so no GPU is needed."""
    import json
    r = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", env={"APS_BENCH_RANK_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out == {"probe": True, "n_gpus": 2, "gpus_arg": 2, "steps": 3, "warmup": 1, "rank_sum": 3.0, "master_addr": "127.0.0.1"}


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench("--gpus", "2", env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0", "APS_BENCH_RANK_PROBE": "1"})
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr
    r = _bench("--gpus", "0")
    assert r.returncode == 2


def test_bench_launcher_command_is_the_drivers_form():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cmd = mod.launcher_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], port=29999)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port",
                       "29999", os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"]
