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
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
        raise e


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
