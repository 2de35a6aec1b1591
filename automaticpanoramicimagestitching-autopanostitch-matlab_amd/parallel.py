"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

What shards how:
  1. SIFT            : images are independent (loadImages.m:82-99) -> image i on rank i % world.
  2. exchange        : ONE all-gather of the descriptor blocks (+ a small one for keypoint coordinates and
                       counts) so every rank holds all F x 128 descriptors.
  3. match           : the pair list of featureMatchingPairwise.m:48 is partitioned over ranks, balanced by
                       N_i * N_j; match lists are all-gathered (a few MB).
  4. RANSAC          : candidate pairs (imageMatching.m:121) round-robin over ranks; draws are keyed by the
                       global pair index so the result does not depend on the sharding.
  5. host segment    : every rank repeats the tiny deterministic graph/camera step (no broadcast needed).
  6. render          : source images are all-gathered once (uint8), panorama tiles (independent in the
                       reference, renderPanorama.m:342-406) go to rank t % world, and the canvas is combined
                       with one all-reduce(MAX) over disjoint tiles.
There is no other collective on the data path.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def shard_indices(n, world_size, rank):
    """Image i lives on rank i % world (block-cyclic keeps pixel counts balanced for equal-size images)."""
    return [i for i in range(n) if i % world_size == rank]


def partition_weighted(weights, world_size):
    """Longest-processing-time greedy partition of items with the given costs; returns owner[item]."""
    order = np.argsort(-np.asarray(weights, np.float64), kind="stable")
    load = np.zeros(world_size)
    owner = np.zeros(len(weights), np.int64)
    for it in order:
        r = int(np.argmin(load))
        owner[it] = r
        load[r] += weights[it]
    return owner


def allgather_ragged(local, group=None):
    """All-gather of per-rank tensors whose first dimension differs: one count exchange + ONE padded
    all-gather.  Returns the list of per-rank tensors (on the same device as `local`)."""
    ws, _ = world()
    if ws == 1:
        return [local]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n_local) for _ in range(ws)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((ws * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    out = out.view((ws, m) + tuple(local.shape[1:]))
    return [out[r, : counts[r]] for r in range(ws)]


def gather_by_owner(local_items, owner_of, n, make_tensor, split_sizes_local):
    """Generic helper: every rank holds some of n ragged items; after the call every rank holds all of them.
    local_items: dict idx -> tensor [len_i, ...]; owner_of(idx) -> rank.  Items of one rank are concatenated
    in ascending idx order, exchanged with allgather_ragged, and split again with the all-gathered lengths."""
    ws, rank = world()
    mine = sorted(local_items)
    lens = torch.zeros(n, dtype=torch.int64)
    for i in mine:
        lens[i] = local_items[i].shape[0]
    dev = make_tensor.device
    lens = lens.to(dev)
    if ws > 1:
        dist.all_reduce(lens, op=dist.ReduceOp.SUM)
    lens = lens.cpu().tolist()
    cat = torch.cat([local_items[i] for i in mine]) if mine else make_tensor[:0]
    parts = allgather_ragged(cat)
    out = [None] * n
    for r in range(ws):
        off = 0
        for i in range(n):
            if owner_of(i) == r:
                out[i] = parts[r][off: off + lens[i]]
                off += lens[i]
    return out


def exchange_items(local_ids, local_arrays, n_items, owner_of, width, dtype, dev):
    """Every rank owns some of n_items ragged numpy arrays ([len_i, width]); returns the list of all n_items
    arrays on every rank.  One all-reduce of the lengths + ONE padded all-gather of the concatenated payload
    (no per-item tensors)."""
    ws, rank = world()
    if ws == 1:
        out = [None] * n_items
        for i, a in zip(local_ids, local_arrays):
            out[i] = a
        return out
    lens = torch.zeros(n_items, dtype=torch.int64)
    for i, a in zip(local_ids, local_arrays):
        lens[i] = a.shape[0]
    lens = lens.to(dev)
    dist.all_reduce(lens, op=dist.ReduceOp.SUM)
    lens = lens.cpu().numpy()
    order = np.argsort(np.asarray(local_ids, np.int64), kind="stable") if len(local_ids) else np.zeros(0, np.int64)
    cat = (np.concatenate([local_arrays[k].reshape(-1, width) for k in order]) if len(order)
           else np.zeros((0, width), dtype))
    parts = allgather_ragged(torch.from_numpy(np.ascontiguousarray(cat)).to(dev))
    parts = [p_.cpu().numpy() for p_ in parts]
    out = [None] * n_items
    offs = [0] * ws
    for i in range(n_items):
        r = owner_of(i)
        out[i] = parts[r][offs[r]: offs[r] + lens[i]]
        offs[r] += int(lens[i])
    return out


class ImageGather:
    """The all-gather of the uint8 source images, started early and collected late: the images exist before
    anything is computed, the render needs them at the very end, and the ~1.6 GB (64 x 4K) it moves over xGMI
    would otherwise sit on the critical path between RANSAC and render.  Equal-size images; rank r owns i % world."""

    def __init__(self, local_images, n):
        self.ws, self.rank = world()
        self.n = n
        self.local = local_images
        self.work = None
        if self.ws == 1:
            return
        first = next(iter(local_images.values()))
        self.shape = tuple(first.shape)
        per = (n + self.ws - 1) // self.ws  # slots per rank (the last ones may stay empty)
        flat = first.numel()
        self.send = torch.zeros((per, flat), dtype=torch.uint8, device=first.device)
        for slot, i in enumerate(sorted(local_images)):
            self.send[slot] = local_images[i].reshape(-1)
        self.recv = torch.empty((self.ws * per, flat), dtype=torch.uint8, device=first.device)
        self.per = per
        self.work = dist.all_gather_into_tensor(self.recv, self.send, async_op=True)

    def wait(self):
        if self.ws == 1:
            return [self.local[i] for i in range(self.n)]
        self.work.wait()
        return [self.recv[(i % self.ws) * self.per + i // self.ws].reshape(self.shape) for i in range(self.n)]


def tile_rects(H, W, tile):
    """The tile loop of renderPanorama.m:342-406 / aps_render_tiles: rows outer, columns inner."""
    TH, TW = int(tile[0]), int(tile[1])
    return [(r0, c0, min(TH, H - r0), min(TW, W - c0)) for r0 in range(0, H, TH) for c0 in range(0, W, TW)]


def gather_tiles_to_root(pano, tile, root=0):
    """Rank r holds the tiles t % world == r of `pano` (H x W x C, zero elsewhere).  Instead of an all-reduce of
    the whole canvas, every rank sends exactly its tiles to `root` (xGMI is point to point: the transfers of
    the other ranks run side by side), which copies them into place.  Returns pano (complete on root only)."""
    ws, rank = world()
    if ws == 1:
        return pano
    H, W = int(pano.shape[0]), int(pano.shape[1])
    rects = tile_rects(H, W, tile)
    C_ = int(pano.shape[2])
    sizes = [sum(ht * wt * C_ for t, (_, _, ht, wt) in enumerate(rects) if t % ws == r) for r in range(ws)]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=pano.dtype, device=pano.device)
    off = 0
    for t, (r0, c0, ht, wt) in enumerate(rects):
        if t % ws == rank:
            m = ht * wt * C_
            buf[off:off + m] = pano[r0:r0 + ht, c0:c0 + wt].reshape(-1)
            off += m
    parts = [torch.empty_like(buf) for _ in range(ws)] if rank == root else None
    dist.gather(buf, parts, dst=root)
    if rank == root:
        for r in range(ws):
            if r == root:
                continue
            off = 0
            for t, (r0, c0, ht, wt) in enumerate(rects):
                if t % ws == r:
                    m = ht * wt * C_
                    pano[r0:r0 + ht, c0:c0 + wt] = parts[r][off:off + m].reshape(ht, wt, C_)
                    off += m
    return pano


def stitch_distributed(input, local_images, n, Ks, tile=(2048, 2048), seed=0, cameras=None, pano_root=None):
    """The whole stitch with the work sharded over the ranks of the default process group (see module doc).
    local_images: dict image index -> uint8 H x W x 3 CUDA tensor for the indices shard_indices(n, world, rank).
    pano_root: None = the panorama is combined on every rank (all-reduce); r = only rank r receives the tiles of
    the others (cheaper; the other ranks return their own tiles only).
    Returns (panorama uint8 H x W x 3 CUDA tensor; info dict)."""
    from . import _capi
    from . import featureMatching as fm
    from . import imageMatching as im
    from . import pipeline as pl
    from . import renderPanorama as rp

    ws, rank = world()
    dev = next(iter(local_images.values())).device if local_images else torch.device("cuda")
    times = pl.StageTimes()
    owner = lambda i: i % ws  # noqa: E731

    # 0) the render will need every source image everywhere: start that all-gather now, collect it at step 6
    t0 = time.perf_counter()
    img_gather = ImageGather(local_images, n)
    times.add("exchange", t0)

    # 1) SIFT on the local shard
    t0 = time.perf_counter()
    ldesc, lkps, lkps_host = {}, {}, {}
    for i, (d, p) in zip(sorted(local_images), pl.sift_many(input, [local_images[i] for i in sorted(local_images)])):
        ldesc[i] = d
        lkps_host[i] = p
        lkps[i] = torch.from_numpy(p).to(dev)
    times.add("features", t0)

    # 2) the exchange: descriptors (one all-gather), keypoints (small)
    t0 = time.perf_counter()
    if ws > 1:
        descs = gather_by_owner(ldesc, owner, n, torch.empty((0, 128), dtype=torch.float32, device=dev), None)
        kps_t = gather_by_owner(lkps, owner, n, torch.empty((0, 2), dtype=torch.float64, device=dev), None)
        descs = [d.contiguous() for d in descs]
        torch.cuda.synchronize()  # gathered blocks are produced on RCCL's / torch's streams, consumed on the library's
        kps = [k.cpu().numpy() for k in kps_t]
    else:
        descs = [ldesc[i] for i in range(n)]
        kps_t = [lkps[i] for i in range(n)]
        kps = [lkps_host[i] for i in range(n)]  # the host copies SIFT returned: no read-back
    counts = [int(d.shape[0]) for d in descs]
    times.add("exchange", t0)

    # 3) match: pair list partitioned by N_i * N_j
    t0 = time.perf_counter()
    order = fm.pair_order(n)
    w = [float(counts[i]) * float(counts[j]) for (i, j) in order]
    pown = partition_weighted(w, ws) if ws > 1 else np.zeros(len(order), np.int64)
    my = [p for p in range(len(order)) if pown[p] == rank]
    # one rank: the match lists never leave the device (no exchange, no host index arithmetic); APS_PARALLEL_HOST_LISTS=1
    # forces the multi-rank code path (host lists) for A/B tests
    resident = ws == 1 and not os.environ.get("APS_PARALLEL_HOST_LISTS")
    if resident:
        pp, ia_d, ib_d, _ = fm.match_pairs_csr(descs, [order[p] for p in my], input["Ratiothreshold"],
                                               input["Matchingthreshold"], True, device_out=True)
        n_match = np.diff(pp)  # my == every pair, in `order`
    else:
        pp, ia, ib, _ = fm.match_pairs_csr(descs, [order[p] for p in my], input["Ratiothreshold"], input["Matchingthreshold"], True)
        mine_m = [np.stack([ia[int(pp[k]):int(pp[k + 1])], ib[int(pp[k]):int(pp[k + 1])]], 1).astype(np.int64)
                  for k in range(len(my))]
        allm = exchange_items(my, mine_m, len(order), lambda p: int(pown[p]), 2, np.int64, dev)
        n_match = np.array([len(m) for m in allm], np.int64)
    times.add("matching", t0)

    # 4) candidate selection (redundant, deterministic) + RANSAC sharded round-robin
    t0 = time.perf_counter()
    put = np.zeros((n, n), np.int64)
    for p, (i, j) in enumerate(order):
        put[i, j] = n_match[p]
    sym = put + put.T
    srt = np.argsort(-sym, axis=1, kind="stable")[:, : min(int(input["mBrownLowe"]), n - 1)]
    cand = np.zeros((n, n), bool)
    cand[np.repeat(np.arange(n), srt.shape[1]), srt.reshape(-1)] = True
    cand = np.triu(cand | cand.T, 1)
    pidx = {ij: p for p, ij in enumerate(order)}
    cj, ci = np.nonzero(cand.T)
    work = [pidx[(i, j)] for (i, j) in zip(ci.tolist(), cj.tolist()) if n_match[pidx[(i, j)]] >= 4]
    mine = [p for k, p in enumerate(work) if k % ws == rank]
    times.add("im_select", t0)
    t0 = time.perf_counter()
    n_samples = int(input["maxIter"]) + 64
    recs = []
    if mine:
        cnts = [int(n_match[p]) for p in mine]
        wptr = np.concatenate([[0], np.cumsum(cnts)]).astype(np.int64)
        # one device gather over the concatenated keypoint table (the keypoints are resident already; a host
        # fancy-index of ~1e6 rows costs tens of milliseconds)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        kp_all = torch.cat([k.to(dev) for k in kps_t]).to(torch.float64)
        if resident:
            # segment p of the CSR lists -> rows of the concatenated keypoint table, all on the device
            cnt_t = torch.tensor(cnts, dtype=torch.int64, device=dev)
            seg0 = torch.tensor([int(pp[p]) for p in mine], dtype=torch.int64, device=dev)
            pos = torch.arange(int(wptr[-1]), dtype=torch.int64, device=dev) + \
                torch.repeat_interleave(seg0 - torch.from_numpy(wptr[:-1]).to(dev), cnt_t)
            off_i = torch.repeat_interleave(torch.from_numpy(offs[[order[p][0] for p in mine]]).to(dev), cnt_t)
            off_j = torch.repeat_interleave(torch.from_numpy(offs[[order[p][1] for p in mine]]).to(dev), cnt_t)
            row_i = off_i + ia_d.index_select(0, pos).to(torch.int64) - 1
            row_j = off_j + ib_d.index_select(0, pos).to(torch.int64) - 1
        else:
            cat = np.concatenate([allm[p] for p in mine])
            img_i = np.repeat(offs[[order[p][0] for p in mine]], cnts)
            img_j = np.repeat(offs[[order[p][1] for p in mine]], cnts)
            idx = torch.from_numpy(np.stack([img_i + cat[:, 0] - 1, img_j + cat[:, 1] - 1])).to(dev)
            row_i, row_j = idx[0], idx[1]
        dst = kp_all.index_select(0, row_i).t().contiguous()
        src = kp_all.index_select(0, row_j).t().contiguous()
        torch.cuda.current_stream().synchronize()  # torch's stream produced dst/src; the library runs on its own
        samples = im.draw_samples_device(cnts, n_samples, seed, keys=mine)
        times.add("im_gather", t0)
        t0 = time.perf_counter()
        models, mask, found, ninl = im.ransac_batch(src, dst, wptr, samples, input)
        times.add("im_ransac", t0)
        t0 = time.perf_counter()
        for k in range(len(mine)):
            rec = np.empty(11 + cnts[k])
            rec[:9] = models[k].reshape(-1)
            rec[9] = found[k]
            rec[10] = ninl[k]
            rec[11:] = mask[wptr[k]:wptr[k + 1]]
            recs.append(rec.reshape(-1, 1))
    wk = {p: k for k, p in enumerate(work)}
    allr = exchange_items([wk[p] for p in mine], recs, len(work), lambda k: k % ws, 1, np.float64, dev)
    allr = [r.reshape(-1) for r in allr]
    pairs, models_l, num_matches = [], [], np.zeros((n, n))
    for k, p in enumerate(work):
        rec = allr[k]
        nf = int(n_match[p])
        ni = int(rec[10]) if rec[9] else 0
        if ni > 8 + 0.3 * nf:
            i, j = order[p]
            pairs.append((i, j))
            models_l.append(rec[:9].reshape(3, 3))
            num_matches[i, j] = ni
    times.add("im_merge", t0)

    # 5) host segment (redundant on every rank)
    t0 = time.perf_counter()
    ncomp, labels = pl.connected_components(num_matches)
    if cameras is None:
        cameras, ref = pl.cameras_from_models(n, pairs, models_l, num_matches, Ks)
        cameras = pl.straightening(cameras)
    else:
        ref = int(np.argmax((num_matches + num_matches.T).sum(1)))
    times.add("host_cameras", t0)

    # 6) render: all images everywhere, tiles t % world == rank, one all-reduce(MAX) of the canvas
    t0 = time.perf_counter()
    images = img_gather.wait()
    if ws > 1:
        torch.cuda.synchronize()  # the collective ran on RCCL's stream; the library reads the images on its own
    times.add("exchange", t0)
    t0 = time.perf_counter()
    comp = labels[ref]
    members = [k for k in range(n) if labels[k] == comp and cameras[k] is not None]
    sizes = [(int(images[k].shape[0]), int(images[k].shape[1]), 3) for k in members]
    opts = {"anglePower": 2, "blending": input["blending"], "pyrLevels": input["bands"], "pyrSigma": input["MBBsigma"],
            "canvasColor": input["canvasColor"], "tile": tile, "cropBorder": False}
    gains = None
    tg = time.perf_counter()
    if input.get("gainCompensation"):
        # gainCompensationRKf between cameras and render (renderPanorama.m:303-330).  The device sums in an
        # unspecified order, so rank 0's gains are broadcast: every tile must be rendered with the same numbers.
        from . import gainCompensation as gc

        mem_cams = [cameras[k] for k in members]
        o_ = rp.default_opts(opts, mem_cams, members.index(ref))
        geo = rp.canvas_geometry(mem_cams, sizes, input["panorama2DisplaynSave"], members.index(ref), o_)
        gains = gc.gainCompensationRKf([images[k] for k in members], mem_cams, input["panorama2DisplaynSave"],
                                       members.index(ref), input, geo)
        if ws > 1:
            gt_ = torch.from_numpy(np.ascontiguousarray(gains)).to(dev)
            dist.broadcast(gt_, 0)
            gains = gt_.cpu().numpy()
        times.add("gain_compensation", tg)
        t0 = time.perf_counter()
    pano, _ = rp.renderPanorama(input, [images[k] for k in members], sizes, [cameras[k] for k in members],
                                input["panorama2DisplaynSave"], members.index(ref), opts, gains=gains, device_out=True,
                                tile_subset=(rank, ws) if ws > 1 else None)
    pl._sync()
    if ws > 1:
        torch.cuda.synchronize()
        if pano_root is None:
            dist.all_reduce(pano, op=dist.ReduceOp.MAX)  # disjoint tiles, zero elsewhere
        else:
            pano = gather_tiles_to_root(pano, tile, pano_root)
        torch.cuda.synchronize()
    times.add("render", t0)
    info = {"times": dict(times), "n_features": counts, "n_pairs_verified": len(pairs), "n_components": int(ncomp),
            "panorama_shape": tuple(int(v) for v in pano.shape), "members": members, "cameras": cameras, "pairs": pairs, "models": models_l}
    return pano, info
