"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for the tests, and on device tensors through host staging - see _staged()).

What shards how:
  1. SIFT            : images are independent (loadImages.m:82-99) -> image i on rank i % world.
  2. exchange        : ONE all-gather of the descriptor blocks (+ a small one for keypoint coordinates and
                       counts) so every rank holds all F x 128 descriptors.
  3. match           : the pair list of featureMatchingPairwise.m:48 is partitioned over ranks, balanced by
                       N_i * N_j; the match index lists are all-gathered on the device (a few MB) and stay there.
  4. RANSAC          : candidate pairs (imageMatching.m:121) round-robin over ranks; draws are keyed by the
                       global pair index so the result does not depend on the sharding; the fixed-size verdicts
                       (model, found, inlier count) are combined with one all-reduce.
  5. host segment    : every rank repeats the tiny deterministic graph/camera step (no broadcast needed):
                       connected components, one camera set + reference per component (recognizePanoramas.m).
  6. render          : source images are all-gathered once (uint8, started before SIFT); EVERY connected component
                       is rendered (displayPanorama.m:88-116): components first (whole panoramas to ranks, balanced
                       by canvas area, when there are at least as many as ranks), tiles second (tiles t % world of
                       one canvas, renderPanorama.m:342-406, when there are fewer).
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


def _multi(ws):
    """True when the multi-rank code path has to run.  APS_PARALLEL_FORCE_COLLECTIVES=1 (a test hook) takes that path with a
    ONE-rank process group too: every collective, hand-over and stream wait of the N > 1 driver then runs on the real
    backend (RCCL refuses two ranks on one GPU, so this is how a one-GPU box exercises the nccl calls)."""
    return ws > 1 or (os.environ.get("APS_PARALLEL_FORCE_COLLECTIVES") == "1" and dist.is_available() and dist.is_initialized())


def _staged(t=None):
    """gloo moves host memory: collectives on device tensors are staged through the host (used by the 2-rank
    equality test, which runs both ranks on one GPU; production is "nccl" = RCCL, device to device)."""
    return dist.get_backend() == "gloo" and (t is None or t.is_cuda)


def _all_gather_into(out, inp, async_op=False):
    if _staged(inp):
        ho = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(ho, inp.cpu())
        out.copy_(ho)
        return None
    return dist.all_gather_into_tensor(out, inp, async_op=async_op)


def _all_reduce(t, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def _broadcast(t, src):
    if _staged(t):
        h = t.cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


def _gather(t, parts, dst):
    if _staged(t):
        hp = [torch.empty(t.shape, dtype=t.dtype) for _ in parts] if parts is not None else None
        dist.gather(t.cpu(), hp, dst=dst)
        if parts is not None:
            for a, b in zip(parts, hp):
                a.copy_(b)
    else:
        dist.gather(t, parts, dst=dst)


def _is_cuda_tensor(x):
    return isinstance(x, torch.Tensor) and x.is_cuda


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


def partition_pairs_blocked(pairs, weights, n, world_size, scatter=False):
    """Owner of every image pair for the matching stage.  The matcher prepares every descriptor set a rank's pairs touch
    (a fixed cost per image and rank), so the pairs are dealt in BLOCKS of the pair matrix instead of one by one: images
    are cut into ceil(sqrt(2 * world)) groups, the pairs are walked group-pair by group-pair (stable inside one), and the
    walk is cut into `world` contiguous segments of equal total weight.  A rank then touches the images of one or two
    group pairs (at 8 ranks and 64 views: ~32-48 of them instead of all 64); the balance is that of a prefix-sum cut: within
    one pair's weight of even.  The result depends on (pairs, weights, n, world) only - every rank computes the same.
    scatter (off): the groups are drawn by a fixed pseudo-random permutation of the images instead of by index ranges.
    The weights N_i N_j are what the proof pass costs; the survivor pass costs what the pairs OVERLAP, which nobody knows
    yet - but image sets come in capture order, neighbours in index overlap, and index-range groups put most overlapping
    pairs into the diagonal blocks.  Measured on the 64 x 4K scene at 8 ranks (scripts/probe/probe_rank_costs.py,
    profiles/r04h_rank_costs.txt): overlapping pairs per rank 33-97 -> 55-79, slowest rank 13.2 -> 12.9 ms, descriptor
    sets per rank 25-42 -> 28-48: the proof pass is 2/3 of a rank's matching whatever its pairs overlap, so the spread of
    the survivor pass is worth 2 % and costs preparation; index ranges stay the default."""
    P = len(pairs)
    owner = np.zeros(P, np.int64)
    if world_size <= 1 or P == 0:
        return owner
    nb = int(np.ceil(np.sqrt(2.0 * world_size)))
    B = max(1, -(-n // nb))
    arr = np.asarray(pairs, np.int64).reshape(P, 2)
    if scatter:
        place = np.empty(n, np.int64)
        place[np.random.RandomState(20240 + n).permutation(n)] = np.arange(n)
        ga, gb = place[arr[:, 0]] // B, place[arr[:, 1]] // B
    else:
        ga, gb = arr[:, 0] // B, arr[:, 1] // B
    key = np.minimum(ga, gb) * (nb + 1) + np.maximum(ga, gb)
    walk = np.argsort(key, kind="stable")
    w = np.asarray(weights, np.float64)[walk]
    total = float(w.sum())
    if total <= 0:
        owner[walk] = np.arange(P) * world_size // P
        return owner
    mid = np.cumsum(w) - 0.5 * w  # a pair belongs to the segment its midpoint falls into
    owner[walk] = np.minimum((mid * world_size / total).astype(np.int64), world_size - 1)
    return owner


def allgather_ragged(local, group=None):
    """All-gather of per-rank tensors whose first dimension differs: one count exchange + ONE padded
    all-gather.  Returns the list of per-rank tensors (on the same device as `local`)."""
    ws, _ = world()
    if not _multi(ws):
        return [local]
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    cnt_all = torch.zeros(ws, dtype=torch.int64, device=local.device)
    _all_gather_into(cnt_all, n_local)
    counts = [int(c) for c in cnt_all.cpu().tolist()]
    m = max(max(counts), 1)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((ws * m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    _all_gather_into(out, pad)
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
    if _multi(ws):
        _all_reduce(lens)
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
    if not _multi(ws):
        out = [None] * n_items
        for i, a in zip(local_ids, local_arrays):
            out[i] = a
        return out
    lens = torch.zeros(n_items, dtype=torch.int64)
    for i, a in zip(local_ids, local_arrays):
        lens[i] = a.shape[0]
    lens = lens.to(dev)
    _all_reduce(lens)
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
    would otherwise sit on the critical path between RANSAC and render.  Rank r owns the images i % world == r;
    sizes may differ and a rank may own none (n < world): a small all-reduce of the shapes comes first."""

    def __init__(self, local_images, n, dev=None):
        self.ws, self.rank = world()
        self.n = n
        self.local = local_images
        self.work = None
        if not _multi(self.ws):
            return
        if dev is None:
            dev = next(iter(local_images.values())).device if local_images else torch.device("cuda")
        shp = torch.zeros((n, 3), dtype=torch.int64, device=dev)
        for i, im_ in local_images.items():
            shp[i] = torch.tensor([int(im_.shape[0]), int(im_.shape[1]), int(im_.shape[2]) if im_.dim() == 3 else 1],
                                  dtype=torch.int64, device=dev)
        _all_reduce(shp)
        self.shapes = [tuple(int(v) for v in row) for row in shp.cpu().tolist()]
        per = (n + self.ws - 1) // self.ws  # slots per rank (the last ones may stay empty)
        flat = max(max(h * w * c for (h, w, c) in self.shapes), 1)
        self.send = torch.empty((per, flat), dtype=torch.uint8, device=dev)  # (padding is never read: the shapes are known)
        for slot, i in enumerate(sorted(local_images)):
            v = local_images[i].reshape(-1)
            self.send[slot, : v.numel()] = v
        self.recv = torch.empty((self.ws * per, flat), dtype=torch.uint8, device=dev)
        self.per = per
        self.work = _all_gather_into(self.recv, self.send, async_op=True)

    def finish(self):
        """Blocks until the collective has completed (idempotent).  Called before the matching stage: the int8 screening
        kernel claims whole CUs (match_screen_i8_kernel), so a collective left running beside it would only time-slice
        against it - and nothing may share a SIMD with its waves."""
        if _multi(self.ws) and self.work is not None:
            self.work.wait()
            self.work = None
            if self.recv.is_cuda:
                torch.cuda.current_stream().synchronize()

    def wait(self):
        if not _multi(self.ws):
            return [self.local[i] for i in range(self.n)]
        self.finish()
        out = []
        for i in range(self.n):
            h, w, c = self.shapes[i]
            out.append(self.recv[(i % self.ws) * self.per + i // self.ws, : h * w * c].reshape(h, w, c))
        return out


class FeatureExchange:
    """The descriptor exchange in chunks that overlap the extraction: image i lives on rank
    i % world in slot i // world; the slots are cut into `rounds` chunks, and as soon as a rank's images of a chunk are
    extracted the chunk is all-gathered (descriptors: one padded collective, keypoints: a second, small one) while the
    worker streams go on with the next chunk.  Per chunk the only host visit is the count exchange (one small
    all-gather), which the main thread waits for while the GPU keeps extracting.  Every rank runs the same rounds, also
    those in which it owns no image.
    futures: dict image index -> future resolving to (descriptors [k,128] float32 tensor, keypoints [k,2] float64
    numpy or tensor); dev: device of the exchanged tensors."""

    def __init__(self, futures, n, dev, rounds=None):
        self.ws, self.rank = world()
        self.n, self.dev = n, dev
        self.per = (n + self.ws - 1) // self.ws
        if rounds is None:
            # a rank's images are extracted APS_SIFT_WORKERS at a time: with no more images than workers they all finish
            # together and chunks have nothing to hide behind - one exchange then (8 ranks x 8 views), else one chunk per
            # batch of workers, four at most (2 ranks x 32 views)
            workers = max(1, int(os.environ.get("APS_SIFT_WORKERS", 10)))
            rounds = min(4, (self.per + workers - 1) // workers)
        rounds = max(1, min(int(os.environ.get("APS_EXCHANGE_ROUNDS", rounds)), self.per))
        self.chunk = (self.per + rounds - 1) // rounds
        self.rounds = (self.per + self.chunk - 1) // self.chunk
        self.futures = futures
        self.parts = []  # per round: (slot0, nslot, counts [ws, nslot], desc buffer, kps buffer, handles)

    def run(self):
        """Issues every round (blocking on the local futures of each in turn); returns self."""
        ws, rank, C = self.ws, self.rank, self.chunk
        for r in range(self.rounds):
            s0 = r * C
            ns = min(C, self.per - s0)
            mine = [(s, s * ws + rank) for s in range(s0, s0 + ns) if s * ws + rank < self.n]
            got = {s: self.futures[i].result() for (s, i) in mine}
            cnt = torch.zeros(ns, dtype=torch.int64)
            for s, (d, _) in got.items():
                cnt[s - s0] = int(d.shape[0])
            cnt = cnt.to(self.dev)
            cnt_all = torch.empty(ws * ns, dtype=torch.int64, device=self.dev)
            _all_gather_into(cnt_all, cnt)
            counts = cnt_all.cpu().view(ws, ns)  # the one host visit of the round
            m = max(int(counts.max()), 1)
            send_d = torch.empty((ns, m, 128), dtype=torch.float32, device=self.dev)  # (rows past a slot's count are never read)
            send_k = torch.empty((ns, m, 2), dtype=torch.float64, device=self.dev)
            for s, (d, p) in got.items():
                k = int(d.shape[0])
                send_d[s - s0, :k] = d
                send_k[s - s0, :k] = p if torch.is_tensor(p) else torch.from_numpy(np.ascontiguousarray(p))
            # (the futures resolve after their worker's stream has drained; the packing above and the collectives below are
            # ordered by torch's current stream)
            recv_d = torch.empty((ws, ns, m, 128), dtype=torch.float32, device=self.dev)
            recv_k = torch.empty((ws, ns, m, 2), dtype=torch.float64, device=self.dev)
            h1 = _all_gather_into(recv_d.view(ws * ns, m, 128), send_d, async_op=True)
            h2 = _all_gather_into(recv_k.view(ws * ns, m, 2), send_k, async_op=True)
            self.parts.append((s0, ns, counts, recv_d, recv_k, (h1, h2), (send_d, send_k)))
        return self

    def wait(self):
        """(descriptors, keypoints) of all n images on every rank, as views into the gathered chunk buffers."""
        descs, kps = [None] * self.n, [None] * self.n
        for (s0, ns, counts, recv_d, recv_k, handles, _keep) in self.parts:
            for hnd in handles:
                if hnd is not None:
                    hnd.wait()
            for r in range(self.ws):
                for q in range(ns):
                    i = (s0 + q) * self.ws + r
                    if i < self.n:
                        k = int(counts[r, q])
                        descs[i] = recv_d[r, q, :k]
                        kps[i] = recv_k[r, q, :k]
        return descs, kps


def tile_rects(H, W, tile):
    """The tile loop of renderPanorama.m:342-406 / aps_render_tiles: rows outer, columns inner."""
    TH, TW = int(tile[0]), int(tile[1])
    return [(r0, c0, min(TH, H - r0), min(TW, W - c0)) for r0 in range(0, H, TH) for c0 in range(0, W, TW)]


def tile_ranges(H, W, tile, world_size):
    """Contiguous runs of the row-major tile list, one per rank, cut where the cumulative tile AREA passes k / world of
    the canvas (the last tile row and column are partial): [(begin, end)] * world.  A rank's tiles are neighbours, so its
    render converts only the views under its band of the canvas (64 x 4K on 8 ranks: ~20 of 64, where tiles dealt
    t % world met 47 - profiles/r04b_rank_costs.txt) and its output is ~one rectangle."""
    rects = tile_rects(H, W, tile)
    area = np.cumsum([ht * wt for (_, _, ht, wt) in rects], dtype=np.float64)
    cuts = [0]
    for k in range(1, world_size):
        # first tile whose cumulative area reaches k / world of the total; never before the previous cut
        cuts.append(max(cuts[-1], int(np.searchsorted(area, area[-1] * k / world_size, side="left")) + 1 if len(rects) else 0))
    cuts.append(len(rects))
    cuts = [min(c, len(rects)) for c in cuts]
    return [(cuts[r], max(cuts[r], cuts[r + 1])) for r in range(world_size)]


def gather_tiles_to_root(pano, tile, root=0, ranges=None):
    """Rank r holds the tiles t % world == r of `pano` (H x W x C, zero elsewhere) - or, with `ranges` (tile_ranges), the
    tiles ranges[r][0] <= t < ranges[r][1].  Instead of an all-reduce of the whole canvas, every rank sends exactly its
    tiles to `root` (xGMI is point to point: the transfers of the other ranks run side by side), which copies them into
    place.  Returns pano (complete on root only)."""
    ws, rank = world()
    if not _multi(ws):
        return pano
    H, W = int(pano.shape[0]), int(pano.shape[1])
    rects = tile_rects(H, W, tile)
    C_ = int(pano.shape[2])
    if ranges is not None:
        owner = np.full(len(rects), -1, np.int64)
        for r, (b, e) in enumerate(ranges):
            owner[b:e] = r
        assert (owner >= 0).all(), "tile ranges must cover the canvas"
    else:
        owner = np.arange(len(rects)) % ws
    sizes = [sum(ht * wt * C_ for t, (_, _, ht, wt) in enumerate(rects) if owner[t] == r) for r in range(ws)]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=pano.dtype, device=pano.device)
    off = 0
    for t, (r0, c0, ht, wt) in enumerate(rects):
        if owner[t] == rank:
            m = ht * wt * C_
            buf[off:off + m] = pano[r0:r0 + ht, c0:c0 + wt].reshape(-1)
            off += m
    parts = [torch.empty_like(buf) for _ in range(ws)] if rank == root else None
    _gather(buf, parts, root)
    if rank == root:
        for r in range(ws):
            if r == root:
                continue
            off = 0
            for t, (r0, c0, ht, wt) in enumerate(rects):
                if owner[t] == r:
                    m = ht * wt * C_
                    pano[r0:r0 + ht, c0:c0 + wt] = parts[r][off:off + m].reshape(ht, wt, C_)
                    off += m
    return pano


def submit_features(input, local_images, image_events=None, first=None):
    """Starts the feature extraction of this rank's images on the worker streams and returns at once: the handle that
    stitch_distributed(features=...) accepts instead of extracting itself.  What a loop that stitches set after set calls
    for the NEXT set from the current call's after_matching hook (round 6): from there on the current stitch runs RANSAC
    (three latency-bound launches), replicated host work and the bandwidth-bound render - the extraction of the next set
    fills the gaps between them and shares the memory system with the render (it may NOT run beside the int8 matching
    kernels: they keep their SIMDs to themselves, and the matcher opens with a bandwidth-bound preparation).
    local_images / image_events: as for stitch_distributed.
    first: submit only the first `first` images now (a few worker streams fill the idle chip without crowding the latency-bound
    RANSAC launches of the current set); submit_features_rest(handle) submits the others (e.g. from the after_ransac hook)."""
    from . import pipeline as pl

    ws, _ = world()
    mine_img = sorted(local_images)
    if not mine_img:
        return {"images": [], "futures": [], "resident": False, "rest": None}
    dev = local_images[mine_img[0]].device
    resident = (not _multi(ws)) and dev.type == "cuda" and all(_is_cuda_tensor(local_images[i]) for i in mine_img)
    k = len(mine_img) if first is None else max(0, min(int(first), len(mine_img)))

    def submit(idx):
        ready = [image_events[i] for i in idx] if image_events is not None else None
        return pl.sift_submit(input, [local_images[i] for i in idx], ready=ready, points_device=resident) if idx else []

    handle = {"images": mine_img, "futures": submit(mine_img[:k]), "resident": resident, "rest": None}
    if k < len(mine_img):
        handle["rest"] = lambda: submit(mine_img[k:])
    return handle


def submit_features_rest(handle):
    """Submits the images submit_features(first=...) left out; a handle without any is left alone."""
    if handle is not None and handle.get("rest") is not None:
        rest, handle["rest"] = handle["rest"], None
        handle["futures"] = list(handle["futures"]) + list(rest())
    return handle


def _match_pass(input, local_images, n, seed, times, dev, image_events=None, before_match=None, after_features=None,
                features=None, after_matching=None, after_ransac=None):
    """Steps 1-4 on the current images: SIFT on the local shard, the descriptor exchange, the sharded pair matching
    and the sharded RANSAC verification (main.m:88-107 up to imageMatching).  Everything that is exchanged stays on
    the device; the host sees counts, candidate lists and the 3 x 3 models.
    features: the handle of submit_features() for exactly these images (extraction already under way), or None.
    after_matching: optional callable run when the match lists are complete, before candidate selection and RANSAC.
    Returns dict(counts, kps_t, pairs, models, num_matches, n_match, order)."""
    from . import featureMatching as fm
    from . import imageMatching as im
    from . import pipeline as pl

    ws, rank = world()
    owner = lambda i: i % ws  # noqa: E731

    submit_features_rest(features)  # (a handle whose second part was never submitted)
    use_global = not int(input.get("matchFeaturesPairwise", 1))  # main.m:95-99
    host_lists = bool(os.environ.get("APS_PARALLEL_HOST_LISTS"))  # A/B switch: lists through the host (round-1 path)
    order = fm.pair_order(n)
    mine_img = sorted(local_images)
    ready = [image_events[i] for i in mine_img] if image_events is not None else None
    # One rank, optional (APS_MATCH_OVERLAP_CHUNK=<images per chunk>, default off): feature extraction and matching overlap.
    # The pair order is j-major (featureMatchingPairwise.m:48), so the pairs whose later image lies in a chunk of images form
    # one contiguous run of it: as soon as a chunk is extracted, that run is matched on this thread's stream while the
    # worker streams go on with the next images.  The match lists are the same, only earlier.  Measured on the 64 x 4K
    # scene it buys nothing (236.5 ms against 236.8; chunks of 8: 263): the screening kernel has to keep its SIMDs to
    # itself (see match_screen_i8_kernel), so the overlap is time slicing at CU granularity, paid for with repeated
    # descriptor preparation.  Kept as a switch for scenes with fewer, larger images.
    chunk = int(os.environ.get("APS_MATCH_OVERLAP_CHUNK", "0"))
    if not _multi(ws) and not use_global and not host_lists and chunk > 0 and n > chunk and len(mine_img) == n:
        t0 = time.perf_counter()
        futs = pl.sift_submit(input, [local_images[i] for i in range(n)], ready=ready)
        descs, kps_t, pps, ias, ibs = [], [], [np.zeros(1, np.int64)], [], []
        t_match = 0.0
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            for i in range(lo, hi):
                d, p = futs[i].result()
                descs.append(d)
                kps_t.append(torch.from_numpy(p).to(dev))
            if hi == n:
                times.add("features", t0)  # wall time until the last image is extracted (part of the matching ran under it)
                t0 = time.perf_counter()
            p0, p1 = lo * (lo - 1) // 2 if lo else 0, hi * (hi - 1) // 2
            if p1 > p0:
                pp_k, ia_k, ib_k, _ = fm.match_pairs_csr(descs, order[p0:p1], input["Ratiothreshold"], input["Matchingthreshold"],
                                                         True, device_out=True)
                pps.append(pp_k[1:] + pps[-1][-1])
                ias.append(ia_k)
                ibs.append(ib_k)
        counts = [int(d.shape[0]) for d in descs]
        pp = np.concatenate(pps).astype(np.int64)
        ia_d, ib_d = torch.cat(ias), torch.cat(ibs)
        my = list(range(len(order)))
        pown = np.zeros(len(order), np.int64)
        times.add("exchange", time.perf_counter())
    else:
        # 1) SIFT on the local shard; 2) the exchange of descriptors and keypoints.  With more than one rank the exchange
        # runs in chunks BESIDE the extraction (FeatureExchange): a chunk is all-gathered while the worker streams
        # extract the next one, so only the last chunk's collective is left on the critical path.
        t0 = time.perf_counter()
        if _multi(ws):
            if features is not None:
                assert features["images"] == mine_img, "submit_features() was called for other images"
                futs = dict(zip(mine_img, features["futures"]))
            else:
                futs = dict(zip(mine_img, pl.sift_submit(input, [local_images[i] for i in mine_img], ready=ready))) if mine_img else {}
            ex = FeatureExchange(futs, n, dev).run()
            times.add("features", t0)  # (the last local image is extracted; all but the last chunk have been sent)
            t0 = time.perf_counter()
            descs, kps_t = ex.wait()
            if before_match is not None:
                before_match()  # e.g. the early image all-gather: no collective may run beside the matching kernels
            if dev.type == "cuda":
                torch.cuda.synchronize()  # gathered blocks are produced on RCCL's / torch's streams, consumed on the library's
        else:
            ldesc, lkps = {}, {}
            if mine_img:
                # resident images: the keypoints never leave the device (the host copy and the 64 small uploads after the last
                # image were 2 ms of every step); host images: uploaded as soon as their image is done
                resident = dev.type == "cuda" and all(_is_cuda_tensor(local_images[i]) for i in mine_img)
                if features is not None:
                    assert features["images"] == mine_img and features["resident"] == resident, "submit_features() was called for other images"
                    futs = features["futures"]
                else:
                    futs = pl.sift_submit(input, [local_images[i] for i in mine_img], ready=ready, points_device=resident)
                for i, f in zip(mine_img, futs):
                    d, p = f.result()
                    ldesc[i] = d
                    lkps[i] = p if resident else torch.from_numpy(p).to(dev)
            times.add("features", t0)
            t0 = time.perf_counter()
            descs = [ldesc[i] for i in range(n)]
            kps_t = [lkps[i] for i in range(n)]
        counts = [int(d.shape[0]) for d in descs]
        times.add("exchange", t0)
        if after_features is not None:
            after_features()  # caller's hook: the worker streams are idle from here until the next extraction

        # 3) match: pair list partitioned by N_i * N_j; the index lists stay on the device
        t0 = time.perf_counter()
        if _multi(ws):
            w = [float(counts[i]) * float(counts[j]) for (i, j) in order]
            pown = partition_pairs_blocked(order, w, n, ws)
            my = [p for p in range(len(order)) if pown[p] == rank]
            my_pairs = [order[p] for p in my]
        else:  # (one rank: every pair, handed over as the cached array - no per-step walk over 2016 tuples)
            pown = np.zeros(len(order), np.int64)
            my = slice(None)
            my_pairs = fm.pair_order_array(n)
        if use_global:
            # featureMatchingGlobal (the reference's default, inputs.m:46): pooled exact k-NN + per-query filter.  Every rank
            # holds all descriptors after the exchange and computes the (deterministic) result itself: no further exchange.
            pp, ia_d, ib_d = fm.match_global_csr(descs, input["Ratiothreshold"], int(input.get("k", 4)), device_out=True)
            my = slice(None)
        else:
            pp, ia_d, ib_d, _ = fm.match_pairs_csr(descs, my_pairs, input["Ratiothreshold"],
                                                   input["Matchingthreshold"], True, device_out=not host_lists)
    if host_lists and not use_global:
        ia_d = torch.from_numpy(ia_d.astype(np.int32)).to(dev)
        ib_d = torch.from_numpy(ib_d.astype(np.int32)).to(dev)
    n_match = np.zeros(len(order), np.int64)
    n_match[my] = np.diff(pp)
    if _multi(ws) and not use_global:
        nm = torch.from_numpy(n_match).to(dev)
        _all_reduce(nm)
        n_match = nm.cpu().numpy()
        _sync_lib()
        parts = allgather_ragged(torch.stack([ia_d, ib_d], 1).contiguous())
        base = np.concatenate([[0], np.cumsum([int(p_.shape[0]) for p_ in parts])])
        allidx = torch.cat(parts)
        ia_d, ib_d = allidx[:, 0], allidx[:, 1]
        # start of pair p in the concatenated lists: rank base + the lengths of that rank's earlier pairs
        gpos = np.zeros(len(order), np.int64)
        for r in range(ws):
            sel = np.nonzero(pown == r)[0]
            if sel.size:
                gpos[sel] = int(base[r]) + np.concatenate([[0], np.cumsum(n_match[sel])[:-1]])
    else:
        gpos = pp[:-1].astype(np.int64)
    times.add("matching", t0)
    if after_matching is not None:
        after_matching()  # caller's hook: no int8 kernel runs from here on (e.g. submit_features() for the next set)

    # 4) candidate selection (redundant, deterministic) + RANSAC sharded round-robin
    t0 = time.perf_counter()
    oi, oj = _pair_index_arrays(n)
    put = np.zeros((n, n), np.int64)
    put[oi, oj] = n_match
    sym = put + put.T
    srt = np.argsort(-sym, axis=1, kind="stable")[:, : min(int(input["mBrownLowe"]), n - 1)]
    cand = np.zeros((n, n), bool)
    cand[np.repeat(np.arange(n), srt.shape[1]), srt.reshape(-1)] = True
    cand = np.triu(cand | cand.T, 1)
    cj, ci = np.nonzero(cand.T)  # column-major walk of the candidate matrix, like the reference's find
    pw = cj * (cj - 1) // 2 + ci  # position of pair (i, j), i < j, in the j-major pair order
    work = pw[n_match[pw] >= 4].tolist()
    mine = [p for k, p in enumerate(work) if k % ws == rank]
    times.add("im_select", t0)
    t0 = time.perf_counter()
    # model (9), found, inliers per candidate pair: a device tensor only when it has to be all-reduced
    rec = (torch.zeros((max(len(work), 1), 11), dtype=torch.float64, device=dev) if _multi(ws)
           else np.zeros((max(len(work), 1), 11), np.float64))
    if mine:
        cnts = [int(n_match[p]) for p in mine]
        wptr = np.concatenate([[0], np.cumsum(cnts)]).astype(np.int64)
        # imageMatching.m:121-135 (keypoints{i}(matches(:,1),:), keypoints{j}(matches(:,2),:)) as one library launch over
        # the resident keypoint tables and match lists (a host fancy-index of ~1e6 rows costs tens of milliseconds)
        dst, src = im.gather_match_points(kps_t, ia_d, ib_d, gpos[mine], wptr, [order[p][0] for p in mine],
                                          [order[p][1] for p in mine])
        times.add("im_gather", t0)
        t0 = time.perf_counter()
        models, mask, found, ninl = im.ransac_batch_drawn(src, dst, wptr, cnts, input, seed, keys=mine)
        times.add("im_ransac", t0)
        if after_ransac is not None:
            after_ransac()
            after_ransac = None
        t0 = time.perf_counter()
        wk = {p: k for k, p in enumerate(work)}
        vals = np.concatenate([models.reshape(len(mine), 9), found.reshape(-1, 1).astype(np.float64),
                               ninl.reshape(-1, 1).astype(np.float64)], axis=1)
        if _multi(ws):
            rows = torch.tensor([wk[p] for p in mine], dtype=torch.int64, device=dev)
            rec[rows] = torch.from_numpy(vals).to(dev)
        else:
            rec[[wk[p] for p in mine]] = vals
    if after_ransac is not None:
        after_ransac()  # (a rank without candidate pairs of its own)
    if _multi(ws):
        _all_reduce(rec)  # every row is written by exactly one rank, zero elsewhere
        rec = rec.cpu().numpy()
    pairs, models_l, num_matches = [], [], np.zeros((n, n))
    if work:
        wk_ = np.asarray(work, np.int64)
        nf = n_match[wk_]
        ni = np.where(rec[: len(work), 9] != 0, rec[: len(work), 10].astype(np.int64), 0)
        for k in np.nonzero(ni > 8 + 0.3 * nf)[0].tolist():  # imageMatching.m:150
            i, j = order[work[k]]
            pairs.append((i, j))
            models_l.append(rec[k, :9].reshape(3, 3).copy())
            num_matches[i, j] = ni[k]
    times.add("im_merge", t0)
    return {"counts": counts, "kps_t": kps_t, "pairs": pairs, "models": models_l, "num_matches": num_matches,
            "n_match": n_match, "order": order}


_PAIR_INDEX = {}


def _pair_index_arrays(n):
    """(i, j) of every pair of fm.pair_order(n) as two index arrays (pair (i, j), i < j, sits at j (j - 1) / 2 + i)."""
    if n not in _PAIR_INDEX:
        jj = np.repeat(np.arange(1, n), np.arange(1, n))
        ii = np.concatenate([np.arange(j) for j in range(1, n)]) if n > 1 else np.zeros(0, np.int64)
        _PAIR_INDEX[n] = (ii.astype(np.int64), jj.astype(np.int64))
    return _PAIR_INDEX[n]


def _sync_lib():
    from . import pipeline as pl

    pl._sync()


def stitch_distributed(input, local_images, n, Ks, tile=(2048, 2048), seed=0, cameras=None, pano_root=None,
                       local_originals=None, image_events=None, after_features=None, features=None, after_matching=None,
                       after_ransac=None):
    """The whole stitch with the work sharded over the ranks of the default process group (see module doc).
    local_images: dict image index -> uint8 H x W x 3 CUDA tensor for the indices shard_indices(n, world, rank).
    pano_root: None = every panorama is combined on every rank; r = only rank r receives the tiles / panoramas of the
    others (cheaper; the other ranks return what they rendered themselves).
    local_originals: the unresized originals of the local images; with input.resizeImage and
    input.resizeImagePanoramaCluster set and more than one connected component they are resized per component and the
    extract -> match -> verify chain runs a second time (imageMatchingPanoramaConComps.m:48-91).
    image_events: dict image index -> torch CUDA event that marks the local image as uploaded (end-to-end runs issue
    the host-to-device copies on a side stream; SIFT of image k then waits for event k only).
    after_features: optional callable run once the (first-pass) feature extraction has finished and before the matching
    starts - the point from which the per-image worker streams are idle; a caller that streams the PREVIOUS result to
    the host starts that copy here, where it cannot sit in front of a worker stream's kernels in a shared hardware queue.
    features: the handle submit_features() returned for these very images (their extraction was started earlier, e.g. from
    the previous call's after_matching hook); after_matching: optional callable run when the (first-pass) match lists are
    complete - the point from which no int8 kernel runs any more in this call; after_ransac: the same after this rank's
    RANSAC batch (its three launches are latency-bound: beside a busy chip they take 6 instead of 2.3 ms).
    Returns (panorama of the component that holds the best-connected image, uint8 H x W x 3 CUDA tensor; info dict
    with info["panoramas"]: one entry per connected component of at least two images, in component order)."""
    from . import pipeline as pl
    from . import renderPanorama as rp

    ws, rank = world()
    dev = next(iter(local_images.values())).device if local_images else torch.device("cuda", torch.cuda.current_device())
    times = pl.StageTimes()

    # 0) the render will need every source image everywhere: start that all-gather now, collect it at step 6
    t0 = time.perf_counter()
    if _multi(ws) and image_events is not None:
        for ev in image_events.values():
            ev.synchronize()  # the early image all-gather reads every local image
        image_events = None
    img_gather = ImageGather(local_images, n, dev)
    times.add("exchange", t0)

    res = _match_pass(input, local_images, n, seed, times, dev, image_events, before_match=img_gather.finish,
                      after_features=after_features, features=features, after_matching=after_matching, after_ransac=after_ransac)

    # 5) host segment (redundant on every rank): components, second pass if asked for, cameras per component
    t0 = time.perf_counter()
    ncomp, labels = pl.connected_components(res["num_matches"])
    times.add("host_cameras", t0)
    if (int(input.get("resizeImage", 0)) == 1 and int(input.get("resizeImagePanoramaCluster", 0)) == 1 and ncomp > 1
            and local_originals is not None):
        # imageMatchingPanoramaConComps.m:48-91: images resized per component (the common size is that of the
        # component), features re-extracted, everything re-matched and re-verified; the component labels of the
        # FIRST pass are kept (the reference does not recompute them)
        t0 = time.perf_counter()
        szs = torch.zeros((n, 2), dtype=torch.int64, device=dev)
        for k, im_ in local_originals.items():
            szs[k, 0], szs[k, 1] = int(im_.shape[0]), int(im_.shape[1])
        if _multi(ws):
            _all_reduce(szs)
        first_hw = torch.zeros((n, 2), dtype=torch.int64, device=dev)
        for k, im_ in local_images.items():
            first_hw[k, 0], first_hw[k, 1] = int(im_.shape[0]), int(im_.shape[1])
        local_images = pl.resize_per_component(input, local_originals, labels, szs.cpu().tolist())  # resident: no host hop
        new_hw = torch.zeros((n, 2), dtype=torch.int64, device=dev)
        for k, im_ in local_images.items():
            new_hw[k, 0], new_hw[k, 1] = int(im_.shape[0]), int(im_.shape[1])
        if _multi(ws):
            both_hw = torch.cat([first_hw, new_hw], 1)
            _all_reduce(both_hw)
            first_hw, new_hw = both_hw[:, :2], both_hw[:, 2:]
        if Ks is not None:  # the intrinsics follow the per-component resize (pipeline.rescale_K)
            fh, nh = first_hw.cpu().tolist(), new_hw.cpu().tolist()
            Ks = [pl.rescale_K(K, fh[k], nh[k]) for k, K in enumerate(Ks)]
        img_gather.wait()
        img_gather = ImageGather(local_images, n, dev)
        times.add("exchange", t0)
        first_counts = res["counts"]
        res = _match_pass(input, local_images, n, seed, times, dev, before_match=img_gather.finish)
        res["second_pass"], res["counts_first_pass"] = True, first_counts
    t0 = time.perf_counter()
    comps = pl.recognize_panoramas(n, res["pairs"], res["models"], res["num_matches"], Ks, labels, cameras)
    times.add("host_cameras", t0)

    # 6) render every component
    t0 = time.perf_counter()
    images = img_gather.wait()
    if _multi(ws):
        torch.cuda.synchronize()  # the collective ran on RCCL's stream; the library reads the images on its own
    times.add("exchange", t0)
    t0 = time.perf_counter()
    opts = {"anglePower": 2, "blending": input["blending"], "pyrLevels": input["bands"], "pyrSigma": input["MBBsigma"],
            "canvasColor": input["canvasColor"], "tile": tile, "cropBorder": bool(input.get("cropBorder", True))}  # displayPanorama.m:101
    mode = input["panorama2DisplaynSave"]
    # geometry of every canvas (host, f64) decides the sharding: components to ranks when there are enough of them
    geos = []
    for c in comps:
        sizes = [(int(images[k].shape[0]), int(images[k].shape[1]), 3) for k in c["members"]]
        o_ = rp.default_opts(opts, c["cameras"], c["ref"])
        geos.append((sizes, rp.canvas_geometry(c["cameras"], sizes, mode, c["ref"], o_)))
    by_component = _multi(ws) and len(comps) >= ws
    comp_owner = partition_weighted([float(g["H"]) * float(g["W"]) for (_, g) in geos], ws) if by_component else None
    panos = []
    for ci, c in enumerate(comps):
        sizes, geo = geos[ci]
        members = c["members"]
        gains = None
        if input.get("gainCompensation"):
            # gainCompensationRKf between cameras and render (renderPanorama.m:303-330).  The device sums in an
            # unspecified order, so rank 0's gains are broadcast: every tile must be rendered with the same numbers.
            from . import gainCompensation as gc

            tg = time.perf_counter()
            gains = gc.gainCompensationRKf([images[k] for k in members], c["cameras"], mode, c["ref"], input, geo)
            if _multi(ws):
                gt_ = torch.from_numpy(np.ascontiguousarray(gains)).to(dev)
                _broadcast(gt_, 0)
                gains = gt_.cpu().numpy()
            times.add("gain_compensation", tg)
        if by_component:
            pano = None
            if int(comp_owner[ci]) == rank:
                pano, _ = rp.renderPanorama(input, [images[k] for k in members], sizes, c["cameras"], mode, c["ref"], opts,
                                            gains=gains, device_out=True, geo=geo)
                pl._sync()
            root = pano_root if pano_root is not None else None
            pano = _deliver_panorama(pano, int(comp_owner[ci]), root, dev)
        else:
            # tiles second: every rank a contiguous, area-balanced run of the tile list (APS_TILE_DEAL=mod: t % world, rounds 1-4)
            tranges = None
            if _multi(ws) and os.environ.get("APS_TILE_DEAL") != "mod":
                tranges = tile_ranges(int(geo["H"]), int(geo["W"]), rp.effective_tile(opts, geo), ws)
            subset = None if not _multi(ws) else (("range",) + tranges[rank]) if tranges is not None else (rank, ws)
            pano, _ = rp.renderPanorama(input, [images[k] for k in members], sizes, c["cameras"], mode, c["ref"], opts,
                                        gains=gains, device_out=True, tile_subset=subset, geo=geo,
                                        # (runs gathered to a root cover the canvas: nothing outside a rank's run is read)
                                        zero_rest=not (tranges is not None and pano_root is not None))
            pl._sync()
            if _multi(ws):
                torch.cuda.synchronize()
                if pano_root is None:
                    _all_reduce(pano, dist.ReduceOp.MAX)  # disjoint tiles, zero elsewhere
                else:
                    pano = gather_tiles_to_root(pano, rp.effective_tile(opts, geo), pano_root, ranges=tranges)
                torch.cuda.synchronize()
                if opts["cropBorder"] and (pano_root is None or rank == pano_root):
                    pano = rp.cropNonzeroBbox(pano, opts["canvasColor"])[0]  # the combined canvas (renderPanorama.m:430-432)
        panos.append(pano)
    times.add("render", t0)
    main = 0
    if comps:
        deg = (res["num_matches"] + res["num_matches"].T).sum(1)
        best = int(np.argmax(deg))
        main = next((ci for ci, c in enumerate(comps) if best in c["members"]), 0)
    pano = panos[main] if panos else torch.zeros((0, 0, 3), dtype=torch.uint8, device=dev)
    info = {"times": dict(times), "n_features": res["counts"], "n_pairs_verified": len(res["pairs"]),
            "n_components": int(ncomp), "panorama_shape": tuple(int(v) for v in pano.shape) if pano is not None else None,
            "members": comps[main]["members"] if comps else [], "cameras": _scatter_cameras(comps, n),
            "pairs": res["pairs"], "models": res["models"], "components": comps, "panoramas": panos, "labels": labels,
            "second_pass": bool(res.get("second_pass", False)), "n_features_first_pass": res.get("counts_first_pass")}
    return pano, info


def _scatter_cameras(comps, n):
    cams = [None] * n
    for c in comps:
        for k, cam in zip(c["members"], c["cameras"]):
            cams[k] = cam
    return cams


def _deliver_panorama(pano, owner, root, dev):
    """A panorama rendered whole by `owner`: to `root` only (point to point), or to every rank (broadcast) when root
    is None.  Ranks that neither rendered nor receive it return None."""
    ws, rank = world()
    shp = torch.zeros(3, dtype=torch.int64, device=dev)
    if rank == owner:
        shp = torch.tensor([int(v) for v in pano.shape], dtype=torch.int64, device=dev)
    _broadcast(shp, owner)
    shape = tuple(int(v) for v in shp.cpu().tolist())
    if root is None:
        if rank != owner:
            pano = torch.empty(shape, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        _broadcast(pano, owner)
        return pano
    if owner == root:
        return pano if rank == root else None
    if rank == owner:
        torch.cuda.synchronize()
        if _staged(pano):
            dist.send(pano.cpu(), dst=root)
        else:
            dist.send(pano, dst=root)
        return pano
    if rank == root:
        if dist.get_backend() == "gloo":
            h = torch.empty(shape, dtype=torch.uint8)
            dist.recv(h, src=owner)
            return h.to(dev)
        out = torch.empty(shape, dtype=torch.uint8, device=dev)
        dist.recv(out, src=owner)
        return out
    return None
