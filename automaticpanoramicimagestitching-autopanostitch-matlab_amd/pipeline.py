"""The reference driver's stage order (PP/main.m:83-138) over the device hot path.

    loadImages/getFeaturePoints -> featureMatchingPairwise -> imageMatching (+ connected components)
    -> [host: camera initialisation; bundle adjustment / straightening / gain compensation are the
       reference's host code and out of scope] -> renderPanorama (warp + multiband blend)

Everything heavy stays resident in HBM between stages: descriptors never leave the device, images are
uploaded once, only keypoint coordinates, match index lists and 3x3 models visit the host (they feed the
host-side graph logic exactly as in the reference).
"""
from __future__ import annotations

import threading
import time

import numpy as np

from . import _capi
from . import featureMatching as fm
from . import imageMatching as im
from . import renderPanorama as rp

DEFAULT_INPUT = {
    # inputs.m:31-40
    "detector": "SIFT", "Sigma": 1.6, "NumLayersInOctave": 4, "ContrastThreshold": 0.00133, "EdgeThreshold": 6,
    # inputs.m:44-59 (the north-star configuration: pairwise, exhaustive, scratch matcher)
    "matchFeaturesPairwise": 1, "useMATLABFeatureMatch": 0, "Matchingmethod": "Exhaustive",
    "Matchingthreshold": 1.5, "Ratiothreshold": 0.6,
    # inputs.m:62-74
    "useMATLABImageMatching": 0, "imageMatchingMethod": "ransac", "mBrownLowe": 6, "maxIter": 500,
    "maxDistance": 5.5, "inliersConfidence": 99.9, "transformationType": "projective",
    # inputs.m:94-113
    "gainCompensation": 0, "blending": "multiband", "bands": 3, "MBBsigma": 1, "resizeImage": 0,
    "resizeImagePanoramaCluster": 0, "heightLimit": 800, "widthLimit": 800,
    "panorama2DisplaynSave": "spherical", "canvasColor": "black", "forcePlanarScan": False,
}


def default_input(**overrides):
    d = dict(DEFAULT_INPUT)
    d.update(overrides)
    return d


class StageTimes(dict):
    def add(self, name, t0):
        self[name] = self.get(name, 0.0) + (time.perf_counter() - t0)


def _sync():
    _capi.check(_capi.lib.aps_synchronize())


_SIFT_POOL = None


# Host threads / HIP streams of the per-image feature extraction.  Measured on the 64 x 4K scene on several boxes:
# 4, 5, 10, 11, 12, 14 threads give 88-93 ms, 10 the best or tied everywhere; 7-9 threads read 100 ms on some boxes and
# 92 on others (reproducibly per box; the HIP runtime maps streams onto 8 hardware queues), 12 costs the end-to-end
# step's uploads 14 ms.
SIFT_WORKERS_DEFAULT = 10


def sift_many(input, images, workers=SIFT_WORKERS_DEFAULT, ready=None):
    """getFeaturePoints for many images — the reference runs this loop as a parfor (loadImages.m:82-99).
    Here a few host threads each drive their own HIP stream (the C ABI is thread-safe with per-thread streams
    and workspaces), so the small-octave launches and the count read-backs of one image overlap with the
    large-octave kernels of another.  Results are returned in input order and are independent of the
    interleaving (every kernel is deterministic).
    ready: optional list of torch CUDA events, one per image: the worker of image k waits for ready[k] (e.g. the end
    of that image's host-to-device copy on a side stream) instead of a device-wide synchronisation, so the uploads of
    later images overlap with the SIFT kernels of earlier ones."""
    global _SIFT_POOL
    import os
    import torch
    from concurrent.futures import ThreadPoolExecutor

    workers = int(os.environ.get("APS_SIFT_WORKERS", workers))

    dev = _capi.is_torch(images[0]) and images[0].is_cuda
    if len(images) <= 1 or workers <= 1:
        if ready is not None:
            for ev in ready:
                ev.synchronize()
        elif dev:
            torch.cuda.synchronize()
        out = [fm.sift_extract(input, img, device_out=dev) for img in images]
        _sync()
        return out
    return [f.result() for f in sift_submit(input, images, workers, ready)]


_INIT_LOCK = threading.Lock()


def _init_worker(device):
    """Pool initializer: each worker creates its library context (device binding, stream, events) on its own, one
    worker after the other - ten threads racing through their first HIP calls at once is what rocprofv3 was seen to
    crash under."""
    import os
    with _INIT_LOCK:
        _capi.check(_capi.lib.aps_set_thread_device(int(device)))
        if os.environ.get("APS_SIFT_STREAM_PRIORITY"):  # (-1: the workers' streams on the device's lowest priority level; see bench.py)
            _capi.check(_capi.lib.aps_set_thread_stream_priority(int(os.environ["APS_SIFT_STREAM_PRIORITY"])))
        _capi.check(_capi.lib.aps_synchronize())


def release_device_memory():
    """Hands cached device memory back to the driver: the library's per-thread workspace pools (aps_release_workspace on the
    calling thread and on every thread of the extraction pool - a thread only frees its own blocks) and torch's caching
    allocator.  For a host that is about to start another process on the same GPU, or that has finished a large job: after
    a 256 x 4K stitch the pools hold tens of gigabytes that nothing else can use.  Blocks in use are kept."""
    import threading
    import torch

    _capi.check(_capi.lib.aps_release_workspace())
    from . import renderPanorama as _rp

    for pool in (_SIFT_POOL, getattr(_rp, "_RENDER_POOL", None)):
        n = len(getattr(pool, "_threads", ())) if pool is not None else 0
        if not n:
            continue
        gate = threading.Barrier(n)

        def free(_, gate=gate):
            gate.wait(timeout=30)  # one task per pool thread: nobody takes two
            return _capi.lib.aps_release_workspace()

        try:
            for rc in pool.map(free, range(n)):
                _capi.check(rc)
        except threading.BrokenBarrierError:
            pass  # a pool thread was busy: its blocks stay (they are in use or will be reused)
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


def sift_submit(input, images, workers=SIFT_WORKERS_DEFAULT, ready=None, points_device=False):
    """The asynchronous form of sift_many: one future per image, submitted in input order to the worker pool, so that
    a caller can start matching the first images while the later ones are still being extracted (parallel._match_pass).
    Each future resolves to (descriptors, keypoints) once that worker's stream has finished the image.
    points_device: resident images only - the keypoints stay on the device (fm.sift_extract)."""
    global _SIFT_POOL
    import os
    import torch
    from concurrent.futures import ThreadPoolExecutor

    workers = max(1, int(os.environ.get("APS_SIFT_WORKERS", workers)))
    dev = _capi.is_torch(images[0]) and images[0].is_cuda
    if dev and ready is None:
        torch.cuda.synchronize()  # the images were produced on torch's stream; worker streams must see them
    if _SIFT_POOL is None:
        _SIFT_POOL = ThreadPoolExecutor(max_workers=workers, initializer=_init_worker, initargs=(_capi.lib.aps_get_device(),))
    pool = _SIFT_POOL

    # every worker binds the device of the image it is handed (or the submitting thread's device for host images)
    # instead of inheriting the process-wide default, which is whatever device ANY thread selected last
    here = _capi.lib.aps_get_device()

    def work(k):
        _capi.check(_capi.lib.aps_set_thread_device(images[k].device.index if dev else here))
        if ready is not None:
            ready[k].synchronize()
        r = fm.sift_extract(input, images[k], device_out=dev, points_device=bool(dev and points_device), compact=len(images) > 96)
        _sync()  # this thread's stream
        return r

    return [pool.submit(work, k) for k in range(len(images))]


def extract_features(input, images, times=None):
    """loadImages' parfor body (loadImages.m:82-99): one getFeaturePoints per image, descriptors resident."""
    t0 = time.perf_counter()
    descs, kps = [], []
    for d, p in sift_many(input, images):
        descs.append(d)
        kps.append(p)
    if times is not None:
        times.add("features", t0)
    return descs, kps


def match_and_verify(input, descs, kps, seed=0, times=None, pair_subset=None):
    """featureMatchingPairwise + imageMatching (main.m:95-107) in CSR form.

    Returns dict(pairs=[(i,j)], models=[3x3 j->i], inliers=[K x 2 index arrays], numMatches n x n)."""
    n = len(descs)
    t0 = time.perf_counter()
    pair_ptr, ii, jj, _ = fm.match_pairwise_csr(descs, input["Ratiothreshold"], input["Matchingthreshold"], True)
    order = fm.pair_order(n)
    if times is not None:
        times.add("matching", t0)
    t0 = time.perf_counter()
    # top-m candidate selection (imageMatching.m:76-100) straight from the CSR counts
    put = np.zeros((n, n), np.int64)
    for p, (i, j) in enumerate(order):
        put[i, j] = pair_ptr[p + 1] - pair_ptr[p]
    sym = put + put.T
    srt = np.argsort(-sym, axis=1, kind="stable")[:, : min(int(input["mBrownLowe"]), n - 1)]
    cand = np.zeros((n, n), bool)
    cand[np.repeat(np.arange(n), srt.shape[1]), srt.reshape(-1)] = True
    cand = np.triu(cand | cand.T, 1)
    pidx = {ij: p for p, ij in enumerate(order)}
    work = []
    cj, ci = np.nonzero(cand.T)
    for (i, j) in zip(ci.tolist(), cj.tolist()):
        p = pidx[(i, j)]
        s, e = int(pair_ptr[p]), int(pair_ptr[p + 1])
        if e - s < 4:
            continue
        work.append((i, j, s, e))
    out = {"pairs": [], "models": [], "inliers": [], "numMatches": np.zeros((n, n)), "putative": put}
    if work:
        counts = [e - s for (_, _, s, e) in work]
        wptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        src = np.concatenate([kps[j][jj[s:e].astype(np.int64) - 1] for (_, j, s, e) in work])
        dst = np.concatenate([kps[i][ii[s:e].astype(np.int64) - 1] for (i, _, s, e) in work])
        models, mask, found, ninl = im.ransac_batch_drawn(src, dst, wptr, counts, input, seed)
        for w, (i, j, s, e) in enumerate(work):
            nf = e - s
            ni = int(ninl[w]) if found[w] else 0
            if ni > 8 + 0.3 * nf:  # imageMatching.m:150
                sel = np.nonzero(mask[wptr[w]:wptr[w + 1]])[0]
                out["pairs"].append((i, j))
                out["models"].append(models[w])
                out["inliers"].append(np.stack([ii[s:e][sel], jj[s:e][sel]], axis=1).astype(np.int64))
                out["numMatches"][i, j] = ni
    if times is not None:
        times.add("image_matching", t0)
    return out


def connected_components(numMatches):
    """graph(numMatches,'upper') + conncomp (imageMatchingPanoramaConComps.m:43-45)."""
    # union-find over the (few hundred) edges; components are numbered by their smallest member, ascending - what
    # scipy.sparse.csgraph.connected_components returns (it was 0.8 ms of argument checking per call for 64 nodes)
    a = np.asarray(numMatches) > 0
    n = a.shape[0]
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    ii, jj = np.nonzero(a)
    for i, j in zip(ii.tolist(), jj.tolist()):
        ri, rj = find(i), find(j)
        if ri != rj:
            parent[max(ri, rj)] = min(ri, rj)
    labels = np.empty(n, np.int32)
    seen = {}
    for v in range(n):
        r = find(v)
        if r not in seen:
            seen[r] = len(seen)
        labels[v] = seen[r]
    return len(seen), labels


def cameras_from_models(n, pairs, models, num_matches, Ks):
    """Minimal stand-in for the reference's HOST camera initialisation (initializeCameraMatrices.m:332-455:
    maximum spanning tree over the match graph + rotation propagation).  Bundle adjustment, straightening
    and gain compensation are the reference's own host code and are not rebuilt here.
    models[p] maps image-j pixels to image-i pixels for pairs[p] = (i, j): H = K_i R_i R_j' K_j^-1.
    Returns (cameras list or None for unreachable images, seed index)."""
    adj = {k: [] for k in range(n)}
    for p, (i, j) in enumerate(pairs):
        adj[i].append((num_matches[i, j], j, p, False))
        adj[j].append((num_matches[i, j], i, p, True))
    deg = [sum(w for w, *_ in adj[k]) for k in range(n)]
    seed = int(np.argmax(deg))
    import heapq

    # the tree first (it depends on the weights only), then every edge's relative rotation in ONE batched svd: edge by
    # edge the 63 small inv / svd calls of a 64-view set were most of this step's 1.7 ms
    reached = {seed}
    edges = []  # (parent, child, pair, child-is-the-pair's-first-image) in visiting order
    heap = [(-w, seed, nb, p, inv) for (w, nb, p, inv) in adj[seed]]
    heapq.heapify(heap)
    while heap:
        _, a, b, p, inv = heapq.heappop(heap)
        if b in reached:
            continue
        reached.add(b)
        edges.append((a, b, p, inv))
        for (w, nb, pp, inv2) in adj[b]:
            if nb not in reached:
                heapq.heappush(heap, (-w, b, nb, pp, inv2))
    R = {seed: np.eye(3)}
    if edges:
        Kinv = {}
        Ms = []
        for (_, _, p, _) in edges:
            i, j = pairs[p]
            if i not in Kinv:
                Kinv[i] = np.linalg.inv(Ks[i])
            Ms.append(Kinv[i] @ models[p] @ Ks[j])  # ~ R_i R_j'
        U, _, Vt = np.linalg.svd(np.stack(Ms))
        Rall = U @ Vt  # (stacked products and determinants: the per-matrix routines, called once)
        Rall = np.where((np.linalg.det(Rall) < 0)[:, None, None], -Rall, Rall)
        for e, (a, b, p, inv) in enumerate(edges):
            Rij = Rall[e]
            # a == i, b == j: R_j = Rij' R_i ; a == j, b == i: R_i = Rij R_j
            R[b] = Rij.T @ R[a] if not inv else Rij @ R[a]
    cams = [({"K": Ks[k], "R": R[k], "f": float(Ks[k][0, 0]), "noRotation": 0} if k in R else None) for k in range(n)]
    return cams, seed


def straightening(cameras, up_angle_t=(60, 60, 105), theta_t=90):
    """Host step, PP/straightening/straightening.m:74-176: global rotation S that makes the cameras' X axes
    horizontal (Brown-Lowe), R <- R*S, with the reference's skip rules.  3x3 algebra per component."""
    cams = [c for c in cameras if c is not None]
    if len(cams) < 2:
        return cameras
    X = np.stack([c["R"][0, :] for c in cams], axis=1)
    _, _, Vt = np.linalg.svd(X @ X.T)
    up = Vt[-1]
    avgY = np.mean(np.stack([c["R"][1, :] for c in cams], axis=1), axis=1)
    avgY = avgY / np.linalg.norm(avgY)
    if up @ avgY < 0:
        up = -up
    Zsum = np.sum(np.stack([c["R"][2, :] for c in cams], axis=1), axis=1)
    xhat = np.cross(up, Zsum)
    if np.linalg.norm(xhat) < np.finfo(float).eps:
        e1 = np.array([1.0, 0, 0]) if abs(up[0]) <= 0.99 else np.array([0, 0, 1.0])
        xhat = np.cross(up, e1)
    if np.linalg.norm(xhat) < np.finfo(float).eps:
        return cameras
    xhat = xhat / np.linalg.norm(xhat)
    zhat = np.cross(xhat, up)
    if np.linalg.norm(zhat) < np.finfo(float).eps:
        return cameras
    zhat = zhat / np.linalg.norm(zhat)
    S = np.stack([xhat, up, zhat], axis=1)
    theta = np.degrees(np.arccos(np.clip((np.trace(S) - 1) / 2, -1, 1)))
    up_angle = np.degrees(np.arccos(np.clip(abs(up[1]), -1, 1)))
    if up_angle_t[0] < up_angle < up_angle_t[2]:
        return cameras
    if up_angle > up_angle_t[1] and theta > theta_t:
        return cameras
    out = []
    for c in cameras:
        out.append(None if c is None else dict(c, R=c["R"] @ S))
    return out


def recognize_panoramas(n, pairs, models, num_matches, Ks, labels, cameras=None):
    """recognizePanoramas.m:70-113 + straightening per panorama (main.m:110-118) over the match graph: every connected
    component with at least two images becomes one panorama with its own cameras and its own reference image.
    The reference runs bundleAdjustmentRKf per component (host code, out of scope); here the cameras are the
    caller's (`cameras`, e.g. from that bundle adjustment) or the host stand-in cameras_from_models, run per component
    from that component's best-connected image.  Components come in conncomp order (by lowest image index).
    Returns a list of dict(members=[global image indices], ref=index INTO members, cameras=[one per member])."""
    labels = np.asarray(labels)
    deg = (np.asarray(num_matches) + np.asarray(num_matches).T).sum(1)
    comps = []
    for c in range(int(labels.max()) + 1 if labels.size else 0):
        members = [int(k) for k in np.nonzero(labels == c)[0]]
        if len(members) < 2:
            continue  # 'Skipping bundle adjustment as only one image found' (recognizePanoramas.m:86-89)
        if cameras is not None:
            cams = [cameras[k] for k in members]
            ref = int(np.argmax(deg[members]))
        else:
            if Ks is None:
                raise ValueError("either cameras or the intrinsics Ks must be given (focal estimation/BA are host code out of scope)")
            inside = set(members)
            sel = [p for p, (i, j) in enumerate(pairs) if i in inside and j in inside]
            cams_all, seed = cameras_from_models(n, [pairs[p] for p in sel], [models[p] for p in sel], num_matches, Ks)
            cams = straightening([cams_all[k] for k in members])
            ref = members.index(seed)
        keep = [q for q, cam in enumerate(cams) if cam is not None]
        if len(keep) < 2:
            continue
        if len(keep) != len(members):
            ref = keep.index(ref) if ref in keep else 0
            members = [members[q] for q in keep]
            cams = [cams[q] for q in keep]
        comps.append({"members": members, "ref": ref, "cameras": cams})
    return comps


def fit_size(h, w, heightLimit, widthLimit):
    """Stage-1 size of resizeImagesToLimits 'fit' (resizeImagesToLimits.m:57-61): isotropic shrink into the box,
    imresize's scalar form gives ceil(s * size); images inside the box keep their size."""
    import math

    sc = min(heightLimit / h, widthLimit / w)
    if not np.isfinite(sc) or sc <= 0:
        sc = 1.0
    return (int(math.ceil(h * sc)), int(math.ceil(w * sc)), sc) if sc < 1 else (int(h), int(w), 1.0)


def resize_per_component(input, local_originals, labels, all_sizes):
    """imageMatchingPanoramaConComps.m:48-56: imagesProcessed(idxs) = resizeImagesToLimits(imagesOriginal(idxs), ...,
    'fit') for every connected component - the same rule as loadImages.m:66-68, applied per panorama: shrink into the
    box, then bring the component's images to ITS common largest size (resizeImagesToLimits.m:49-106).
    local_originals: dict image index -> uint8 image (numpy or CUDA tensor) held by this process; all_sizes: (h, w)
    of EVERY original (the common size of a component depends on members other processes hold).
    Returns dict image index -> resized image of the same kind."""
    from . import imageProcessing as ip

    HL, WL = float(input["heightLimit"]), float(input["widthLimit"])
    labels = np.asarray(labels)
    out = {}
    for c in range(int(labels.max()) + 1):
        idxs = [int(k) for k in np.nonzero(labels == c)[0]]
        orig = [tuple(int(v) for v in all_sizes[k][:2]) for k in idxs]
        too_large = any(h > HL or w > WL for (h, w) in orig)
        if not too_large and len(set(orig)) == 1:
            for k in idxs:
                if k in local_originals:
                    out[k] = local_originals[k]  # all the same size, nothing to do (:49-55)
            continue
        st1 = [fit_size(h, w, HL, WL) for (h, w) in orig]
        common = None
        if len({(a, b) for (a, b, _) in st1}) > 1:
            common = (max(a for (a, _, _) in st1), max(b for (_, b, _) in st1))
        for k, (h1, w1, sc) in zip(idxs, st1):
            if k not in local_originals:
                continue
            a = local_originals[k]  # numpy stays numpy, a CUDA tensor stays resident (aps_imresize_u8 takes either)
            if sc < 1:
                a = ip.imresize(a, sc, "bicubic")
            if common is not None:  # every image of the set, also those already at that size (resizeImagesToLimits.m:100-104)
                a = ip.imresize(a, common, "bicubic")
            out[k] = a
    return out


def rescale_K(K, old_hw, new_hw):
    """Intrinsics of an image after imresize from old_hw to new_hw = (rows, cols): pixel x scales with the column count,
    y with the row count (imresize may be anisotropic when a set is brought to a common size).  The reference has no
    such step - its cameras come out of the bundle adjustment on the resized images; this serves the host stand-in,
    which takes the intrinsics as an input."""
    sy, sx = new_hw[0] / old_hw[0], new_hw[1] / old_hw[1]
    if sx == 1.0 and sy == 1.0:
        return np.asarray(K, np.float64)
    return np.diag([sx, sy, 1.0]) @ np.asarray(K, np.float64)


def imageMatchingPanoramaConComps(input, images_original, images_processed, descs, kps, seed=0, times=None):
    """[allMatchesRefined, numMatches, initialTforms, imagesProcessed, ..., concomps] =
    imageMatchingPanoramaConComps(...) (imageMatchingPanoramaConComps.m:39-91) in CSR form: first-pass matching and
    verification, connected components, and - with input.resizeImage and input.resizeImagePanoramaCluster set and
    more than one component - the second pass on images resized per component.
    Returns (result dict of match_and_verify, images_processed, descs, kps, ncomp, labels, second_pass: bool)."""
    res = match_and_verify(input, descs, kps, seed, times)
    ncomp, labels = connected_components(res["numMatches"])
    second = int(input.get("resizeImage", 0)) == 1 and int(input.get("resizeImagePanoramaCluster", 0)) == 1 and ncomp > 1
    if second:
        sizes = [(int(im_.shape[0]), int(im_.shape[1])) for im_ in images_original]
        resized = resize_per_component(input, dict(enumerate(images_original)), labels, sizes)
        images_processed = [resized[k] for k in range(len(images_original))]
        descs, kps = extract_features(input, images_processed, times)
        res = match_and_verify(input, descs, kps, seed, times)
    return res, images_processed, descs, kps, ncomp, labels, second


def stitch(input, images, Ks=None, cameras=None, tile=(2048, 2048), seed=0, device_out=True, profile=False,
           images_original=None):
    """main.m for one dataset.  images: list of uint8 H x W x 3 (torch CUDA tensors stay resident).
    cameras: optional externally supplied cameras (e.g. from the reference's own bundle adjustment);
    otherwise they are initialised on the host from the verified homographies and the intrinsics Ks.
    Every connected component of at least two images is rendered (displayPanorama.m:88-116).
    Returns (panoramas [one per component, in component order], info dict with per-stage wall times in seconds)."""
    times = StageTimes()
    n = len(images)
    descs, kps = extract_features(input, images, times)
    first_hw = [(int(im_.shape[0]), int(im_.shape[1])) for im_ in images]
    first_counts = [len(k) for k in kps]
    res, images, descs, kps, ncomp, labels, second = imageMatchingPanoramaConComps(
        input, images if images_original is None else images_original, images, descs, kps, seed, times)
    sizes = [(int(im_.shape[0]), int(im_.shape[1]), 3) for im_ in images]
    if second and Ks is not None:  # the intrinsics follow the per-component resize
        Ks = [rescale_K(K, first_hw[k], sizes[k][:2]) for k, K in enumerate(Ks)]
    t0 = time.perf_counter()
    comps = recognize_panoramas(n, res["pairs"], res["models"], res["numMatches"], Ks, labels, cameras)
    times.add("host_cameras", t0)
    panos = []
    t0 = time.perf_counter()
    opts = {"anglePower": 2, "blending": input["blending"], "pyrLevels": input["bands"], "pyrSigma": input["MBBsigma"],
            "canvasColor": input["canvasColor"], "tile": tile, "cropBorder": bool(input.get("cropBorder", True))}  # displayPanorama.m:101
    for c in comps:
        members = c["members"]
        pano, _ = rp.renderPanorama(input, [images[k] for k in members], [sizes[k] for k in members],
                                    c["cameras"], input["panorama2DisplaynSave"], c["ref"], opts,
                                    device_out=device_out)
        panos.append(pano)
    _sync()
    times.add("render", t0)
    cams = [None] * n
    for c in comps:
        for k, cam in zip(c["members"], c["cameras"]):
            cams[k] = cam
    info = {"times": dict(times), "n_features": [len(k) for k in kps], "n_pairs_verified": len(res["pairs"]),
            "n_components": int(ncomp), "putative": res["putative"], "result": res, "cameras": cams,
            "components": comps, "labels": labels, "second_pass": bool(second), "n_features_first_pass": first_counts,
            "images_processed": images}
    return panos, info
