"""Host-side mirror of PP/imageMatching/ (imageMatching.m, estimateTransformationRANSAC.m) on top of the
batched device RANSAC of libaps_hip.so.

The reference draws its 4-point samples from MATLAB's unseeded global RNG inside parfor workers
(estimateTransformationRANSAC.m:96); here the draws are an explicit, seeded input so runs are
reproducible and the device result can be compared bit-exactly with the oracle on the same draws.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr


def draw_samples(counts, n_samples, seed=0, keys=None):
    """randperm(numPoints, 4) for every loop iteration of every pair (estimateTransformationRANSAC.m:96).

    counts: matches per pair.  Returns uint32 [n_pairs, n_samples, 4], 1-based, distinct within a draw
    (a pair with fewer than 4 matches gets that many distinct entries, then ones).  Counter-based hash stream keyed
    by (`seed`, keys[p] or p): pass the GLOBAL pair index as key and the draws do not depend on how the
    pairs are sharded over GPUs."""
    counts = np.asarray(counts, np.int64).reshape(-1)
    P = counts.size
    out = np.ones((P, n_samples, 4), np.uint32)
    if P == 0:
        return out
    # counter-based uniforms: u[p, it, k] = mix64(seed, key_p, 4*it + k) (splitmix64 finaliser), vectorised
    # over pairs; a draw depends only on (seed, global pair id, iteration), never on the sharding or batch
    kk = np.arange(P, dtype=np.uint64) if keys is None else np.asarray(keys, np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + kk[:, None, None] * np.uint64(0xD1B54A32D192ED03)
             + np.arange(n_samples * 4, dtype=np.uint64).reshape(1, n_samples, 4) * np.uint64(0x8CB92BA72F3D8DD7)
             + np.uint64(0x2545F4914F6CDD1D))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u = (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    n = counts[:, None]
    # partial Fisher-Yates without a table: the k-th draw picks among the n-k values not chosen yet, so it skips over
    # the earlier picks in ascending order.  A pair with fewer than 4 matches gets n distinct entries, then ones.
    srt = []  # the picks so far, ascending (element-wise)
    for k in range(4):
        left = np.maximum(n - k, 1)
        v = np.minimum((u[..., k] * left).astype(np.int64), left - 1)
        for s_j in srt:
            v = v + (v >= s_j)
        have = np.broadcast_to(k < n, v.shape)
        out[..., k] = np.where(have, v + 1, 1).astype(np.uint32)
        ins = v
        for j in range(len(srt)):  # insert into the ascending list
            lo, hi = np.minimum(srt[j], ins), np.maximum(srt[j], ins)
            srt[j], ins = lo, hi
        srt.append(ins)
    return out


def draw_samples_device(counts, n_samples, seed=0, keys=None):
    """The same draws produced on the device (aps_ransac_draw_samples); returns a torch CUDA int32 tensor
    [n_pairs, n_samples, 4] that can be handed to ransac_batch without a host round trip."""
    import torch

    counts = np.ascontiguousarray(counts, np.int64).reshape(-1)
    P = counts.size
    out = torch.empty((P, n_samples, 4), dtype=torch.int32, device="cuda")
    k = None if keys is None else np.ascontiguousarray(keys, np.uint64)
    check(lib.aps_ransac_draw_samples(ptr(counts), ptr(k), P, int(n_samples), int(seed), ptr(out)))
    return out


TFORM_TYPES = {"projective": _capi.APS_TFORM_PROJECTIVE, "affine": _capi.APS_TFORM_AFFINE,
               "similarity": _capi.APS_TFORM_SIMILARITY, "rigid": _capi.APS_TFORM_RIGID,
               "translation": _capi.APS_TFORM_TRANSLATION}
MIN_POINTS = {"projective": 4, "affine": 3, "similarity": 2, "rigid": 2, "translation": 1}  # getTransformParams :612-660


def _ransac_opts(input, method=None, transformType=None):
    """aps_ransac_opts from the reference's `input` struct.  method: 'ransac' | 'mlesac' (default:
    input.imageMatchingMethod, inputs.m:66); MLESAC's own defaults are maxDistance 2, 1000 trials (:740-765).
    transformType defaults to input.transformationType (inputs.m:74)."""
    method = str(method if method is not None else input.get("imageMatchingMethod", "ransac")).lower()
    if method not in ("ransac", "mlesac"):
        raise ValueError(f"unknown imageMatchingMethod '{method}'")
    o = _capi.aps_ransac_opts()
    o.max_distance = float(input.get("maxDistance", 2.0))
    o.confidence = float(input.get("inliersConfidence", 99.9))
    o.max_iter = int(input.get("maxIter", 1000 if method == "mlesac" else 500))
    tform = _check_type(transformType if transformType is not None else input.get("transformationType", "projective"), method)
    o.tform_type = TFORM_TYPES[tform]
    o.method = _capi.APS_ROBUST_MLESAC if method == "mlesac" else _capi.APS_ROBUST_RANSAC
    return o


def _check_type(transformType, method="ransac"):
    """The lower-cased transform type (getTransformParams :648-660 lower-cases too).  All five run on the device through
    both estimateTransformationRANSAC (:227-452) and estimateTransformationMLESAC (:345-510)."""
    tform = str(transformType).lower()
    if tform not in TFORM_TYPES:
        raise ValueError("Unknown transform type")  # estimateTransformationRANSAC.m:658-659
    return tform


def _estimate_robust(matchedPoints1, matchedPoints2, transformType, input, sample_idx, seed, method):
    tform = _check_type(transformType, method)
    input = {} if input is None else input
    p1 = np.asfortranarray(np.asarray(matchedPoints1, np.float64))
    p2 = np.asfortranarray(np.asarray(matchedPoints2, np.float64))
    if p1.ndim != 2 or p1.shape[1] != 2 or p2.ndim != 2 or p2.shape[1] != 2:
        raise ValueError("matchedPoints must be M-by-2")
    if p1.shape[0] != p2.shape[0]:
        raise ValueError("matchedPoints1 and matchedPoints2 must have the same number of rows.")
    m = p1.shape[0]
    if m < MIN_POINTS[tform]:  # :71-76
        return None, np.zeros(m, bool), False
    o = _ransac_opts(input, method, tform)
    if sample_idx is None:
        sample_idx = draw_samples([m], o.max_iter + 64, seed)[0]
    s = np.ascontiguousarray(sample_idx, np.uint32)
    model = np.zeros(9, np.float64)
    mask = np.zeros(m, np.uint8)
    found = C.c_int(0)
    trials = C.c_int(0)
    check(lib.aps_ransac_homography(ptr(p1), ptr(p2), m, m, ptr(s), s.shape[0], C.byref(o), ptr(model),
                                    ptr(mask), C.byref(found), C.byref(trials)))
    if not found.value:
        return None, mask.astype(bool), False
    return model.reshape(3, 3).T.copy(), mask.astype(bool), True


def estimateTransformationRANSAC(matchedPoints1, matchedPoints2, transformType, input=None,
                                 sample_idx=None, seed=0):
    """[model, inliers, isFound] = estimateTransformationRANSAC(matchedPoints1, matchedPoints2,
    transformType, input) (estimateTransformationRANSAC.m:1-183).

    model maps matchedPoints1 -> matchedPoints2.  Returns (3x3 float64 or None, bool[M], bool)."""
    if input is None:
        input = {"maxDistance": 2.0, "inliersConfidence": 99.9, "maxIter": 500}
    return _estimate_robust(matchedPoints1, matchedPoints2, transformType, input, sample_idx, seed, "ransac")


def estimateTransformationMLESAC(points1, points2, transformationType, input=None, sample_idx=None, seed=0):
    """[tform, inlierIdx, isFound] = estimateTransformationMLESAC(points1, points2, transformationType, input)
    (estimateTransformationMLESAC.m:1-254): truncated-loss consensus on the one-way reprojection distance,
    the refit on the best model's inliers is the answer.  All five transformationTypes; draws are explicit
    (sample_idx, uint32 n x 4, 1-based) or seeded.  Returns (3x3 float64 or None, bool[M], bool)."""
    return _estimate_robust(points1, points2, transformationType, input, sample_idx, seed, "mlesac")


def ransac_score(Hs, p1, p2, thr, transformType="projective"):
    """findInliers for T hypotheses (estimateTransformationRANSAC.m:444-516): (n_inl, mean_err, mask[T,M])."""
    Hs = np.asarray(Hs, np.float64)
    T = Hs.shape[0]
    Hc = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))
    a = np.asfortranarray(np.asarray(p1, np.float64))
    b = np.asfortranarray(np.asarray(p2, np.float64))
    m = a.shape[0]
    n = np.zeros(T, np.int32)
    e = np.zeros(T, np.float64)
    mask = np.zeros((T, max(m, 1)), np.uint8)
    check(lib.aps_ransac_score(ptr(Hc), T, ptr(a), ptr(b), m, m, float(thr), TFORM_TYPES[_check_type(transformType)],
                               ptr(n), ptr(e), ptr(mask)))
    return n, e, mask[:, :m]


def ransac_batch(pts_src, pts_dst, pair_ptr, samples, input):
    """Batched form used by imageMatching and the resident pipeline.
    pts_src/pts_dst: total x 2 float64 (numpy) or 2 x total float64 CUDA tensors; pair_ptr: int64[P+1];
    samples: uint32[P, S, 4] (numpy or CUDA tensor).
    Returns (models [P,3,3], mask uint8[total], found int32[P], n_inl int32[P])."""
    pair_ptr = np.ascontiguousarray(pair_ptr, np.int64)
    P = pair_ptr.size - 1
    total = int(pair_ptr[-1])
    if _capi.is_torch(pts_src):
        # resident form: 2 x total float64 CUDA tensors (row 0 = x, row 1 = y), i.e. column-major total x 2
        import torch
        a, b = pts_src.contiguous(), pts_dst.contiguous()
        assert a.dtype == torch.float64 and b.dtype == torch.float64 and a.shape == (2, total) and b.shape == (2, total)
        ld = max(total, 1)
    else:
        a = np.asfortranarray(np.asarray(pts_src, np.float64))
        b = np.asfortranarray(np.asarray(pts_dst, np.float64))
        ld = max(total, 1) if a.shape[0] == 0 else a.shape[0]
    if not _capi.is_torch(samples):
        samples = np.ascontiguousarray(samples, np.uint32)
    assert samples.shape[0] == P and samples.shape[2] == 4
    models = np.zeros((P, 9), np.float64)
    mask = np.zeros(max(total, 1), np.uint8)
    found = np.zeros(P, np.int32)
    ninl = np.zeros(P, np.int32)
    o = _ransac_opts(input)
    check(lib.aps_ransac_homography_batch(ptr(a), ptr(b), ld,
                                          ptr(pair_ptr), P, ptr(samples), samples.shape[1], C.byref(o),
                                          ptr(models), ptr(mask), ptr(found), ptr(ninl)))
    return models.reshape(P, 3, 3).transpose(0, 2, 1).copy(), mask[:total], found, ninl


def gather_match_points(keypoints, idx_a, idx_b, list_start, work_ptr, img_a, img_b):
    """The matched keypoints of a list of candidate pairs (imageMatching.m:121-135) gathered on the device: work pair q
    is images (img_a[q], img_b[q]) and owns entries list_start[q] .. + (work_ptr[q+1] - work_ptr[q]) of the resident
    1-based match lists idx_a / idx_b (torch int32, CUDA).  keypoints: per image a CUDA float64 tensor [n_i, 2] = [x y].
    Returns (pts_a, pts_b): CUDA float64 [2, total], row 0 = x, row 1 = y - the layout ransac_batch takes."""
    import torch

    n_img, n_work = len(keypoints), len(img_a)
    kps = [k.contiguous() if k.dtype == torch.float64 else k.to(torch.float64).contiguous() for k in keypoints]
    work_ptr = np.ascontiguousarray(work_ptr, np.int64)
    total = int(work_ptr[-1])
    dev = idx_a.device
    pa = torch.empty((2, total), dtype=torch.float64, device=dev)
    pb = torch.empty((2, total), dtype=torch.float64, device=dev)
    if total == 0:
        return pa, pb
    tab = (C.c_void_p * n_img)(*[k.data_ptr() if k.numel() else None for k in kps])
    cnt = np.ascontiguousarray([int(k.shape[0]) for k in kps], np.int64)
    ls = np.ascontiguousarray(list_start, np.int64)
    ia = np.ascontiguousarray(img_a, np.int32)
    ib = np.ascontiguousarray(img_b, np.int32)
    la, lb = idx_a.contiguous(), idx_b.contiguous()  # (named: the copies must outlive the call)
    torch.cuda.current_stream().synchronize()  # the tables / lists / fresh blocks may still be in use on torch's stream
    check(lib.aps_gather_match_points(tab, ptr(cnt), n_img, ptr(la), ptr(lb), ptr(ls), ptr(work_ptr),
                                      ptr(ia), ptr(ib), n_work, ptr(pa), ptr(pb), total))
    return pa, pb


def ransac_batch_drawn(pts_src, pts_dst, pair_ptr, counts, input, seed=0, keys=None):
    """ransac_batch with the draws made here: maxIter + 64 subsets per pair from the counter-based stream, and - when a
    pair burns through them on invalid / degenerate subsets before the loop's own stopping rule (the reference keeps
    drawing up to 10*maxIter skipped trials, estimateTransformationRANSAC.m:94) - again with four times as many.  The
    stream only gets longer, so pairs that did not run out reproduce their result."""
    max_iter = int(input.get("maxIter", 500))
    n_samples, limit = max_iter + 64, 11 * max_iter + 64
    while True:
        samples = draw_samples_device(counts, n_samples, seed, keys=keys)
        out = ransac_batch(pts_src, pts_dst, pair_ptr, samples, input)
        if lib.aps_ransac_draws_exhausted() == 0 or n_samples >= limit:
            return out
        n_samples = min(4 * n_samples, limit)


def candidate_pairs(matchesAll, n, m):
    """Top-m candidate selection of imageMatching.m:76-100 (Brown-Lowe m = 6): union over images of the m
    partners with the most putative matches (stable descending sort), upper triangle, column-major order."""
    put = np.zeros((n, n), np.int64)
    for i in range(n):
        for j in range(n):
            c = matchesAll[i][j]
            put[i, j] = 0 if c is None else len(c)
    sym = put + put.T
    np.fill_diagonal(sym, 0)
    order = np.argsort(-sym, axis=1, kind="stable")  # MATLAB sort(...,'descend') is stable
    top = order[:, : min(m, n - 1)]
    cand = np.zeros((n, n), bool)
    cand[np.repeat(np.arange(n), top.shape[1]), top.reshape(-1)] = True
    cand = np.triu(cand | cand.T, 1)
    jj, ii = np.nonzero(cand.T)  # find() walks column-major
    return list(zip(ii.tolist(), jj.tolist()))


def imageMatching(input, n, keypoints, matchesAll, imagesProcessed=None, seed=0):
    """[allMatches, numMatches, tforms] = imageMatching(input, n, keypoints, matchesAll, imagesProcessed)
    (imageMatching.m:1-167).  tforms[i][j] maps image-j points to image-i points (:242), tforms[j][i] is
    its inverse (:154); a pair is accepted iff ni > 8 + 0.3*nf (:150)."""
    n = int(n)
    if len(matchesAll) != n or any(len(r) != n for r in matchesAll):
        raise ValueError("matchesAll must be an n-by-n cell array.")
    if len(keypoints) != n:
        raise ValueError("keypoints must contain n elements (one per image).")
    if imagesProcessed is not None and len(imagesProcessed) != n:
        raise ValueError("images must contain n elements (one per image).")
    _check_type(input.get("transformationType", "projective"), str(input.get("imageMatchingMethod", "ransac")).lower())
    allMatches = [[None] * n for _ in range(n)]
    numMatches = np.zeros((n, n))
    tforms = [[None] * n for _ in range(n)]
    pairs = candidate_pairs(matchesAll, n, int(input.get("mBrownLowe", 6)))
    work = []
    for (i, j) in pairs:
        mt = matchesAll[i][j]
        if mt is None or len(mt) < 4:  # :133
            continue
        mt = np.asarray(mt)
        k1 = np.asarray(keypoints[i], np.float64)
        k2 = np.asarray(keypoints[j], np.float64)
        a = mt[:, 0].astype(np.int64)
        b = mt[:, 1].astype(np.int64)
        if a.max() > len(k1) or b.max() > len(k2) or a.min() < 1 or b.min() < 1:
            raise ValueError("Match indices exceed keypoint array sizes.")  # refineMatch :222-226
        work.append((i, j, mt, k2[b - 1], k1[a - 1]))  # (pts_j, pts_i): model maps j -> i (:242)
    if not work:
        return allMatches, numMatches, tforms
    counts = [len(w[2]) for w in work]
    pair_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    src = np.concatenate([w[3] for w in work])
    dst = np.concatenate([w[4] for w in work])
    n_samples = int(input.get("maxIter", 500)) + 64
    samples = draw_samples(counts, n_samples, seed)
    models, mask, found, ninl = ransac_batch(src, dst, pair_ptr, samples, input)
    for p, (i, j, mt, _, _) in enumerate(work):
        nf = len(mt)
        inl = np.nonzero(mask[pair_ptr[p]:pair_ptr[p + 1]])[0]
        ni = len(inl) if found[p] else 0
        if ni > 8 + 0.3 * nf:  # :150
            allMatches[i][j] = np.asarray(mt, np.float64)[inl]
            numMatches[i, j] = ni
            tforms[i][j] = models[p]
            tforms[j][i] = np.linalg.inv(models[p])
    return allMatches, numMatches, tforms
