"""Host-side mirror of PP/featureMatching/ — same operator names, argument meaning and outputs as the
reference's MATLAB functions, with the arithmetic done by libaps_hip.so on the MI355X.

MATLAB cell arrays become Python lists (n x n cells -> list of lists), structs become dicts, and
index outputs stay 1-based ``float64`` exactly like ``double(matches)`` in
featureMatchingPairwise.m:120.  Inputs may be numpy arrays (host; staged through HBM by the library)
or torch CUDA tensors (device resident; no copies).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import check, lib, ptr

DIM = 128


def _as_desc(x):
    """Row-major float32 N x 128 view of a descriptor matrix (numpy or torch); returns (obj, n, ld, layout)."""
    if _capi.is_torch(x):
        import torch

        if x.dtype != torch.float32:
            x = x.float()
        if x.dim() != 2:
            raise ValueError("descriptors must be 2-D")
        if x.stride(1) == 1 and (x.shape[0] <= 1 or x.stride(0) >= x.shape[1]):
            return x, x.shape[0], max(x.stride(0), x.shape[1]), _capi.APS_ROWMAJOR
        if x.stride(0) == 1:  # column-major (MATLAB) storage
            return x, x.shape[0], max(x.stride(1), x.shape[0]), _capi.APS_COLMAJOR
        x = x.contiguous()
        return x, x.shape[0], x.shape[1], _capi.APS_ROWMAJOR
    x = np.asarray(x)
    if x.dtype != np.float32:
        x = x.astype(np.float32)  # normalizeInputs casts non-float input to single (matchFeaturesScratch.m:277-278)
    if x.ndim != 2:
        raise ValueError("descriptors must be 2-D")
    if x.flags.f_contiguous and not x.flags.c_contiguous:
        return x, x.shape[0], x.shape[0], _capi.APS_COLMAJOR
    x = np.ascontiguousarray(x)
    return x, x.shape[0], x.shape[1], _capi.APS_ROWMAJOR


def _opts(MaxRatio, MatchThreshold, Unique, normalize=2):
    o = _capi.aps_match_opts()
    o.max_ratio = float(MaxRatio)
    o.match_threshold = float(MatchThreshold)
    o.unique = 1 if Unique else 0
    o.normalize = int(normalize)
    return o


def nearest2SSDExhaustive(A, B):
    """[idx1, idx2, d1, d2] = nearest2SSDExhaustive(A, B) (matchFeaturesScratch.m:322-366)."""
    A, n1, lda, la = _as_desc(A)
    B, n2, ldb, lb = _as_desc(B)
    if la != lb:
        raise ValueError("A and B must share a storage order")
    if n1 and A.shape[1] != DIM or n2 and B.shape[1] != DIM:
        raise ValueError("Descriptor dimensions must match for non-binary.")
    idx2 = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    check(lib.aps_match_2nn_ssd(ptr(A), n1, lda, ptr(B), n2, ldb, DIM, la, ptr(idx2), ptr(d1), ptr(d2)))
    return np.arange(1, n1 + 1, dtype=np.float64), idx2, d1, d2


def _normalize_like_reference(A, B):
    """matchFeaturesScratch.m:105-110: both sets are L2-row-normalised (x ./ (||x|| + eps('single')), :232-233)
    iff either has an entry above 2 in magnitude."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    if (A.size and np.abs(A).max() > 2) or (B.size and np.abs(B).max() > 2):
        eps = np.float32(np.finfo(np.float32).eps)
        A = A / (np.sqrt((A * A).sum(1, dtype=np.float32, keepdims=True)) + eps)
        B = B / (np.sqrt((B * B).sum(1, dtype=np.float32, keepdims=True)) + eps)
    return np.ascontiguousarray(A, np.float32), np.ascontiguousarray(B, np.float32)


def _pad_dim(X):
    """The device 2-NN kernel is built for 128-D rows; shorter rows are zero-padded (distances unchanged)."""
    X = np.ascontiguousarray(X, np.float32)
    if X.shape[1] == DIM:
        return X
    if X.shape[1] > DIM:
        raise ValueError(f"descriptor length {X.shape[1]} > {DIM}")
    out = np.zeros((X.shape[0], DIM), np.float32)
    out[:, : X.shape[1]] = X
    return out


def nearest2KDTree(A, B, bucketSize=40):
    """[idx1, idx2, d1, d2] = nearest2KDTree(A, B, bucketSize) (matchFeaturesScratch.m:411-440).  knnsearch on a
    kd-tree is EXACT, so this is the exhaustive device search; d are Euclidean distances (the caller squares
    them, :152-153).  bucketSize only shapes the reference's tree and has no effect on the result."""
    idx1, idx2, d1, d2 = nearest2SSDExhaustive(_pad_dim(A), _pad_dim(B))
    return idx1, idx2, np.sqrt(np.maximum(d1, 0)), np.sqrt(np.maximum(d2, 0))


def nearest2SubsetPdist2(A, B, subset=12000, candB=None, seed=0):
    """[idx1, idx2, d1, d2] = nearest2SubsetPdist2(A, B, subset) (matchFeaturesScratch.m:368-409): exact 2-NN
    (Euclidean) against a random `subset`-row sample of B, indices mapped back into B.  The sample is an
    explicit input here (candB, 1-based) or drawn from `seed`; the reference uses the unseeded randperm (:377)."""
    B = np.asarray(B, np.float32)
    n2 = B.shape[0]
    subset = min(int(subset), n2)
    if candB is None:
        candB = np.random.default_rng(seed).permutation(n2)[:subset] + 1
    candB = np.asarray(candB, np.int64).reshape(-1)
    if candB.size != subset or candB.min() < 1 or candB.max() > n2 or np.unique(candB).size != subset:
        raise ValueError("candB must hold `subset` distinct 1-based rows of B")
    idx1, i2, d1, d2 = nearest2SSDExhaustive(_pad_dim(A), _pad_dim(B[candB - 1]))
    e1, e2 = np.sqrt(np.maximum(d1, 0)), np.sqrt(np.maximum(d2, 0))
    if subset == 1:  # :385-388: a second "neighbour" one ulp away
        e2 = e1 + np.spacing(e1)
    return idx1, candB[i2.astype(np.int64) - 1].astype(np.uint32), e1, e2


def nearest2ApproxFloatFast(A, B, opts=None, return_basis=False):
    """[idx1, idx2, dBest, dSecond] = nearest2ApproxFloatFast(A, B, opts) (matchFeaturesScratch.m:442-573): PCA of B
    to ApproxNumComponents (48) dimensions, both sets centred with B's mean and projected, L2-normalised, then the top
    two cosine similarities per row of A, returned as d = 2 - 2*sim.  All of it on the device (aps_match_pca2nn: column
    means and the covariance B'B through the f32 MFMA path, projection, normalisation, the cosine product and its top
    two); only the 128 x 128 symmetric eigen-problem is solved on the host (inside the library).  opts: 'UsePCA',
    'ApproxNumComponents' as in the reference; BlockRows / UseParfor / UseGPU only partition the reference's loop and
    have no effect on a row's result.  return_basis: also (mu, coeff) of the projection (None when it is skipped)."""
    o = {"ApproxNumComponents": 48, "UsePCA": True}
    o.update(opts or {})
    A, n1, lda, la = _as_desc(A)
    B, n2, ldb, lb = _as_desc(B)
    if la != lb:
        raise ValueError("A and B must share a storage order")
    if n1 == 0 or n2 == 0:
        raise ValueError("Expected input to be nonempty.")
    if A.shape[1] != DIM or B.shape[1] != DIM:
        raise ValueError("Descriptor dimensions must match for non-binary.")
    k = int(o["ApproxNumComponents"])
    idx2 = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    projected = bool(o["UsePCA"]) and DIM > k
    mu = np.zeros(DIM, np.float32) if return_basis and projected else None
    coeff = np.zeros((DIM, k), np.float32) if return_basis and projected else None
    check(lib.aps_match_pca2nn(ptr(A), n1, lda, ptr(B), n2, ldb, DIM, la, k, int(bool(o["UsePCA"])), ptr(idx2), ptr(d1), ptr(d2),
                               ptr(mu) if mu is not None else None, ptr(coeff) if coeff is not None else None))
    out = (np.arange(1, n1 + 1, dtype=np.float64), idx2, d1, d2)
    return out + ((mu, coeff),) if return_basis else out


def filter_matches(idx2, dBest, dSecond, n2, MaxRatio, MatchThreshold, Unique):
    """matchFeaturesScratch.m:170-211 on the host (used by the approximate back ends, whose 2-NN lists live on
    the host): ratio on SSD (r^2), threshold, finiteness, greedy one-to-one by ascending distance (stable)."""
    dBest = np.asarray(dBest, np.float32)
    dSecond = np.asarray(dSecond, np.float32)
    # the reference holds dBest/dSecond in double arrays (inf(N1,1), :344) and evaluates MaxRatio^2 in double
    r2 = float(MaxRatio) * float(MaxRatio)
    b64, s64 = dBest.astype(np.float64), dSecond.astype(np.float64)
    keep = (b64 <= r2 * s64) & (b64 <= float(MatchThreshold)) & np.isfinite(b64) & np.isfinite(s64)
    i1 = np.flatnonzero(keep).astype(np.uint32) + 1
    i2 = np.asarray(idx2, np.uint32)[keep]
    d = dBest[keep]
    if Unique and i1.size:
        order = np.argsort(d, kind="stable")
        i1, i2, d = i1[order], i2[order], d[order]
        used2 = np.zeros(n2 + 1, bool)
        sel = np.zeros(i1.size, bool)
        for k in range(i1.size):  # rows are unique already; first come, first served on the columns
            if not used2[i2[k]]:
                used2[i2[k]] = True
                sel[k] = True
        i1, i2, d = i1[sel], i2[sel], d[sel]
    return np.stack([i1, i2], axis=1), d.astype(np.float32)


def matchFeaturesScratch(F1, F2, Method="Exhaustive", MatchThreshold=3.5, MaxRatio=0.6, Unique=True,
                         ApproxFloatNNMethod="pca2nn", ApproxKDBucketSize=40, candB=None, seed=0,
                         **_ignored_approx_args):
    """[matches, matchMetric] = matchFeaturesScratch(F1, F2, 'Method', ..., 'MatchThreshold', ...,
    'MaxRatio', ..., 'Unique', ...) for float descriptors (matchFeaturesScratch.m:1-215).

    'Exhaustive' (the north-star path): 2-NN + ratio/threshold/uniqueness fused on the device.
    'Approximate' float back ends (:142-160), selected by ApproxFloatNNMethod: 'pca2nn' (PCA-48 + cosine),
    'kdtree' (exact), 'subsetpdist2' (random 12000-row subset of F2; pass candB or seed): their 2-NN search is
    the same device kernel on the transformed data, the filter runs on the host.
    Returns (matches K x 2 uint32 1-based, matchMetric K float32)."""
    method = str(Method).lower()
    if method not in ("exhaustive", "approximate"):
        raise ValueError(f"Unknown Method: {Method}")
    if not (0 < MaxRatio <= 1) or MatchThreshold < 0:
        raise ValueError("invalid MaxRatio/MatchThreshold")
    if method == "approximate":
        An, Bn = _normalize_like_reference(F1, F2)
        if An.shape[0] == 0 or Bn.shape[0] == 0:
            raise ValueError("Expected input to be nonempty.")
        am = str(ApproxFloatNNMethod).lower()
        if am == "pca2nn":
            _, idx2, d1, d2 = nearest2ApproxFloatFast(An, Bn, {"ApproxNumComponents": 48, "UsePCA": True})
        elif am == "kdtree":
            _, idx2, e1, e2 = nearest2KDTree(An, Bn, ApproxKDBucketSize)
            d1, d2 = e1 * e1, e2 * e2  # :152-153
        elif am == "subsetpdist2":
            _, idx2, e1, e2 = nearest2SubsetPdist2(An, Bn, 12000, candB, seed)
            d1, d2 = e1 * e1, e2 * e2  # :156-157
        else:
            raise ValueError("Select a approximate method")
        return filter_matches(idx2, d1, d2, Bn.shape[0], MaxRatio, MatchThreshold, Unique)
    A, n1, ld1, la = _as_desc(F1)
    B, n2, ld2, lb = _as_desc(F2)
    if n1 == 0 or n2 == 0:
        raise ValueError("Expected input to be nonempty.")  # validateattributes(... 'nonempty') :279-280
    if A.shape[1] != B.shape[1]:
        raise ValueError("Descriptor dimensions must match for non-binary.")
    if la != lb:
        B = np.ascontiguousarray(B) if la == _capi.APS_ROWMAJOR else np.asfortranarray(B)
        B, n2, ld2, lb = _as_desc(B)
    cap = n1
    i1 = np.zeros(cap, np.uint32)
    i2 = np.zeros(cap, np.uint32)
    met = np.zeros(cap, np.float32)
    cnt = C.c_int64(0)
    o = _opts(MaxRatio, MatchThreshold, Unique)
    check(lib.aps_match_features(ptr(A), n1, ld1, ptr(B), n2, ld2, A.shape[1], la, C.byref(o), ptr(i1),
                                 ptr(i2), ptr(met), cap, C.byref(cnt)))
    k = cnt.value
    return np.stack([i1[:k], i2[:k]], axis=1), met[:k].copy()


def match_pairwise_csr(allDescriptors, MaxRatio, MatchThreshold, Unique=True, normalize=2,
                       device_out=False):
    """Batched all-pairs matcher: the CSR form of featureMatchingPairwise used by the resident pipeline.
    Returns (pair_ptr int64[P+1], idx_i, idx_j, metric) with pairs in the reference's triu order."""
    n = len(allDescriptors)
    prepared = [_as_desc(d) for d in allDescriptors]
    layouts = {p[3] for p in prepared if p[1] > 0}
    if len(layouts) > 1:
        raise ValueError("all descriptor matrices must share a storage order")
    layout = layouts.pop() if layouts else _capi.APS_ROWMAJOR
    ptrs = (C.c_void_p * n)(*[ptr(p[0]) if p[1] > 0 else None for p in prepared])
    counts = (C.c_int64 * n)(*[p[1] for p in prepared])
    lds = (C.c_int64 * n)(*[max(p[2], DIM if layout == _capi.APS_ROWMAJOR else p[1]) for p in prepared])
    npairs = n * (n - 1) // 2
    pair_ptr = np.zeros(npairs + 1, np.int64)
    o = _opts(MaxRatio, MatchThreshold, Unique, normalize)
    cnt = C.c_int64(0)
    # resident outputs: room for every row of every pair (see match_pairs_csr); host outputs: a modest first capacity,
    # APS_E_CAP reports the exact need
    rows = sum(prepared[i][1] * (n - 1 - i) for i in range(n))  # pair (i, j), i < j, has image i's rows
    cap = max(1, rows if device_out else sum(p[1] for p in prepared) // 4)
    while True:
        if device_out:
            import torch

            i_i = torch.empty(cap, dtype=torch.int32, device="cuda")
            i_j = torch.empty(cap, dtype=torch.int32, device="cuda")
            met = torch.empty(cap, dtype=torch.float32, device="cuda")
            _fence_fresh_blocks()
        else:
            i_i = np.zeros(cap, np.uint32)
            i_j = np.zeros(cap, np.uint32)
            met = np.zeros(cap, np.float32)
        rc = lib.aps_match_pairwise(ptrs, counts, lds, n, DIM, layout, C.byref(o), ptr(pair_ptr),
                                    ptr(i_i), ptr(i_j), ptr(met), cap, C.byref(cnt))
        if rc == _capi.APS_E_CAP:
            cap = cnt.value
            continue
        check(rc)
        break
    k = cnt.value
    return pair_ptr, i_i[:k], i_j[:k], met[:k]


def match_pairs_csr(allDescriptors, pairs, MaxRatio, MatchThreshold, Unique=True, normalize=2, device_out=False):
    """aps_match_pairs: the matcher on an explicit list of (a, b) image pairs (0-based) — the unit that is
    sharded across GPUs.  Returns (pair_ptr int64[P+1], idx_a, idx_b, metric) as numpy arrays, or with
    device_out=True the three match arrays as resident torch tensors (int32, int32, float32)."""
    n = len(allDescriptors)
    prepared = [_as_desc(d) for d in allDescriptors]
    layouts = {p[3] for p in prepared if p[1] > 0}
    if len(layouts) > 1:
        raise ValueError("all descriptor matrices must share a storage order")
    layout = layouts.pop() if layouts else _capi.APS_ROWMAJOR
    ptrs = (C.c_void_p * n)(*[ptr(p[0]) if p[1] > 0 else None for p in prepared])
    counts = (C.c_int64 * n)(*[p[1] for p in prepared])
    lds = (C.c_int64 * n)(*[max(p[2], DIM if layout == _capi.APS_ROWMAJOR else p[1]) for p in prepared])
    if isinstance(pairs, np.ndarray):  # [P, 2] (pair_order_array and slices of it)
        pa = np.ascontiguousarray(pairs[:, 0], np.int32)
        pb = np.ascontiguousarray(pairs[:, 1], np.int32)
    else:
        pa = np.ascontiguousarray([p[0] for p in pairs], np.int32)
        pb = np.ascontiguousarray([p[1] for p in pairs], np.int32)
    P = len(pairs)
    pair_ptr = np.zeros(P + 1, np.int64)
    o = _opts(MaxRatio, MatchThreshold, Unique, normalize)
    cnt = C.c_int64(0)
    rows = int(np.asarray([p[1] for p in prepared], np.int64)[pa].sum()) if P else 0
    # resident outputs: room for every row (uninitialised device memory from torch's cache costs nothing, and a second
    # pass over all pairs after APS_E_CAP costs the whole matcher again - well-overlapping small sets keep more than a
    # sixteenth of their rows); host outputs: a sixteenth, grown on demand
    cap = max(1, rows if device_out else rows // 16)
    while True:
        if device_out:
            import torch

            i_a = torch.empty(cap, dtype=torch.int32, device="cuda")
            i_b = torch.empty(cap, dtype=torch.int32, device="cuda")
            met = torch.empty(cap, dtype=torch.float32, device="cuda")
            _fence_fresh_blocks()
        else:
            i_a = np.zeros(cap, np.uint32)
            i_b = np.zeros(cap, np.uint32)
            met = np.zeros(cap, np.float32)
        rc = lib.aps_match_pairs(ptrs, counts, lds, n, DIM, layout, ptr(pa), ptr(pb), P, C.byref(o), ptr(pair_ptr),
                                 ptr(i_a), ptr(i_b), ptr(met), cap, C.byref(cnt))
        if rc == _capi.APS_E_CAP:
            cap = cnt.value
            continue
        check(rc)
        break
    k = cnt.value
    return pair_ptr, i_a[:k], i_b[:k], met[:k]


_PAIR_ORDER = {}


def pair_order(numImg):
    """The reference's pair order: nonzeros(triu(reshape(1:n^2,n,n),1)) (featureMatchingPairwise.m:48).  (Cached per image
    count: a step asks for it several times; callers do not modify the list.)"""
    numImg = int(numImg)
    if numImg not in _PAIR_ORDER:
        _PAIR_ORDER[numImg] = [(i, j) for j in range(1, numImg) for i in range(j)]
    return _PAIR_ORDER[numImg]


_PAIR_ORDER_ARR = {}


def pair_order_array(numImg):
    """pair_order as a read-only int32 array [P, 2] (cached): match_pairs_csr takes it without walking 2016 tuples."""
    numImg = int(numImg)
    if numImg not in _PAIR_ORDER_ARR:
        a = np.asarray(pair_order(numImg), np.int32).reshape(-1, 2)
        a.setflags(write=False)
        _PAIR_ORDER_ARR[numImg] = a
    return _PAIR_ORDER_ARR[numImg]


def featureMatchingPairwise(input, allDescriptors, numImg):
    """matches = featureMatchingPairwise(input, allDescriptors, numImg) (featureMatchingPairwise.m:1-63).

    Returns an n x n list of lists; cell (i, j), i < j, holds an M x 2 float64 array of 1-based
    [idx_i idx_j]; every other cell is None (MATLAB: empty)."""
    numImg = int(numImg)
    if numImg <= 0 or len(allDescriptors) != numImg:
        raise ValueError("numImg must be positive and match numel(allDescriptors)")
    thr = input.get("Matchingthreshold", 1.5)
    ratio = input.get("Ratiothreshold", 0.6)
    if int(input.get("useMATLABFeatureMatch", 0)) == 1:
        # getMatches :103-107 hands the pair to the toolbox's matchFeatures (closed code): no device counterpart
        raise NotImplementedError("input.useMATLABFeatureMatch = 1 selects the Computer Vision Toolbox's matchFeatures; "
                                  "set it to 0 for the device matcher (the MATLAB overlay forwards this switch to the reference)")
    matches = [[None] * numImg for _ in range(numImg)]
    if str(input.get("Matchingmethod", "Exhaustive")).lower() == "approximate":
        # getMatches :108-117 with Method = 'Approximate': pair by pair through matchFeaturesScratch, as the reference's parfor
        # (:48-63) does; 'pca2nn' = nearest2ApproxFloatFast on the device, 'kdtree' / 'subsetpdist2' the exact device search
        for (i, j) in pair_order(numImg):
            m, _ = matchFeaturesScratch(allDescriptors[i], allDescriptors[j], Method="Approximate", MatchThreshold=thr, MaxRatio=ratio,
                                        Unique=True, ApproxFloatNNMethod=input.get("ApproxFloatNNMethod", "pca2nn"))
            matches[i][j] = np.asarray(m, np.float64).reshape(-1, 2)
        return matches
    pair_ptr, ii, jj, _ = match_pairwise_csr(allDescriptors, ratio, thr, True)
    for p, (i, j) in enumerate(pair_order(numImg)):
        s, e = pair_ptr[p], pair_ptr[p + 1]
        matches[i][j] = np.stack([ii[s:e], jj[s:e]], axis=1).astype(np.float64)
    return matches


# ---- getFeaturePoints (getFeaturePoints.m:1-76) ----------------------------------------------------
def _sift_params(input):
    p = _capi.aps_sift_params()
    p.sigma = float(input.get("Sigma", 1.6))
    p.n_layers = int(input.get("NumLayersInOctave", 4))
    p.contrast_threshold = float(input.get("ContrastThreshold", 0.00133))
    p.edge_threshold = float(input.get("EdgeThreshold", 6))
    p.max_features = int(input.get("maxFeatures", 0))
    return p


def _fence_fresh_blocks():
    """torch's caching allocator recycles a freed block for the next torch.empty at once, which is safe for consumers on
    torch's own stream only: kernels that were queued there before the block was freed (by ANY thread) may still be
    running.  The library writes on its private per-thread streams, so a freshly allocated output buffer is fenced
    against torch's stream before it is handed over (the stream is nearly idle in this pipeline: the wait is short)."""
    import torch

    torch.cuda.current_stream().synchronize()


def sift_extract(input, image, device_out=False, want_aux=False, points_device=False, compact=False):
    """aps_sift_extract with automatic capacity: returns (features, validPts[, aux]).

    Retained memory of a resident result: with device_out the descriptors come back as a VIEW of the capacity
    buffer (H*W/64 rows: 66 MB for a 4K view, ~6x what ~20 k features need), which stays allocated as long as the caller keeps
    the descriptors.  compact=True returns a right-sized clone instead (one device copy per view); pipeline.sift_many asks for
    it when a call extracts more than 96 views, so that a 256- or 500-view set holds what it uses.

    image: H x W x 3 or H x W uint8, numpy (host) or torch (host/device), row-major.
    device_out=True keeps the descriptors on the GPU (torch float32 [n,128]) for the resident pipeline;
    points_device=True (with device_out) leaves the keypoints there as well (torch float64 [n,2]) instead of bringing
    them to the host - the resident pipeline only ever hands them back to the device (aps_gather_match_points)."""
    if _capi.is_torch(image):
        img = image.contiguous()
        h, w = int(img.shape[0]), int(img.shape[1])
        c = 1 if img.dim() == 2 else int(img.shape[2])
    else:
        img = np.ascontiguousarray(image, np.uint8)
        h, w = img.shape[:2]
        c = 1 if img.ndim == 2 else img.shape[2]
    if c not in (1, 3):
        raise ValueError("image must be gray or RGB")
    prm = _sift_params(input)
    cap = max(4096, (h * w) // 64)
    cnt = C.c_int64(0)
    while True:
        if device_out:
            import torch

            desc = torch.empty((cap, DIM), dtype=torch.float32, device="cuda")
            _fence_fresh_blocks()
        else:
            desc = np.zeros((cap, DIM), np.float32)
        if device_out and points_device:
            loc = torch.empty((2, cap), dtype=torch.float64, device="cuda")
            _fence_fresh_blocks()
        else:
            loc = np.zeros((2, cap), np.float64)  # column-major cap x 2
        aux = np.zeros((cap, 4), np.float32) if want_aux else None
        rc = lib.aps_sift_extract(ptr(img), h, w, c, _capi.APS_IMG_U8_HWC, C.byref(prm), ptr(desc),
                                  _capi.APS_ROWMAJOR, DIM, ptr(loc), cap, ptr(aux), cap, C.byref(cnt))
        if rc == _capi.APS_E_CAP and cnt.value > cap:
            cap = int(cnt.value)
            continue
        check(rc)
        break
    n = cnt.value
    dev_pts = device_out and points_device
    if dev_pts:
        check(lib.aps_synchronize())  # (with every output resident the call returns without waiting for its last kernel)
        pts = loc[:, :n].t().contiguous()
        torch.cuda.current_stream().synchronize()  # the (small) transpose ran on torch's stream; consumers run on the library's
    else:
        pts = np.ascontiguousarray(loc[:, :n].T)
    # Resident output: the first n rows of the capacity buffer, as a view.  (Until round 4 a right-sized clone was handed
    # back so that the worst-case buffer - H*W/64 rows = 66 MB for a 4K view - could be freed: a 10 MB device copy per view
    # on torch's stream plus a stream synchronisation, 23 us of a chip-filling copy kernel and the host wait, to save
    # 64 x 56 MB = 3.6 GB of a 288 GB device for the length of one step.)
    if device_out and compact:
        check(lib.aps_synchronize())                # the extraction ran on the library's stream,
        d = desc[:n].clone()                        # the copy runs on torch's,
        torch.cuda.current_stream().synchronize()   # and the consumers on the library's again
    elif device_out:
        d = desc[:n]
    else:
        d = np.ascontiguousarray(desc[:n])
    if want_aux:
        return d, pts, aux[:n].copy()
    return d, pts


def getFeaturePoints(input, ImageOriginal):
    """[features, validPts] = getFeaturePoints(input, ImageOriginal) (getFeaturePoints.m:1-76) for
    input.detector == 'SIFT': features Kf x 128 single (unit norm), validPts Kf x 2 double [x y] 1-based.
    The other detectors of the switch (:33-68) are toolbox calls with no device counterpart here."""
    det = input.get("detector", "SIFT")
    if det != "SIFT":
        if det in ("vl_SIFT", "HARRIS", "FAST", "SURF", "BRISK", "ORB", "KAZE"):
            raise NotImplementedError(f"detector '{det}' is a MATLAB toolbox/VLFeat call; only 'SIFT' runs on the device")
        raise ValueError("Need a valid input!")  # getFeaturePoints.m:67
    return sift_extract(input, ImageOriginal)


# ---- global matcher (featureMatchingGlobal.m) and the mex contracts it calls ------------------------
def flann_knn_win(train, query, k, method="flann", trees=4, checks=32):
    """[idx, dist] = flann_knn_win(train, query, k, 'flann' | 'bf', trees, checks) (PP/mex/flann_knn.cpp:118-253) for
    float (single) and binary (uint8) descriptors, computed EXACTLY on the device (the reference's kd-forest is approximate and
    seed-dependent; `trees`/`checks` are accepted and ignored).  idx: Fq x k uint32 1-based, dist: Fq x k
    single squared-L2, ascending."""
    if k != int(k) or k <= 0:
        raise ValueError("k must be > 0")  # flann_knn:k
    tr = train if _capi.is_torch(train) else np.asarray(train)
    if str(tr.dtype).endswith("uint8"):
        # binary descriptors (flann_knn.cpp:199-223 'bf' = BFMatcher knnMatch; :235-240 'flann' = LSH index): ONE exact
        # brute-force Hamming k-NN serves both; dist = Hamming distance as single, ascending, ties -> lower index
        T = np.ascontiguousarray(train, np.uint8)
        Q = np.ascontiguousarray(query, np.uint8)
        if T.ndim != 2 or Q.ndim != 2:
            raise ValueError("train must be 2-D")  # flann_knn:dim (checkReal2D)
        if T.shape[1] != Q.shape[1]:
            raise ValueError("query must have same descriptor dimension as train")  # flann_knn:dim
        k = int(k)
        idx = np.zeros((Q.shape[0], k), np.uint32)
        dist = np.zeros((Q.shape[0], k), np.float32)
        check(lib.aps_knn_hamming(ptr(T), T.shape[0], T.shape[1], ptr(Q), Q.shape[0], Q.shape[1], T.shape[1],
                                  _capi.APS_ROWMAJOR, k, ptr(idx), ptr(dist), k))
        return idx, dist
    if str(method) == "bf":
        raise ValueError("BFMatcher only supports uint8 (binary) descriptors")  # flann_knn:bf
    T, ft, ldt, lt = _as_desc(train)
    Q, fq, ldq, lq = _as_desc(query)
    if (ft and T.shape[1] != DIM) or (fq and Q.shape[1] != DIM) or lt != lq:
        raise ValueError("query must have same descriptor dimension as train")  # flann_knn:dim
    k = int(k)
    idx = np.zeros((fq, k), np.uint32)
    dist = np.zeros((fq, k), np.float32)
    check(lib.aps_knn_global(ptr(T), ft, ldt, ptr(Q), fq, ldq, DIM, lt, k, ptr(idx), ptr(dist), k))
    return idx, dist


def nearest2HammingExhaustiveMEX(Abytes, Bbytes):
    """[idx2, d1, d2] = nearest2HammingExhaustiveMEX(Abytes, Bbytes) (PP/mex/nearest2HammingExhaustiveMEX.cpp)."""
    A = np.ascontiguousarray(Abytes)
    B = np.ascontiguousarray(Bbytes)
    if A.dtype != np.uint8 or B.dtype != np.uint8:
        raise TypeError("Inputs must be uint8.")  # hamm2nn:type
    if A.ndim != 2 or B.ndim != 2:
        raise ValueError("2D only.")  # hamm2nn:dim
    if A.shape[1] != B.shape[1]:
        raise ValueError("Byte width mismatch.")  # hamm2nn:cols
    n1 = A.shape[0]
    idx2 = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    check(lib.aps_hamming_2nn(ptr(A), n1, A.shape[1], ptr(B), B.shape[0], B.shape[1], A.shape[1],
                              _capi.APS_ROWMAJOR, ptr(idx2), ptr(d1), ptr(d2)))
    return idx2, d1, d2


nearest2HammingExhaustiveOMPMEX = nearest2HammingExhaustiveMEX  # same arithmetic (OMP twin)


def match_global_csr(allDescriptors, ratio=0.6, k=4, device_out=False):
    """featureMatchingGlobal.m:69-161 in CSR form, resident when the descriptors are: pool (torch.cat / concatenate),
    row normalisation with eps inside the root (:83-85, aps_global_normalize), exact k-NN of the pool against itself
    (aps_knn_global: blocked f16-screened search with exact distances), per-query filter (aps_global_filter).
    Returns (pair_ptr int64[P+1] in featureMatchingPairwise's pair order, idx_i, idx_j): 1-based local indices in the
    lower- / higher-numbered image of each pair, query order; device_out=True keeps idx_i/idx_j as int32 CUDA tensors."""
    numImg = len(allDescriptors)
    counts = [int(d.shape[0]) for d in allDescriptors]
    F = sum(counts)
    npairs = numImg * (numImg - 1) // 2
    pair_ptr = np.zeros(npairs + 1, np.int64)
    resident = F > 0 and all(_capi.is_torch(d) and d.is_cuda for d in allDescriptors)
    if F == 0:
        z = np.zeros(0, np.uint32)
        return pair_ptr, z, z.copy()
    img_idx = np.repeat(np.arange(1, numImg + 1, dtype=np.uint32), counts)
    local_idx = np.concatenate([np.arange(1, c + 1, dtype=np.uint32) for c in counts])
    if resident:
        import torch

        raw = torch.cat([d.reshape(-1, DIM) for d in allDescriptors]).contiguous()
        torch.cuda.current_stream().synchronize()  # torch produced the pool; the library reads it on its own stream
        pool = torch.empty_like(raw)
        nn_idx = torch.empty((F, k), dtype=torch.int32, device=raw.device)
        nn_dist = torch.empty((F, k), dtype=torch.float32, device=raw.device)
        oi = torch.empty(F, dtype=torch.int32, device=raw.device)
        oj = torch.empty(F, dtype=torch.int32, device=raw.device)
    else:
        raw = np.ascontiguousarray(np.concatenate([np.asarray(d.cpu() if _capi.is_torch(d) else d, np.float32).reshape(-1, DIM)
                                                   for d in allDescriptors]))
        pool = np.empty_like(raw)
        nn_idx = np.zeros((F, k), np.uint32)
        nn_dist = np.zeros((F, k), np.float32)
        oi = np.zeros(F, np.uint32)
        oj = np.zeros(F, np.uint32)
    check(lib.aps_global_normalize(ptr(raw), F, DIM, DIM, _capi.APS_ROWMAJOR, ptr(pool)))
    if k <= 4:
        # the search with the filter in view: queries that featureMatchingGlobal.m:129-147 provably drops at this ratio are
        # not searched (they come back as copies of themselves, which the filter removes as self matches); every other
        # query gets its exact k nearest - the CSR lists equal those of the plain search
        img_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        check(lib.aps_knn_global_screened(ptr(pool), F, DIM, DIM, _capi.APS_ROWMAJOR, ptr(img_off), numImg, float(ratio), k,
                                          ptr(nn_idx), ptr(nn_dist), k))
    else:
        check(lib.aps_knn_global(ptr(pool), F, DIM, ptr(pool), F, DIM, DIM, _capi.APS_ROWMAJOR, k, ptr(nn_idx), ptr(nn_dist), k))
    cnt = C.c_int64(0)
    check(lib.aps_global_filter(ptr(nn_idx), ptr(nn_dist), F, k, k, _capi.APS_ROWMAJOR, ptr(img_idx), ptr(local_idx),
                                numImg, float(ratio), ptr(pair_ptr), ptr(oi), ptr(oj), F, C.byref(cnt)))
    n = cnt.value
    if resident and not device_out:
        return pair_ptr, oi[:n].cpu().numpy().astype(np.uint32), oj[:n].cpu().numpy().astype(np.uint32)
    return pair_ptr, oi[:n], oj[:n]


def featureMatchingGlobal(input, allDescriptors, numImg):
    """matches = featureMatchingGlobal(input, allDescriptors, numImg) (featureMatchingGlobal.m:1-161): pool all
    descriptors, L2-normalise (:80-86), exact kNN (k = input.k) of the pool against itself, per-query filter
    (drop self / same image, need >= 2, ratio on squared L2), append to the upper-triangular cell in query order."""
    numImg = int(numImg)
    counts = [len(d) for d in allDescriptors]
    matches = [[None] * numImg for _ in range(numImg)]
    F = sum(counts)
    if F == 0:
        return matches
    pair_ptr, oi, oj = match_global_csr(allDescriptors, float(input.get("Ratiothreshold", 0.6)), int(input.get("k", 4)))
    for p, (i, j) in enumerate(pair_order(numImg)):
        s, e = pair_ptr[p], pair_ptr[p + 1]
        if e > s:
            matches[i][j] = np.stack([oi[s:e], oj[s:e]], axis=1).astype(np.float64)
    return matches
