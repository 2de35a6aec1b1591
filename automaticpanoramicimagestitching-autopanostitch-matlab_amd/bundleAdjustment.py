"""Host-side mirror of the normal-equation accumulation of PP/bundleAdjustment/bundleAdjustmentRKf.m
(SURVEY.md section 8(f) rank 3).

The per-pair blocks - the reference's `parfor p = 1:numel(pairList)` body (:717-741) - run on the device through
`aps_ba_pair_blocks`; what stays here is the bookkeeping around it, restated from the reference: buildDeltaVector
(:1360-1405), applyIncrements (:1407-1501), the pair list of accumulateNormalEqnsBlock (:680-708) and its serial
reduction into H and g (:743-789).  The Levenberg-Marquardt loop, the Brown-Lowe prior and the solve are the caller's
(they are O(P^2..P^3) in the camera count, not in the match count)."""
from __future__ import annotations

import numpy as np

from ._capi import check, lib, ptr


def skewSymmetric(v):
    v = np.asarray(v, np.float64).reshape(3)
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], np.float64)


def buildDeltaVector(cameras, camList, seed):
    """[Phi, pmap] = buildDeltaVector(cameras, camList, seed): one parameter (df) for the seed camera, four
    ([dthx dthy dthz df]) for every other camera; indices 0-based here."""
    pmap, idx = [], 0
    for i in camList:
        is_seed = i == seed
        pmap.append({"camIdx": i, "startIdx": idx, "isSeed": is_seed})
        idx += 1 if is_seed else 4
    return np.zeros(idx, np.float64), pmap


def _cxcy(cam):
    if cam.get("cx") is not None:
        return float(cam["cx"]), float(cam["cy"])
    if cam.get("K") is not None:
        return float(cam["K"][0, 2]), float(cam["K"][1, 2])
    return 0.0, 0.0


def applyIncrements(cameras, Phi, pmap):
    """camsOut = applyIncrements(cameras, Phi, pmap, ...) (:1407-1501): R <- exp([dth]x) R (Rodrigues, first order
    below 1e-12), f <- clamp(f + df, 100, 5000) when it moves by more than 1e-9."""
    out = [None if c is None else dict(c) for c in cameras]
    for e in pmap:
        i, s = e["camIdx"], e["startIdx"]
        cam = out[i]
        cam["cx"], cam["cy"] = _cxcy(cam)
        if e["isSeed"]:
            df = Phi[s]
        else:
            dth = np.asarray(Phi[s:s + 3], np.float64)
            df = Phi[s + 3]
            a = float(np.sqrt(np.sum(dth * dth)))
            if a < 1e-12:
                Rupd = np.eye(3) + skewSymmetric(dth)
            else:
                K = skewSymmetric(dth / a)
                Rupd = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
            cam["R"] = Rupd @ np.asarray(cam["R"], np.float64)
        oldf = float(cam["f"])
        f = max(100.0, min(5000.0, oldf + float(df)))
        if abs(f - oldf) > 1e-9:
            cam["f"] = f
            Ki = np.eye(3) if cam.get("K") is None else np.array(cam["K"], np.float64)
            Ki[0, 0] = Ki[1, 1] = f
            cam["K"] = Ki
    return out


def _pack_cam(cam):
    cx, cy = _cxcy(cam)
    return np.concatenate([[float(cam["f"]), cx, cy], np.asarray(cam["R"], np.float64).reshape(3, 3).ravel(order="F")])


def ba_pair_blocks(Ui, Uj, pair_ptr, cams, sigmaHuber, both=True):
    """The device call: (n_pairs, 59) = Hii, Hjj, Hij (4 x 4 column-major), gi, gj, E, r2sum, rcnt per pair."""
    Ui = np.asfortranarray(np.asarray(Ui, np.float64).reshape(-1, 2))
    Uj = np.asfortranarray(np.asarray(Uj, np.float64).reshape(-1, 2))
    pp = np.ascontiguousarray(pair_ptr, np.int64)
    cc = np.ascontiguousarray(cams, np.float64)
    n = len(pp) - 1
    if cc.shape != (n, 4, 12):
        raise ValueError("cams must be (n_pairs, 4, 12)")
    out = np.zeros((n, 59), np.float64)
    check(lib.aps_ba_pair_blocks(ptr(Ui), ptr(Uj), Ui.shape[0], ptr(pp), n, ptr(cc), float(sigmaHuber), int(bool(both)),
                                 ptr(out)))
    return out


def accumulateNormalEqnsBlock(Phi, pmap, baseCams, camList, seed, matches, keypoints, imageSizes, sigmaHuber, opts=None,
                              blocks=ba_pair_blocks):
    """[H, g, E, rmse] = accumulateNormalEqnsBlock(...) (:609-791).  `matches[i][j]` (i < j) is an M x 2 array of
    1-based keypoint indices as in the reference (None/empty when there is no edge); cameras are dicts with f, R and K
    or cx/cy; indices are 0-based.  opts.MaxMatches subsampling (:1047-1358) is the caller's: pass the subsampled
    matches.  H is returned dense (P x P, P <= 4 N)."""
    opts = opts or {}
    if opts.get("MaxMatches") is not None and np.isfinite(opts["MaxMatches"]):
        raise NotImplementedError("subsample the matches before the call (subsampleMatches is host-side bookkeeping)")
    camLin = applyIncrements(baseCams, Phi, pmap)
    last = pmap[-1]
    P = last["startIdx"] + (1 if last["isSeed"] else 4)
    cols = {e["camIdx"]: (np.arange(e["startIdx"], e["startIdx"] + (1 if e["isSeed"] else 4))) for e in pmap}
    pairs, Ui, Uj, ptrs, cams = [], [], [], [0], []
    for a, i in enumerate(camList):
        for j in camList[a + 1:]:
            mp = matches[i][j] if matches[i] is not None else None
            if mp is None or len(mp) == 0:
                continue
            mp = np.asarray(mp, np.int64)
            Ui.append(np.asarray(keypoints[i], np.float64)[mp[:, 0] - 1])
            Uj.append(np.asarray(keypoints[j], np.float64)[mp[:, 1] - 1])
            ptrs.append(ptrs[-1] + len(mp))
            cams.append(np.stack([_pack_cam(baseCams[i]), _pack_cam(baseCams[j]), _pack_cam(camLin[i]), _pack_cam(camLin[j])]))
            pairs.append((i, j))
    H = np.zeros((P, P), np.float64)
    g = np.zeros(P, np.float64)
    if not pairs:
        return H, g, 0.0, 0.0
    out = blocks(np.concatenate(Ui), np.concatenate(Uj), ptrs, np.stack(cams), sigmaHuber, not opts.get("OneDirection", False))
    E = R2 = cnt = 0.0
    for (i, j), o in zip(pairs, out):  # the serial reduction of :743-785, pair by pair
        bi, bj = cols[i], cols[j]
        Hii = o[0:16].reshape(4, 4, order="F")[:len(bi), :len(bi)]
        Hjj = o[16:32].reshape(4, 4, order="F")[:len(bj), :len(bj)]
        Hij = o[32:48].reshape(4, 4, order="F")[:len(bi), :len(bj)]
        H[np.ix_(bi, bi)] += Hii
        H[np.ix_(bj, bj)] += Hjj
        H[np.ix_(bi, bj)] += Hij
        H[np.ix_(bj, bi)] += Hij.T
        g[bi] += o[48:48 + len(bi)]
        g[bj] += o[52:52 + len(bj)]
        E += o[56]
        R2 += o[57]
        cnt += o[58]
    return H, g, E, float(np.sqrt(max(R2, 0.0) / max(cnt, 1.0)))
