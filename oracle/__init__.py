"""CPU oracle (TEST INFRASTRUCTURE ONLY) — ctypes access to oracle/lib/libaps_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only
as the checker / reported baseline.  The product package never imports it.
PARITY UNPINNED: see the header of each C file — the reference has no tests or golden vectors and its
toolbox/OpenCV dependencies are not available here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libaps_oracle.so")


def build(force: bool = False) -> None:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


def _load():
    if not os.path.exists(LIB_PATH):
        build()
    return C.CDLL(LIB_PATH)


lib = _load()
_vp, _i, _i64, _d, _f = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float


def _sig(name, argtypes, restype=None):
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = restype
    return fn


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


_orc_normalize_rows = _sig("orc_normalize_rows", [_vp, _i64, _i])
_orc_match_2nn_ssd = _sig("orc_match_2nn_ssd", [_vp, _i64, _vp, _i64, _i, _vp, _vp, _vp])
_orc_match_features = _sig("orc_match_features",
                           [_vp, _i64, _vp, _i64, _i, _d, _d, _i, _i, _vp, _vp, _vp], _i64)
_orc_knn = _sig("orc_knn", [_vp, _i64, _vp, _i64, _i, _i, _vp, _vp])
_orc_global_filter = _sig("orc_global_filter", [_vp, _vp, _i64, _i, _vp, _vp, _d, _vp], _i64)
_orc_hamming_2nn = _sig("orc_hamming_2nn", [_vp, _i64, _vp, _i64, _i, _vp, _vp, _vp])


def normalize_rows(X):
    X = _f32(X).copy()
    _orc_normalize_rows(X.ctypes.data, X.shape[0], X.shape[1])
    return X


def match_2nn_ssd(A, B):
    A, B = _f32(A), _f32(B)
    n1, n2 = A.shape[0], B.shape[0]
    dim = A.shape[1] if n1 else B.shape[1]
    idx = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    _orc_match_2nn_ssd(A.ctypes.data, n1, B.ctypes.data, n2, dim, idx.ctypes.data, d1.ctypes.data,
                       d2.ctypes.data)
    return idx, d1, d2


def match_features(F1, F2, max_ratio=0.6, match_threshold=3.5, unique=True, normalize=2):
    F1, F2 = _f32(F1), _f32(F2)
    n1, n2 = F1.shape[0], F2.shape[0]
    o1 = np.zeros(max(n1, 1), np.uint32)
    o2 = np.zeros(max(n1, 1), np.uint32)
    met = np.zeros(max(n1, 1), np.float32)
    k = _orc_match_features(F1.ctypes.data, n1, F2.ctypes.data, n2, F1.shape[1], float(max_ratio),
                            float(match_threshold), int(bool(unique)), int(normalize),
                            o1.ctypes.data, o2.ctypes.data, met.ctypes.data)
    return np.stack([o1[:k], o2[:k]], axis=1), met[:k].copy()


def knn(train, query, k):
    train, query = _f32(train), _f32(query)
    fq = query.shape[0]
    idx = np.zeros((fq, k), np.uint32)
    dist = np.zeros((fq, k), np.float32)
    _orc_knn(train.ctypes.data, train.shape[0], query.ctypes.data, fq, train.shape[1], k,
             idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def global_filter(nn_idx, nn_dist, img_idx, local_idx, ratio):
    nn_idx = np.ascontiguousarray(nn_idx, np.uint32)
    nn_dist = _f32(nn_dist)
    img_idx = np.ascontiguousarray(img_idx, np.uint32)
    local_idx = np.ascontiguousarray(local_idx, np.uint32)
    f, k = nn_idx.shape
    out = np.zeros((max(f, 1), 4), np.uint32)
    n = _orc_global_filter(nn_idx.ctypes.data, nn_dist.ctypes.data, f, k, img_idx.ctypes.data,
                           local_idx.ctypes.data, float(ratio), out.ctypes.data)
    return out[:n].copy()


def hamming_2nn(A, B):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    n1 = A.shape[0]
    idx = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    _orc_hamming_2nn(A.ctypes.data, n1, B.ctypes.data, B.shape[0], A.shape[1], idx.ctypes.data,
                     d1.ctypes.data, d2.ctypes.data)
    return idx, d1, d2


# ---- RANSAC (ransac_oracle.c) -------------------------------------------------------------------
_orc_ransac_score = _sig("orc_ransac_score", [_vp, _i, _vp, _vp, _i64, _i64, _d, _vp, _vp, _vp])
_orc_fit_homography = _sig("orc_fit_homography", [_vp, _vp, _i64, _vp, _i64, _vp], _i)
_orc_check_model = _sig("orc_check_model", [_vp], _i)
_orc_ransac_homography = _sig("orc_ransac_homography",
                              [_vp, _vp, _i64, _i64, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp])


def _pts(p):
    """M x 2 [x y] -> column-major buffer (x column then y column), returns (array, ldp)."""
    p = np.asarray(p, np.float64)
    return np.asfortranarray(p), p.shape[0]


def ransac_score(Hs, p1, p2, thr, want_mask=True):
    """Hs: T x 3 x 3 (usual row/col indexing); returns (n_inl[T], mean_err[T], mask[T, M])."""
    Hs = np.asarray(Hs, np.float64)
    T = Hs.shape[0]
    Hc = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))  # each page column-major
    a, m = _pts(p1)
    b, _ = _pts(p2)
    n = np.zeros(T, np.int32)
    e = np.zeros(T, np.float64)
    mask = np.zeros((T, m), np.uint8) if want_mask else None
    _orc_ransac_score(Hc.ctypes.data, T, a.ctypes.data, b.ctypes.data, m, m, float(thr), n.ctypes.data,
                      e.ctypes.data, mask.ctypes.data if want_mask else None)
    return n, e, mask


def fit_homography(p1, p2, sel):
    a, m = _pts(p1)
    b, _ = _pts(p2)
    sel = np.ascontiguousarray(sel, np.int64)
    H = np.zeros(9, np.float64)
    ok = _orc_fit_homography(a.ctypes.data, b.ctypes.data, m, sel.ctypes.data, len(sel), H.ctypes.data)
    return H.reshape(3, 3).T.copy(), bool(ok)


def check_model(H):
    Hc = np.ascontiguousarray(np.asarray(H, np.float64).T)
    return bool(_orc_check_model(Hc.ctypes.data))


def ransac_homography(p1, p2, sample_idx, max_distance=5.5, confidence=99.9, max_iter=500):
    """sample_idx: n_samples x 4, 1-based.  Returns (model 3x3, mask bool[M], found, trials)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    s = np.ascontiguousarray(sample_idx, np.uint32)
    model = np.zeros(9, np.float64)
    mask = np.zeros(max(m, 1), np.uint8)
    found = C.c_int(0)
    trials = C.c_int(0)
    _orc_ransac_homography(a.ctypes.data, b.ctypes.data, m, m, s.ctypes.data, s.shape[0],
                           float(max_distance), float(confidence), int(max_iter), model.ctypes.data,
                           mask.ctypes.data, C.byref(found), C.byref(trials))
    return model.reshape(3, 3).T.copy(), mask[:m].astype(bool), bool(found.value), trials.value
