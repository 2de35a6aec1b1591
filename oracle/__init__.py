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
# APS_ORACLE_LIB: another build of the same sources (tests/test_oracle_sanitized.py: `make -C oracle asan`)
LIB_PATH = os.environ.get("APS_ORACLE_LIB") or os.path.join(_HERE, "lib", "libaps_oracle.so")


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


def usable_cpus() -> int:
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (a container can
    report 256 cores through nproc while being throttled to a handful — spinning OpenMP barriers across
    phantom cores are catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // period))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, n)


os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # read by libgomp when the oracle library loads it
lib = _load()
_vp, _i, _i64, _d, _f = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
lib.orc_set_num_threads.argtypes = [C.c_int]
lib.orc_get_max_threads.restype = C.c_int
NUM_THREADS = min(usable_cpus(), 32)
lib.orc_set_num_threads(NUM_THREADS)


def set_num_threads(n: int) -> None:
    global NUM_THREADS
    NUM_THREADS = int(n)
    lib.orc_set_num_threads(int(n))



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
# ---- a6: nearest2ApproxFloatFast (pca_oracle.c) -------------------------------------------------------------------
_orc_pca2nn = _sig("orc_pca2nn", [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
_orc_pca_basis = _sig("orc_pca_basis", [_vp, _i64, _i, _i, _vp, _vp, _vp])


def pca_basis(B, n_components=48):
    """(mu [dim], coeff [dim x n_components], cov [dim x dim] f64) of matchFeaturesScratch.m:480-482."""
    B = _f32(B)
    mu = np.zeros(B.shape[1], np.float32)
    coeff = np.zeros((B.shape[1], n_components), np.float32)
    cov = np.zeros((B.shape[1], B.shape[1]), np.float64)
    _orc_pca_basis(B.ctypes.data, B.shape[0], B.shape[1], int(n_components), mu.ctypes.data, coeff.ctypes.data, cov.ctypes.data)
    return mu, coeff, cov


def pca2nn(A, B, n_components=48, use_pca=True):
    """[idx2 (1-based), d1, d2] of nearest2ApproxFloatFast (matchFeaturesScratch.m:442-573)."""
    A, B = _f32(A), _f32(B)
    n1 = A.shape[0]
    idx = np.zeros(n1, np.uint32)
    d1 = np.zeros(n1, np.float32)
    d2 = np.zeros(n1, np.float32)
    _orc_pca2nn(A.ctypes.data, n1, B.ctypes.data, B.shape[0], A.shape[1], int(n_components), int(bool(use_pca)), idx.ctypes.data,
                d1.ctypes.data, d2.ctypes.data, None, None)
    return idx, d1, d2


_orc_ransac_score = _sig("orc_ransac_score", [_vp, _i, _vp, _vp, _i64, _i64, _d, _vp, _vp, _vp])
_orc_fit_homography = _sig("orc_fit_homography", [_vp, _vp, _i64, _vp, _i64, _vp], _i)
_orc_fit_homography_refit = _sig("orc_fit_homography_refit", [_vp, _vp, _i64, _vp, _i64, _vp], _i)
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


def fit_homography(p1, p2, sel, refit=False):
    """estimateHomography on the selected points; refit=True: with the wave-order sums of the refit on the inliers."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    sel = np.ascontiguousarray(sel, np.int64)
    H = np.zeros(9, np.float64)
    fn = _orc_fit_homography_refit if refit else _orc_fit_homography
    ok = fn(a.ctypes.data, b.ctypes.data, m, sel.ctypes.data, len(sel), H.ctypes.data)
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


TFORM_TYPES = {"projective": 0, "affine": 1, "similarity": 2, "rigid": 3, "translation": 4}
_orc_tform_min_points = _sig("orc_tform_min_points", [_i], _i)
_orc_fit_tform = _sig("orc_fit_tform", [_i, _vp, _vp, _i64, _vp, _i64, _vp], _i)
_orc_ransac_score_tform = _sig("orc_ransac_score_tform", [_i, _vp, _i, _vp, _vp, _i64, _i64, _d, _vp, _vp, _vp])
_orc_ransac_tform = _sig("orc_ransac_tform", [_i, _vp, _vp, _i64, _i64, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp])


def tform_min_points(tform):
    return _orc_tform_min_points(TFORM_TYPES[tform])


def fit_tform(tform, p1, p2, sel):
    """estimateTransform for any transformType on the points listed in sel (0-based).  Returns (H 3x3, finite)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    sel = np.ascontiguousarray(sel, np.int64)
    H = np.zeros(9, np.float64)
    ok = _orc_fit_tform(TFORM_TYPES[tform], a.ctypes.data, b.ctypes.data, m, sel.ctypes.data, len(sel), H.ctypes.data)
    return H.reshape(3, 3).T.copy(), bool(ok)


def ransac_score_tform(tform, Hs, p1, p2, thr, want_mask=True):
    """findInliers for any transformType; Hs: T x 3 x 3.  Returns (n_inl[T], mean_err[T], mask[T, M])."""
    Hs = np.asarray(Hs, np.float64)
    T = Hs.shape[0]
    Hc = np.ascontiguousarray(np.transpose(Hs, (0, 2, 1)))
    a, m = _pts(p1)
    b, _ = _pts(p2)
    n = np.zeros(T, np.int32)
    e = np.zeros(T, np.float64)
    mask = np.zeros((T, max(m, 1)), np.uint8) if want_mask else None
    _orc_ransac_score_tform(TFORM_TYPES[tform], Hc.ctypes.data, T, a.ctypes.data, b.ctypes.data, m, m, float(thr),
                            n.ctypes.data, e.ctypes.data, mask.ctypes.data if want_mask else None)
    return n, e, (mask[:, :m] if want_mask else None)


def ransac_tform(tform, p1, p2, sample_idx, max_distance=5.5, confidence=99.9, max_iter=500):
    """estimateTransformationRANSAC for any transformType.  sample_idx: n_samples x 4, 1-based (the first minPoints
    entries of a row are the sample).  Returns (model 3x3, mask bool[M], found, draws consumed)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    s = np.ascontiguousarray(sample_idx, np.uint32)
    model = np.zeros(9, np.float64)
    mask = np.zeros(max(m, 1), np.uint8)
    found = C.c_int(0)
    trials = C.c_int(0)
    _orc_ransac_tform(TFORM_TYPES[tform], a.ctypes.data, b.ctypes.data, m, m, s.ctypes.data, s.shape[0],
                      float(max_distance), float(confidence), int(max_iter), model.ctypes.data, mask.ctypes.data,
                      C.byref(found), C.byref(trials))
    return model.reshape(3, 3).T.copy(), mask[:m].astype(bool), bool(found.value), trials.value


_orc_fit_tform_mlesac = _sig("orc_fit_tform_mlesac", [_i, _vp, _vp, _i64, _vp, _i64, _vp], _i)
_orc_mlesac_eval_tform = _sig("orc_mlesac_eval_tform", [_i, _vp, _vp, _vp, _i64, _i64, _d, _vp, _vp], _d)
_orc_mlesac_tform = _sig("orc_mlesac_tform", [_i, _vp, _vp, _i64, _i64, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp])


def fit_tform_mlesac(tform, p1, p2, sel):
    """MLESAC's estimator of a transformType (estimateTransformationMLESAC.m:345-510) on the points in sel (0-based)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    sel = np.ascontiguousarray(sel, np.int64)
    H = np.zeros(9, np.float64)
    ok = _orc_fit_tform_mlesac(TFORM_TYPES[tform], a.ctypes.data, b.ctypes.data, m, sel.ctypes.data, len(sel), H.ctypes.data)
    return H.reshape(3, 3).T.copy(), bool(ok)


def mlesac_eval_tform(tform, H, p1, p2, thr):
    a, m = _pts(p1)
    b, _ = _pts(p2)
    Hc = np.ascontiguousarray(np.asarray(H, np.float64).T)
    mask = np.zeros(max(m, 1), np.uint8)
    n = C.c_int(0)
    acc = _orc_mlesac_eval_tform(TFORM_TYPES[tform], Hc.ctypes.data, a.ctypes.data, b.ctypes.data, m, m, float(thr),
                                 mask.ctypes.data, C.byref(n))
    return float(acc), n.value, mask[:m].astype(bool)


def mlesac_tform(tform, p1, p2, sample_idx, max_distance=2.0, confidence=99.9, max_num_trials=1000):
    """estimateTransformationMLESAC for any transformationType.  sample_idx: n_samples x 4, 1-based (the first
    sampleSize entries of a row are the sample).  Returns (model 3x3, mask bool[M], found, draws consumed)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    s = np.ascontiguousarray(sample_idx, np.uint32)
    model = np.zeros(9, np.float64)
    mask = np.zeros(max(m, 1), np.uint8)
    found = C.c_int(0)
    trials = C.c_int(0)
    _orc_mlesac_tform(TFORM_TYPES[tform], a.ctypes.data, b.ctypes.data, m, m, s.ctypes.data, s.shape[0],
                      float(max_distance), float(confidence), int(max_num_trials), model.ctypes.data, mask.ctypes.data,
                      C.byref(found), C.byref(trials))
    return model.reshape(3, 3).T.copy(), mask[:m].astype(bool), bool(found.value), trials.value


_orc_mlesac_homography = _sig("orc_mlesac_homography",
                              [_vp, _vp, _i64, _i64, _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp])
_orc_mlesac_eval = _sig("orc_mlesac_eval", [_vp, _vp, _vp, _i64, _i64, _d, _vp, _vp], _d)
_orc_fit_homography_mlesac = _sig("orc_fit_homography_mlesac", [_vp, _vp, _i64, _vp, _i64, _vp], _i)


def mlesac_homography(p1, p2, sample_idx, max_distance=2.0, confidence=99.9, max_num_trials=1000):
    """estimateTransformationMLESAC (projective).  sample_idx: n_samples x 4, 1-based.
    Returns (model 3x3, mask bool[M], found, draws consumed)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    s = np.ascontiguousarray(sample_idx, np.uint32)
    model = np.zeros(9, np.float64)
    mask = np.zeros(max(m, 1), np.uint8)
    found = C.c_int(0)
    trials = C.c_int(0)
    _orc_mlesac_homography(a.ctypes.data, b.ctypes.data, m, m, s.ctypes.data, s.shape[0],
                           float(max_distance), float(confidence), int(max_num_trials), model.ctypes.data,
                           mask.ctypes.data, C.byref(found), C.byref(trials))
    return model.reshape(3, 3).T.copy(), mask[:m].astype(bool), bool(found.value), trials.value


def mlesac_eval(H, p1, p2, thr):
    """evaluateModel of MLESAC for one model: (accumulated truncated distance, inlier count, mask)."""
    a, m = _pts(p1)
    b, _ = _pts(p2)
    Hc = np.ascontiguousarray(np.asarray(H, np.float64).T)
    mask = np.zeros(max(m, 1), np.uint8)
    n = C.c_int(0)
    acc = _orc_mlesac_eval(Hc.ctypes.data, a.ctypes.data, b.ctypes.data, m, m, float(thr), mask.ctypes.data, C.byref(n))
    return float(acc), n.value, mask[:m].astype(bool)


def fit_homography_mlesac(p1, p2, sel):
    a, m = _pts(p1)
    b, _ = _pts(p2)
    sel = np.ascontiguousarray(sel, np.int64)
    H = np.zeros(9, np.float64)
    ok = _orc_fit_homography_mlesac(a.ctypes.data, b.ctypes.data, m, sel.ctypes.data, len(sel), H.ctypes.data)
    return H.reshape(3, 3).T.copy(), bool(ok)


# ---- render (render_oracle.c) ------------------------------------------------------------------------
class orc_image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("height", C.c_int), ("width", C.c_int), ("channels", C.c_int),
                ("K", C.c_double * 9), ("R", C.c_double * 9), ("gain", C.c_float * 3)]


class orc_canvas(C.Structure):
    _fields_ = [("mode", C.c_int), ("height", C.c_int), ("width", C.c_int), ("f_pan", C.c_double),
                ("origin0", C.c_double), ("origin1", C.c_double), ("R_ref", C.c_double * 9)]


class orc_render_opts(C.Structure):
    _fields_ = [("tile_h", C.c_int), ("tile_w", C.c_int), ("angle_power", C.c_float), ("blending", C.c_int),
                ("pyr_levels", C.c_int), ("pyr_sigma", C.c_float), ("none_policy", C.c_int),
                ("canvas_white", C.c_int)]


_orc_tent = _sig("orc_tent", [_i, _vp])
_orc_warp_tile = _sig("orc_warp_tile", [C.POINTER(orc_image), C.POINTER(orc_canvas), _i, _i, _i, _i, _f,
                                        _vp, _vp, _vp, _vp])
_orc_gaussfilt = _sig("orc_gaussfilt", [_vp, _i, _i, _i, _f, _vp])
_orc_imresize = _sig("orc_imresize", [_vp, _i, _i, _i, _i, _i, _vp])
_orc_multiband_blend = _sig("orc_multiband_blend", [_vp, _vp, _i, _i, _i, _i, _f, _vp])
_orc_linear_blend = _sig("orc_linear_blend", [_vp, _vp, _i, _i, _i, _vp])
_orc_render = _sig("orc_render", [C.POINTER(orc_image), _i, C.POINTER(orc_canvas),
                                  C.POINTER(orc_render_opts), _vp, _vp])
_orc_image_warp_h = _sig("orc_image_warp_h", [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, _f, _i, _vp])
_orc_image_warp_h_m = _sig("orc_image_warp_h_m", [_vp, _i, _i, _i, _vp, _i, _i, _d, _d, _d, _d, _f, _i, _i, _vp])

_MODE_IDS = {"cylindrical": 0, "spherical": 1, "equirectangular": 1, "planar": 2, "perspective": 2,
             "stereographic": 3}
_BLEND_IDS = {"none": 0, "linear": 1, "multiband": 2}
_POLICY_IDS = {"last": 0, "first": 1, "maxangle": 2}


def _images(images, cameras, gains=None):
    n = len(images)
    arr = (orc_image * n)()
    keep = []
    for i, img in enumerate(images):
        t = np.ascontiguousarray(img, np.uint8)
        keep.append(t)
        a = arr[i]
        a.data = t.ctypes.data
        a.height, a.width = t.shape[0], t.shape[1]
        a.channels = 1 if t.ndim == 2 else t.shape[2]
        K = np.asarray(cameras[i]["K"], np.float64)
        R = np.asarray(cameras[i]["R"], np.float64)
        for e in range(9):
            a.K[e] = K[e % 3, e // 3]
            a.R[e] = R[e % 3, e // 3]
        g = (1.0, 1.0, 1.0) if gains is None else gains[i]
        for ch in range(3):
            a.gain[ch] = float(g[ch])
    return arr, keep


def _canvas(geo):
    cv = orc_canvas()
    cv.mode = _MODE_IDS[geo["mode"]]
    cv.height, cv.width = geo["H"], geo["W"]
    cv.f_pan, cv.origin0, cv.origin1 = geo["fPan"], geo["o0"], geo["o1"]
    R = np.asarray(geo["Rref"], np.float64)
    for e in range(9):
        cv.R_ref[e] = R[e % 3, e // 3]
    return cv


def tent(n):
    w = np.zeros(n, np.float32)
    _orc_tent(n, w.ctypes.data)
    return w


def warp_tile(image, camera, geo, r0, c0, ht, wt, angle_power=2.0, gain=(1.0, 1.0, 1.0)):
    arr, keep = _images([image], [camera], [gain])
    cv = _canvas(geo)
    S = np.zeros((ht, wt, 3), np.float32)
    M = np.zeros((ht, wt), np.uint8)
    Wa = np.zeros((ht, wt), np.float32)
    Wf = np.zeros((ht, wt), np.float32)
    _orc_warp_tile(arr, C.byref(cv), r0, c0, ht, wt, float(angle_power), S.ctypes.data, M.ctypes.data,
                   Wa.ctypes.data, Wf.ctypes.data)
    return S, M.astype(bool), Wa, Wf


def gaussfilt(img, sigma):
    a = _f32(img)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    out = np.zeros_like(a)
    _orc_gaussfilt(a.ctypes.data, h, w, c, float(sigma), out.ctypes.data)
    return out


def imresize(img, oh, ow):
    a = _f32(img)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    out = np.zeros((oh, ow) + (() if a.ndim == 2 else (c,)), np.float32)
    _orc_imresize(a.ctypes.data, h, w, c, oh, ow, out.ctypes.data)
    return out


def multiband_blend(C_, W_, levels, sigma=1.0):
    C_ = _f32(C_)
    W_ = _f32(W_)
    K, h, w = W_.shape
    F = np.zeros((h, w, 3), np.float32)
    _orc_multiband_blend(C_.ctypes.data, W_.ctypes.data, K, h, w, int(levels), float(sigma), F.ctypes.data)
    return F


def linear_blend(C_, W_):
    C_ = _f32(C_)
    W_ = _f32(W_)
    K, h, w = W_.shape
    F = np.zeros((h, w, 3), np.float32)
    _orc_linear_blend(C_.ctypes.data, W_.ctypes.data, K, h, w, F.ctypes.data)
    return F


def render(images, cameras, geo, tile, angle_power=2.0, blending="multiband", pyr_levels=3, pyr_sigma=1.0,
           none_policy="last", canvas_white=False, gains=None):
    arr, keep = _images(images, cameras, gains)
    cv = _canvas(geo)
    ro = orc_render_opts(int(tile[0]), int(tile[1]), float(angle_power), _BLEND_IDS[blending], int(pyr_levels),
                         float(pyr_sigma), _POLICY_IDS[none_policy], int(bool(canvas_white)))
    pano = np.zeros((geo["H"], geo["W"], 3), np.uint8)
    cov = np.zeros((geo["H"], geo["W"]), np.uint8)
    _orc_render(arr, len(images), C.byref(cv), C.byref(ro), pano.ctypes.data, cov.ctypes.data)
    return pano, cov


_orc_gain_overlap_stats = _sig("orc_gain_overlap_stats", [_vp, _i, C.POINTER(orc_canvas), _i, _vp, _vp, _vp])


def gain_overlap_stats(images, cameras, geo, stride=5):
    """gainCompensationRKf's overlap statistics: (Nij [N,N], sumCi [N,N,3], sumCj [N,N,3]), upper triangle."""
    arr, keep = _images(images, cameras)
    cv = _canvas(geo)
    n = len(images)
    Nij = np.zeros((n, n), np.float64, order="F")
    sCi = np.zeros((n, n, 3), np.float64, order="F")
    sCj = np.zeros((n, n, 3), np.float64, order="F")
    _orc_gain_overlap_stats(arr, n, C.byref(cv), int(stride), Nij.ctypes.data, sCi.ctypes.data, sCj.ctypes.data)
    del keep
    return np.ascontiguousarray(Nij), np.ascontiguousarray(sCi), np.ascontiguousarray(sCj)


_orc_gain_overlap_stats_warped = _sig("orc_gain_overlap_stats_warped", [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp])


def gain_overlap_stats_warped(Iw, Ww, ds=4):
    """gainCompensationH's overlap statistics of images already warped to one canvas (Iw[k]: H x W x C float32, Ww[k]: H x W
    float32): (Nij [N,N], sumCi [N,N,3], sumCj [N,N,3]), upper triangle."""
    n = len(Iw)
    Iw = [np.ascontiguousarray(np.asarray(a, np.float32)) for a in Iw]
    Ww = [np.ascontiguousarray(np.asarray(a, np.float32)) for a in Ww]
    Iw = [a if a.ndim == 3 else a[..., None] for a in Iw]
    h, w, ch = Iw[0].shape
    pi = (C.c_void_p * n)(*[a.ctypes.data for a in Iw])
    pw = (C.c_void_p * n)(*[a.ctypes.data for a in Ww])
    Nij = np.zeros((n, n), np.float64, order="F")
    sCi = np.zeros((n, n, 3), np.float64, order="F")
    sCj = np.zeros((n, n, 3), np.float64, order="F")
    _orc_gain_overlap_stats_warped(C.addressof(pi), C.addressof(pw), n, h, w, ch, int(ds), Nij.ctypes.data, sCi.ctypes.data,
                                   sCj.ctypes.data)
    return np.ascontiguousarray(Nij), np.ascontiguousarray(sCi), np.ascontiguousarray(sCj)


_orc_imresize_u8 = _sig("orc_imresize_u8", [_vp, _i, _i, _i, _i, _i, _d, _d, _i, _vp])


def imresize_u8(img, scale_or_size, method="bicubic"):
    """imresize(I, s) / imresize(I, [oh ow]) for uint8 images (resizeImagesToLimits.m:57-61,103)."""
    a = np.ascontiguousarray(img, np.uint8)
    sq = a.ndim == 2
    if sq:
        a = a[..., None]
    h, w, c = a.shape
    if np.isscalar(scale_or_size):
        s = float(scale_or_size)
        oh, ow, sr, sc = int(np.ceil(h * s)), int(np.ceil(w * s)), s, s
    else:
        oh, ow = int(scale_or_size[0]), int(scale_or_size[1])
        sr, sc = oh / h, ow / w
    out = np.zeros((oh, ow, c), np.uint8)
    _orc_imresize_u8(a.ctypes.data, h, w, c, oh, ow, sr, sc, 1 if method == "bicubic" else 0, out.ctypes.data)
    return out[..., 0] if sq else out


_orc_ba_pair_blocks = _sig("orc_ba_pair_blocks", [_vp, _vp, C.c_int64, _vp, _i, _vp, _d, _i, _vp])


def ba_pair_blocks(Ui, Uj, pair_ptr, cams, sigma, both=True):
    """The parfor body of accumulateNormalEqnsBlock (bundleAdjustmentRKf.m:717-741): (n_pairs, 59) blocks.
    Ui, Uj: total x 2; cams: (n_pairs, 4, 12) = base i, base j, incremented i, incremented j as (f, cx, cy, R col-major)."""
    Ui = np.asfortranarray(np.asarray(Ui, np.float64).reshape(-1, 2))
    Uj = np.asfortranarray(np.asarray(Uj, np.float64).reshape(-1, 2))
    pp = np.ascontiguousarray(pair_ptr, np.int64)
    cc = np.ascontiguousarray(cams, np.float64)
    out = np.zeros((len(pp) - 1, 59), np.float64)
    _orc_ba_pair_blocks(Ui.ctypes.data, Uj.ctypes.data, Ui.shape[0], pp.ctypes.data, len(pp) - 1, cc.ctypes.data,
                        float(sigma), int(both), out.ctypes.data)
    return out


_orc_crop_rect = _sig("orc_crop_rect", [_vp, C.c_int64, C.c_int64, _i, _d, _vp])
_orc_crop_inside = _sig("orc_crop_inside", [_vp, C.c_int64, C.c_int64, _i, _d, _vp])


def crop_rect(img, canvas_white=False, rng=0.0):
    """panoramaCropper.m:73-157: ((offsetx, offsety, cropW, cropH), valid, (ll, rr, hh, nl))."""
    a = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(9, np.int64)
    _orc_crop_rect(a.ctypes.data, a.shape[0], a.shape[1], int(canvas_white), float(rng), out.ctypes.data)
    return tuple(int(v) for v in out[:4]), bool(out[4]), tuple(int(v) for v in out[5:9])


def crop_inside(img, canvas_white=False, rng=0.0):
    """BW2 = imfill(imbinarize(rgb2gray(I), t), 'holes') of panoramaCropper.m:73-89."""
    a = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(a.shape[:2], np.uint8)
    _orc_crop_inside(a.ctypes.data, a.shape[0], a.shape[1], int(canvas_white), float(rng), out.ctypes.data)
    return out.astype(bool)


def image_warp_h(img, H, out_h, out_w, x0, y0, sx, sy, fill=0.0, method="bilinear"):
    a = np.asarray(img)
    is_u8 = a.dtype == np.uint8
    src = _f32(a if a.ndim == 3 else a[..., None])
    Hc = np.ascontiguousarray(np.asarray(H, np.float64).T)
    out = np.zeros((out_h, out_w, src.shape[2]), np.float32)
    m = {"nearest": 0, "bilinear": 1, "bicubic": 2}[method]
    _orc_image_warp_h_m(src.ctypes.data, src.shape[0], src.shape[1], src.shape[2], Hc.ctypes.data, out_h, out_w,
                        float(x0), float(y0), float(sx), float(sy), float(fill), int(is_u8), m, out.ctypes.data)
    out = out.astype(np.uint8) if is_u8 else out
    return out if a.ndim == 3 else out[..., 0]


# ---- SIFT (sift_oracle.c) ------------------------------------------------------------------------------
class orc_sift_params(C.Structure):
    _fields_ = [("sigma", C.c_double), ("n_layers", C.c_int), ("contrast_threshold", C.c_double),
                ("edge_threshold", C.c_double), ("max_features", C.c_int)]


_orc_sift = _sig("orc_sift", [_vp, _i, _i, _i, C.POINTER(orc_sift_params), _vp, _vp, _vp, _i64], _i64)
_orc_sift_exp = _sig("orc_sift_exp", [_f], _f)
_orc_sift_atan2 = _sig("orc_sift_atan2", [_f, _f], _f)
_orc_sift_sincos = _sig("orc_sift_sincos", [_f, C.POINTER(_f), C.POINTER(_f)])
_orc_sift_blur = _sig("orc_sift_blur", [_vp, _i, _i, _d, _vp])
_orc_sift_num_octaves = _sig("orc_sift_num_octaves", [_i, _i], _i)


def sift(img, sigma=1.6, n_layers=4, contrast_threshold=0.00133, edge_threshold=6.0, cap=None):
    """Returns (desc [n,128] f32 unit-norm, loc [n,2] f64 1-based [x y], aux [n,4])."""
    a = np.ascontiguousarray(img, np.uint8)
    h, w = a.shape[:2]
    c = 1 if a.ndim == 2 else a.shape[2]
    prm = orc_sift_params(float(sigma), int(n_layers), float(contrast_threshold), float(edge_threshold), 0)
    cap = int(cap) if cap else max(4096, (h * w) // 20)
    while True:
        desc = np.zeros((cap, 128), np.float32)
        loc = np.zeros((cap, 2), np.float64)
        aux = np.zeros((cap, 4), np.float32)
        n = _orc_sift(a.ctypes.data, h, w, c, C.byref(prm), desc.ctypes.data, loc.ctypes.data, aux.ctypes.data, cap)
        if n <= cap:
            return desc[:n].copy(), loc[:n].copy(), aux[:n].copy()
        cap = int(n)


def sift_exp(x):
    return float(_orc_sift_exp(float(x)))


def sift_atan2(y, x):
    return float(_orc_sift_atan2(float(y), float(x)))


def sift_sincos(a):
    s, c = C.c_float(0), C.c_float(0)
    _orc_sift_sincos(float(a), C.byref(s), C.byref(c))
    return s.value, c.value


def sift_blur(img, sigma):
    a = _f32(img)
    out = np.zeros_like(a)
    _orc_sift_blur(a.ctypes.data, a.shape[0], a.shape[1], float(sigma), out.ctypes.data)
    return out


def sift_num_octaves(h, w):
    return int(_orc_sift_num_octaves(int(h), int(w)))


# ---- canvas geometry (bounds_oracle.c) -----------------------------------------------------------------------------
_GEO_MODES = {"cylindrical": 0, "spherical": 1, "equirectangular": 1, "planar": 2, "perspective": 2, "stereographic": 3}


class _GeoOpts(C.Structure):
    _fields_ = [("f_pan", _d), ("res_scale", _d), ("margin", _d), ("max_megapixel", _d), ("pct_lo", _d), ("pct_hi", _d),
                ("uv_abs_cap", _d), ("pixel_pad", _d), ("auto_ref", _i)]


def _cams_flat(cameras, sizes):
    K = np.ascontiguousarray(np.stack([np.asarray(c["K"], np.float64).T.reshape(-1) for c in cameras]))  # column-major
    R = np.ascontiguousarray(np.stack([np.asarray(c["R"], np.float64).T.reshape(-1) for c in cameras]))
    S = np.ascontiguousarray(np.asarray([[s[0], s[1]] for s in sizes], np.float64))
    return K, R, S


def bounds(mode, cameras, sizes, Rref=None, robust_pct=(1, 99), abs_cap=8.0):
    """(aMin, aMax, bMin, bMax) of cylindricalBounds / sphericalBounds / planarBounds / stereographicBounds."""
    K, R, S = _cams_flat(cameras, sizes)
    rr = np.ascontiguousarray(np.asarray(np.eye(3) if Rref is None else Rref, np.float64).T.reshape(-1))
    out = np.zeros(4)
    lib.orc_bounds.argtypes = [_i, _i, _vp, _vp, _vp, _vp, _d, _d, _d, _vp]
    lib.orc_bounds(_GEO_MODES[str(mode).lower()], len(cameras), K.ctypes.data, R.ctypes.data, S.ctypes.data, rr.ctypes.data,
                   float(robust_pct[0]), float(robust_pct[1]), float(abs_cap), out.ctypes.data)
    return tuple(out.tolist())


def canvas_geometry(cameras, sizes, mode, ref_idx, opts):
    """renderPanorama.m:84-232: dict(W, H, o0, o1, refIdx, resScale)."""
    K, R, S = _cams_flat(cameras, sizes)
    o = _GeoOpts(float(opts["fPan"]), float(opts["resScale"]), float(opts["margin"]), float(opts["maxMegapixel"]),
                 float(opts["robustPct"][0]), float(opts["robustPct"][1]), float(opts["uvAbsCap"]), float(opts["pixelPad"]),
                 int(bool(opts["autoRef"])))
    ref = C.c_int(int(ref_idx))
    out = np.zeros(5)
    lib.orc_canvas_geometry.argtypes = [_i, _i, _vp, _vp, _vp, C.POINTER(C.c_int), C.POINTER(_GeoOpts), _vp]
    lib.orc_canvas_geometry(_GEO_MODES[str(mode).lower()], len(cameras), K.ctypes.data, R.ctypes.data, S.ctypes.data,
                            C.byref(ref), C.byref(o), out.ctypes.data)
    return {"W": int(out[0]), "H": int(out[1]), "o0": float(out[2]), "o1": float(out[3]), "refIdx": int(ref.value),
            "resScale": float(out[4])}


def crop_nonzero_bbox(img, canvas_white=False):
    """cropNonzeroBbox (renderPanorama.m:1459-1504): ((r1, r2, c1, c2) 1-based inclusive, didCrop)."""
    a = np.ascontiguousarray(img, np.uint8)
    rect = np.zeros(4, np.int64)
    lib.orc_crop_nonzero_bbox.argtypes = [_vp, _i64, _i64, _i, _vp]
    lib.orc_crop_nonzero_bbox.restype = _i
    did = lib.orc_crop_nonzero_bbox(a.ctypes.data, a.shape[0], a.shape[1], int(bool(canvas_white)), rect.ctypes.data)
    return tuple(int(v) for v in rect), bool(did)


_orc_knn_hamming = _sig("orc_knn_hamming", [_vp, _i64, _vp, _i64, _i, _i, _vp, _vp])


def knn_hamming(train, query, k):
    """Exact Hamming k-NN for packed binary descriptors (the uint8 branches of flann_knn.cpp): (idx 1-based, dist)."""
    t = np.ascontiguousarray(train, np.uint8)
    q = np.ascontiguousarray(query, np.uint8)
    idx = np.zeros((q.shape[0], k), np.uint32)
    dist = np.zeros((q.shape[0], k), np.float32)
    _orc_knn_hamming(t.ctypes.data, t.shape[0], q.ctypes.data, q.shape[0], t.shape[1], int(k), idx.ctypes.data, dist.ctypes.data)
    return idx, dist
