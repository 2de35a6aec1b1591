// ransac.hip — batched RANSAC / MLESAC on gfx950 (f64).
//
// Restates PP/imageMatching/estimateTransformationRANSAC.m:54-183 (loop), :188-225 (normalised DLT),
// :444-516 (findInliers), :518-535 (checkModel), :537-574 (isDegenerate), :579-610 (normalizePoints)
// for transformType 'projective', and :227-452, :483-497 for 'affine' / 'similarity' / 'rigid' / 'translation' (the section
// "The other transformTypes" below), batched over the candidate image pairs of PP/imageMatching/imageMatching.m:121-156.
//
// Shape of the computation (why it is batched this way):
//   fit kernel    : one lane per (pair, draw): 4-point normalised DLT.  The right null vector of the
//                   8x9 system is the smallest eigenvector of the 9x9 Gram matrix (cyclic Jacobi); the
//                   two 9x9 work matrices of every lane live in LDS as [element][lane].
//   score kernel  : one 64-lane wave per (pair, draw): lanes stride over the pair's matches, symmetric
//                   transfer error in f64, inlier count / error sum / centroid by wavefront butterfly
//                   reductions, then the collinearity (degeneracy) test on the inliers.
//   replay (host) : the data-dependent part of the loop — best-so-far update and the adaptive shrink of
//                   maxTrials (:115-130) — is replayed sequentially over the pre-scored draws, which is
//                   exactly the sequential algorithm on the same draws.
//   finalize      : one wave per pair: inlier mask of the winning draw, refit on all inliers (:146-150),
//                   re-score, fallback rules (:153-176).
// All reductions use one fixed order (lane-strided partial sums + xor butterfly) that the oracle
// restates, so inlier masks compare bit-exactly.  f64 only; no contraction (-ffp-contract=off).
#include <cmath>
#include <vector>

#include "aps_internal.h"

namespace aps {

constexpr double kDblEps = 2.220446049250313e-16;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

struct Mat3 {
    double m[9];  // column-major: m[r + 3c]
};
#define M3(H, r, c) (H).m[(r) + 3 * (c)]

__device__ __forceinline__ Mat3 adjugate3(const Mat3& H) {
    Mat3 A;
    M3(A, 0, 0) = M3(H, 1, 1) * M3(H, 2, 2) - M3(H, 1, 2) * M3(H, 2, 1);
    M3(A, 0, 1) = M3(H, 0, 2) * M3(H, 2, 1) - M3(H, 0, 1) * M3(H, 2, 2);
    M3(A, 0, 2) = M3(H, 0, 1) * M3(H, 1, 2) - M3(H, 0, 2) * M3(H, 1, 1);
    M3(A, 1, 0) = M3(H, 1, 2) * M3(H, 2, 0) - M3(H, 1, 0) * M3(H, 2, 2);
    M3(A, 1, 1) = M3(H, 0, 0) * M3(H, 2, 2) - M3(H, 0, 2) * M3(H, 2, 0);
    M3(A, 1, 2) = M3(H, 0, 2) * M3(H, 1, 0) - M3(H, 0, 0) * M3(H, 1, 2);
    M3(A, 2, 0) = M3(H, 1, 0) * M3(H, 2, 1) - M3(H, 1, 1) * M3(H, 2, 0);
    M3(A, 2, 1) = M3(H, 0, 1) * M3(H, 2, 0) - M3(H, 0, 0) * M3(H, 2, 1);
    M3(A, 2, 2) = M3(H, 0, 0) * M3(H, 1, 1) - M3(H, 0, 1) * M3(H, 1, 0);
    return A;
}

__device__ __forceinline__ double det3(const Mat3& H) {
    const double c0 = M3(H, 1, 1) * M3(H, 2, 2) - M3(H, 1, 2) * M3(H, 2, 1);
    const double c1 = M3(H, 1, 0) * M3(H, 2, 2) - M3(H, 1, 2) * M3(H, 2, 0);
    const double c2 = M3(H, 1, 0) * M3(H, 2, 1) - M3(H, 1, 1) * M3(H, 2, 0);
    return (M3(H, 0, 0) * c0 - M3(H, 0, 1) * c1) + M3(H, 0, 2) * c2;
}

__device__ __forceinline__ double norm1_3(const Mat3& H) {
    double m = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double s = (fabs(M3(H, 0, c)) + fabs(M3(H, 1, c))) + fabs(M3(H, 2, c));
        if (s > m) m = s;
    }
    return m;
}

// checkModel (estimateTransformationRANSAC.m:518-535)
__device__ __forceinline__ bool check_model(const Mat3& H) {
#pragma unroll
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    const double d = det3(H);
    if (!(fabs(d) > kDblEps)) return false;
    const Mat3 A = adjugate3(H);
    Mat3 I;
#pragma unroll
    for (int e = 0; e < 9; ++e) I.m[e] = A.m[e] / d;
    const double rc = 1.0 / (norm1_3(H) * norm1_3(I));
    return rc > kDblEps;
}

// LDS work matrices: element e of lane l at [e*S + l] (S = 64 in the fit kernel, one problem per lane; S = 2 in the
// refit, one problem per workgroup)
#define GE(p, q) sG[((p) * 9 + (q)) * S + lane]
#define VE(p, q) sV[((p) * 9 + (q)) * S + lane]

// Cyclic Jacobi on the symmetric 9x9 in GE; eigenvectors in the columns of VE.
// (N x N problem in the 9-stride layout; N = 9 is the homography, 7 / 5 MLESAC's affine / similarity systems)
template <int S, int N = 9>
__device__ void jacobi9(double* sG, double* sV, int lane) {
    for (int p = 0; p < N; ++p)
        for (int q = 0; q < N; ++q) VE(p, q) = (p == q) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double gpq = GE(p, q);
                const double gpp = GE(p, p), gqq = GE(q, q);
                if (fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq))) continue;
                rotated = true;
                const double theta = (gqq - gpp) / (2.0 * gpq);
                const double t =
                    (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < N; ++k) {
                    if (k == p || k == q) continue;
                    const double gkp = GE(k, p), gkq = GE(k, q);
                    const double np_ = c * gkp - s * gkq;
                    const double nq_ = s * gkp + c * gkq;
                    GE(k, p) = np_;
                    GE(p, k) = np_;
                    GE(k, q) = nq_;
                    GE(q, k) = nq_;
                }
                GE(p, p) = gpp - t * gpq;
                GE(q, q) = gqq + t * gpq;
                GE(p, q) = 0.0;
                GE(q, p) = 0.0;
                for (int k = 0; k < N; ++k) {
                    const double vkp = VE(k, p), vkq = VE(k, q);
                    VE(k, p) = c * vkp - s * vkq;
                    VE(k, q) = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
}

struct Norm {
    double s, tx, ty;
    double cx, cy;  // the centroid: MLESAC's normalised points are (p - centroid) * s (normalizePointsHartleyZisserman
                    // :667-671), RANSAC's are T * [p; 1] = s * p + t (normalizePoints :604-606)
};
__device__ __forceinline__ double norm_coord(const Norm& n, double p, double c, double t, int mlesac) {
    return mlesac ? (p - c) * n.s : n.s * p + t;
}

// Row `half` (0: x-row, 1: y-row) of the DLT matrix for one normalised correspondence (:209-212)
__device__ __forceinline__ double dlt_entry(int k, int half, double x, double y, double u, double v) {
    const double w = half ? v : u;
    if (k >= 6) return k == 6 ? x * w : (k == 7 ? y * w : w);
    const int kk = half ? k - 3 : k;
    if (kk < 0 || kk > 2) return 0.0;
    return kk == 0 ? -x : (kk == 1 ? -y : -1.0);
}

// From the filled Gram matrix to the denormalised H (:214-224).  Returns false if not finite.
template <int S>
__device__ bool gram_to_h(double* sG, double* sV, int lane, const Norm& n1, const Norm& n2, Mat3& H, int mlesac = 0) {
    for (int p = 0; p < 9; ++p)
        for (int q = 0; q < p; ++q) GE(p, q) = GE(q, p);
    jacobi9<S>(sG, sV, lane);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (GE(k, k) < GE(kmin, kmin)) kmin = k;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = VE(k, kmin);
    Mat3 Hn, M;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M3(Hn, r, c) = h[3 * r + c] / h[8];
    for (int c = 0; c < 3; ++c) {
        const double m2 = M3(Hn, 2, c);
        M3(M, 2, c) = m2;
        M3(M, 1, c) = (M3(Hn, 1, c) - n2.ty * m2) / n2.s;
        M3(M, 0, c) = (M3(Hn, 0, c) - n2.tx * m2) / n2.s;
    }
    for (int r = 0; r < 3; ++r) {
        M3(H, r, 0) = M3(M, r, 0) * n1.s;
        M3(H, r, 1) = M3(M, r, 1) * n1.s;
        M3(H, r, 2) = (M3(M, r, 0) * n1.tx + M3(M, r, 1) * n1.ty) + M3(M, r, 2);
    }
    if (mlesac) {  // denormalizeTform: tform ./ tform(end) (estimateTransformationMLESAC.m:713-714)
        const double d = H.m[8];
        for (int e = 0; e < 9; ++e) H.m[e] = H.m[e] / d;
    }
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    return true;
}

// The finalize kernel's form of jacobi9 / gram_to_h: ONE 9x9 problem per 64-lane workgroup (column 0 of the S = 2
// layout).  A rotation (p, q) touches rows/columns k = 0..8 independently, so lane k < 9 updates "its" k while every lane
// evaluates the (uniform) rotation parameters: the same operations on the same operands as the serial routine - bit
// for bit - in a ninth of the dependent LDS round trips (the serial form by lane 0 was most of the kernel's 3.5 ms).
template <int N = 9>
__device__ void jacobi9_wave(double* sG, double* sV, int lane) {
#define G2(p, q) sG[((p) * 9 + (q)) * 2]
#define V2(p, q) sV[((p) * 9 + (q)) * 2]
    const int k = lane;
    if (k < N)
        for (int q = 0; q < N; ++q) V2(k, q) = (k == q) ? 1.0 : 0.0;
    __syncthreads();
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double gpq = G2(p, q);
                const double gpp = G2(p, p), gqq = G2(q, q);
                if (fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq))) continue;  // uniform
                rotated = true;
                const double theta = (gqq - gpp) / (2.0 * gpq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                double gkp = 0, gkq = 0, vkp = 0, vkq = 0;
                if (k < N) {
                    gkp = G2(k, p);
                    gkq = G2(k, q);
                    vkp = V2(k, p);
                    vkq = V2(k, q);
                }
                __syncthreads();  // every read of this rotation before any of its writes
                if (k < N) {
                    if (k != p && k != q) {
                        const double np_ = c * gkp - s * gkq;
                        const double nq_ = s * gkp + c * gkq;
                        G2(k, p) = np_;
                        G2(p, k) = np_;
                        G2(k, q) = nq_;
                        G2(q, k) = nq_;
                    }
                    V2(k, p) = c * vkp - s * vkq;
                    V2(k, q) = s * vkp + c * vkq;
                }
                if (lane == 0) {
                    G2(p, p) = gpp - t * gpq;
                    G2(q, q) = gqq + t * gpq;
                    G2(p, q) = 0.0;
                    G2(q, p) = 0.0;
                }
                __syncthreads();
            }
        if (!rotated) break;
    }
}

// gram_to_h for that layout; every lane returns the same H and verdict.
__device__ bool gram_to_h_wave(double* sG, double* sV, int lane, const Norm& n1, const Norm& n2, Mat3& H, int mlesac) {
    if (lane == 0)
        for (int p = 0; p < 9; ++p)
            for (int q = 0; q < p; ++q) G2(p, q) = G2(q, p);
    __syncthreads();
    jacobi9_wave(sG, sV, lane);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (G2(k, k) < G2(kmin, kmin)) kmin = k;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = V2(k, kmin);
#undef G2
#undef V2
    Mat3 Hn, M;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M3(Hn, r, c) = h[3 * r + c] / h[8];
    for (int c = 0; c < 3; ++c) {
        const double m2 = M3(Hn, 2, c);
        M3(M, 2, c) = m2;
        M3(M, 1, c) = (M3(Hn, 1, c) - n2.ty * m2) / n2.s;
        M3(M, 0, c) = (M3(Hn, 0, c) - n2.tx * m2) / n2.s;
    }
    for (int r = 0; r < 3; ++r) {
        M3(H, r, 0) = M3(M, r, 0) * n1.s;
        M3(H, r, 1) = M3(M, r, 1) * n1.s;
        M3(H, r, 2) = (M3(M, r, 0) * n1.tx + M3(M, r, 1) * n1.ty) + M3(M, r, 2);
    }
    if (mlesac) {  // denormalizeTform: tform ./ tform(end) (estimateTransformationMLESAC.m:713-714)
        const double d = H.m[8];
        for (int e = 0; e < 9; ++e) H.m[e] = H.m[e] / d;
    }
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    return true;
}

// normalisation scale from the mean distance to the centroid: RANSAC 1/md (:592), MLESAC sqrt(2)/md guarded
// against md == 0 (normalizePointsHartleyZisserman, estimateTransformationMLESAC.m:653-657)
__device__ __forceinline__ double norm_scale(double md, int mlesac) {
    if (!mlesac) return 1.0 / md;
    return md > 0 ? sqrt(2.0) / md : 1.0;
}

// ------------------------------------------------------------------------------------------------
// fit kernel: one lane per (pair, draw)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ransac_fit_kernel(const double* __restrict__ pts1,
                                                         const double* __restrict__ pts2,
                                                         int64_t ldp,
                                                         const int64_t* __restrict__ pair_ptr,
                                                         const int* __restrict__ act, int n_act,
                                                         int c0, int nc,
                                                         const uint32_t* __restrict__ sample_idx,
                                                         int n_samples, double* __restrict__ Hs,
                                                         uint8_t* __restrict__ valid, int mlesac) {
    // work item = (active pair a, draw c0 + k): the host hands the draws over in growing chunks and stops a pair as
    // soon as its sequential loop has ended, so most of the n_samples draws of a pair are never fitted or scored.
    // Hs and valid keep the [pair][draw] layout (the finalize kernel picks the winner there; draws may be fitted ahead of
    // the chunk that scores them).
    extern __shared__ __attribute__((aligned(16))) double lds_fit[];
    constexpr int S = 64;  // one 9x9 problem per lane
    double* sG = lds_fit;
    double* sV = lds_fit + 81 * 64;
    const int lane = threadIdx.x;
    const int64_t wid = blockIdx.x * (int64_t)64 + lane;
    if (wid >= (int64_t)n_act * nc) return;  // no barriers below: every lane works on its own LDS column
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    double x1[4], y1[4], x2[4], y2[4];
    bool ok = m >= 4;
    for (int k = 0; k < 4; ++k) {
        const uint32_t id = sample_idx[gid * 4 + k];
        if (id < 1 || (int64_t)id > m) ok = false;
        const int64_t row = r0 + (ok ? (int64_t)id - 1 : 0);
        x1[k] = ok ? pts1[row] : 0.0;
        y1[k] = ok ? pts1[ldp + row] : 0.0;
        x2[k] = ok ? pts2[row] : 0.0;
        y2[k] = ok ? pts2[ldp + row] : 0.0;
    }
    Mat3 H;
    for (int e = 0; e < 9; ++e) H.m[e] = 0.0;
    if (ok) {
        // normalizePoints (:579-610), sums in index order
        Norm n1, n2;
        {
            double sx = 0, sy = 0;
            for (int k = 0; k < 4; ++k) {
                sx = sx + x1[k];
                sy = sy + y1[k];
            }
            const double cx = sx / 4.0, cy = sy / 4.0;
            double sd = 0;
            for (int k = 0; k < 4; ++k) {
                const double dx = x1[k] - cx, dy = y1[k] - cy;
                sd = sd + sqrt(dx * dx + dy * dy);
            }
            n1.s = norm_scale(sd / 4.0, mlesac);
            n1.tx = -n1.s * cx;
            n1.ty = -n1.s * cy;
            n1.cx = cx;
            n1.cy = cy;
        }
        {
            double sx = 0, sy = 0;
            for (int k = 0; k < 4; ++k) {
                sx = sx + x2[k];
                sy = sy + y2[k];
            }
            const double cx = sx / 4.0, cy = sy / 4.0;
            double sd = 0;
            for (int k = 0; k < 4; ++k) {
                const double dx = x2[k] - cx, dy = y2[k] - cy;
                sd = sd + sqrt(dx * dx + dy * dy);
            }
            n2.s = norm_scale(sd / 4.0, mlesac);
            n2.tx = -n2.s * cx;
            n2.ty = -n2.s * cy;
            n2.cx = cx;
            n2.cy = cy;
        }
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b) GE(a, b) = 0.0;
        // Gram sums in the reference's row order: RANSAC all "x" rows then all "y" rows (:209-212); MLESAC per
        // point its "v" row then its "u" row (estimateTransformationMLESAC.m:368-373; a a' is sign-blind)
        for (int step = 0; step < 8; ++step) {
            const int half = mlesac ? 1 - (step & 1) : step >> 2;
            const int k = mlesac ? step >> 1 : step & 3;
            const double x = norm_coord(n1, x1[k], n1.cx, n1.tx, mlesac), y = norm_coord(n1, y1[k], n1.cy, n1.ty, mlesac);
            const double u = norm_coord(n2, x2[k], n2.cx, n2.tx, mlesac), v = norm_coord(n2, y2[k], n2.cy, n2.ty, mlesac);
            double a[9];
            for (int e = 0; e < 9; ++e) a[e] = dlt_entry(e, half, x, y, u, v);
            for (int pp = 0; pp < 9; ++pp)
                for (int qq = pp; qq < 9; ++qq) GE(pp, qq) = GE(pp, qq) + a[pp] * a[qq];
        }
        ok = gram_to_h<64>(sG, sV, lane, n1, n2, H, mlesac) && (mlesac || check_model(H));
    }
    for (int e = 0; e < 9; ++e) Hs[gid * 9 + e] = H.m[e];
    valid[gid] = ok ? 1 : 0;  // [pair][draw], like Hs (the score kernels read the chunk-local copy valid_chunk_kernel makes)
}

// ------------------------------------------------------------------------------------------------
// wave-collective findInliers (:444-516)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double transfer_error(const Mat3& H, const Mat3& A, double x1, double y1,
                                                 double x2, double y2) {
    const double X = (M3(H, 0, 0) * x1 + M3(H, 0, 1) * y1) + M3(H, 0, 2);
    const double Y = (M3(H, 1, 0) * x1 + M3(H, 1, 1) * y1) + M3(H, 1, 2);
    const double W = (M3(H, 2, 0) * x1 + M3(H, 2, 1) * y1) + M3(H, 2, 2);
    const double tx = X / W, ty = Y / W;
    const double IX = (A.m[0] * x2 + A.m[3] * y2) + A.m[6];
    const double IY = (A.m[1] * x2 + A.m[4] * y2) + A.m[7];
    const double IW = (A.m[2] * x2 + A.m[5] * y2) + A.m[8];
    const double ix = IX / IW, iy = IY / IW;
    const double ex = x2 - tx, ey = y2 - ty, fx = x1 - ix, fy = y1 - iy;
    const double d1 = ex * ex + ey * ey;
    const double d2 = fx * fx + fy * fy;
    double e = sqrt(d1 + d2);
    if (!isfinite(e)) e = INFINITY;
    if (fabs(W) < kDblEps) e = INFINITY;
    return e;
}

// All 64 lanes of a wave call this with the same arguments.  mask (may be NULL) receives 0/1 per match.
__device__ int wave_find_inliers(const Mat3& H, const double* __restrict__ x1,
                                 const double* __restrict__ y1, const double* __restrict__ x2,
                                 const double* __restrict__ y2, int64_t m, double thr,
                                 uint8_t* __restrict__ mask, double* mean_err) {
    const int lane = threadIdx.x & 63;
    const Mat3 A = adjugate3(H);
    double pc = 0, pe = 0, px = 0, py = 0;
    for (int64_t i = lane; i < m; i += 64) {
        const double e = transfer_error(H, A, x1[i], y1[i], x2[i], y2[i]);
        const bool in = e < thr;
        if (mask) mask[i] = in ? 1 : 0;
        if (in) {
            pc += 1.0;
            pe = pe + e;
            px = px + x1[i];
            py = py + y1[i];
        }
    }
    const double cnt = wave_sum(pc);
    const double se = wave_sum(pe), sx = wave_sum(px), sy = wave_sum(py);
    const int n = (int)cnt;
    if (n >= 4) {  // isDegenerate on pts1(inliers) (:506-513, :537-574)
        const double mx = sx / cnt, my = sy / cnt;
        double pxx = 0, pxy = 0, pyy = 0;
        for (int64_t i = lane; i < m; i += 64) {
            const double e = transfer_error(H, A, x1[i], y1[i], x2[i], y2[i]);
            if (e < thr) {
                const double dx = x1[i] - mx, dy = y1[i] - my;
                pxx = pxx + dx * dx;
                pxy = pxy + dx * dy;
                pyy = pyy + dy * dy;
            }
        }
        const double sxx = wave_sum(pxx), sxy = wave_sum(pxy), syy = wave_sum(pyy);
        const double hs = 0.5 * (sxx + syy), hd = 0.5 * (sxx - syy);
        const double r = sqrt(hd * hd + sxy * sxy);
        const double l1 = hs + r;
        double l2 = hs - r;
        if (l2 < 0) l2 = 0;
        const double s1 = sqrt(l1), s2 = sqrt(l2);
        if (s2 / s1 < 1e-3) {
            if (mask)
                for (int64_t i = lane; i < m; i += 64) mask[i] = 0;
            *mean_err = NAN;
            return 0;
        }
    }
    *mean_err = n > 0 ? se / cnt : NAN;
    return n;
}

// MLESAC evaluateModel (estimateTransformationMLESAC.m:258-295, :534-562): one-way distance of H*x1 to x2,
// truncated at thr; returns the sum of the truncated distances (wave order), *n_inl = #(d < thr).
__device__ __forceinline__ double oneway_dist(const Mat3& H, double x1, double y1, double x2, double y2) {
    const double X = (M3(H, 0, 0) * x1 + M3(H, 0, 1) * y1) + M3(H, 0, 2);
    const double Y = (M3(H, 1, 0) * x1 + M3(H, 1, 1) * y1) + M3(H, 1, 2);
    const double W = (M3(H, 2, 0) * x1 + M3(H, 2, 1) * y1) + M3(H, 2, 2);
    const double dx = X / W - x2, dy = Y / W - y2;
    double d = sqrt(dx * dx + dy * dy);
    if (fabs(W) < kDblEps) d = INFINITY;
    return d;
}
__device__ double wave_mlesac_eval(const Mat3& H, const double* __restrict__ x1, const double* __restrict__ y1,
                                   const double* __restrict__ x2, const double* __restrict__ y2, int64_t m, double thr,
                                   uint8_t* __restrict__ mask, int* n_inl) {
    const int lane = threadIdx.x & 63;
    double ps = 0, pc = 0;
    for (int64_t i = lane; i < m; i += 64) {
        double d = oneway_dist(H, x1[i], y1[i], x2[i], y2[i]);
        if (d > thr) d = thr;  // NaN stays NaN
        const bool in = d < thr;
        if (mask) mask[i] = in ? 1 : 0;
        ps = ps + d;
        if (in) pc += 1.0;
    }
    *n_inl = (int)wave_sum(pc);
    return wave_sum(ps);
}

// valid [pair][draw] -> the chunk-local [active pair][draw of the chunk] order the score kernels and the host replay walk
__global__ void valid_chunk_kernel(const int* __restrict__ act, int n_act, int c0, int nc, int n_samples,
                                   const uint8_t* __restrict__ valid_abs, uint8_t* __restrict__ valid_loc) {
    const int64_t wid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (wid >= (int64_t)n_act * nc) return;
    valid_loc[wid] = valid_abs[(int64_t)act[wid / nc] * n_samples + c0 + (int)(wid % nc)];
}

__global__ __launch_bounds__(256) void mlesac_score_kernel(
    const double* __restrict__ pts1, const double* __restrict__ pts2, int64_t ldp,
    const int64_t* __restrict__ pair_ptr, const int* __restrict__ act, int n_act, int c0, int nc,
    int n_samples, const double* __restrict__ Hs,
    const uint8_t* __restrict__ valid, double thr, int32_t* __restrict__ n_inl, double* __restrict__ acc_dis) {
    const int64_t wid = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);  // (active pair, draw of the chunk)
    if (wid >= (int64_t)n_act * nc) return;
    const int lane = threadIdx.x & 63;
    if (!valid[wid]) {
        if (lane == 0) {
            n_inl[wid] = 0;
            acc_dis[wid] = NAN;
        }
        return;
    }
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = Hs[gid * 9 + e];
    int n;
    const double acc = wave_mlesac_eval(H, pts1 + r0, pts1 + ldp + r0, pts2 + r0, pts2 + ldp + r0, m, thr, nullptr, &n);
    if (lane == 0) {
        n_inl[wid] = n;
        acc_dis[wid] = acc;
    }
}

__device__ __forceinline__ int wave_find_inliers_any(int type, const Mat3& H, const double* x1, const double* y1,
                                                     const double* x2, const double* y2, int64_t m, double thr,
                                                     uint8_t* mask, double* mean_err);  // (defined with the other transformTypes)

// one wave per (pair, draw); 4 waves per block
__global__ __launch_bounds__(256) void ransac_score_kernel(int type,
    const double* __restrict__ pts1, const double* __restrict__ pts2, int64_t ldp,
    const int64_t* __restrict__ pair_ptr, const int* __restrict__ act, int n_act, int c0, int nc,
    int n_samples, const double* __restrict__ Hs,
    const uint8_t* __restrict__ valid, double thr, int32_t* __restrict__ n_inl,
    double* __restrict__ mean_err) {
    const int64_t wid = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);  // (active pair, draw of the chunk)
    if (wid >= (int64_t)n_act * nc) return;
    const int lane = threadIdx.x & 63;
    if (!valid[wid]) {
        if (lane == 0) {
            n_inl[wid] = 0;
            mean_err[wid] = NAN;
        }
        return;
    }
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = Hs[gid * 9 + e];
    double me;
    const int n = wave_find_inliers_any(type, H, pts1 + r0, pts1 + ldp + r0, pts2 + r0, pts2 + ldp + r0, m, thr,
                                        nullptr, &me);
    if (lane == 0) {
        n_inl[wid] = n;
        mean_err[wid] = me;
    }
}

// explicit-hypothesis scoring for aps_ransac_score: one wave per hypothesis, optional masks
__global__ __launch_bounds__(256) void ransac_score_explicit_kernel(int type,
    const double* __restrict__ p1, const double* __restrict__ p2, int64_t ldp, int64_t m,
    const double* __restrict__ Hs, int n_hyp, double thr, int32_t* __restrict__ n_inl,
    double* __restrict__ mean_err, uint8_t* __restrict__ mask) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_hyp) return;
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = Hs[(int64_t)t * 9 + e];
    double me;
    const int n = wave_find_inliers_any(type, H, p1, p1 + ldp, p2, p2 + ldp, m, thr,
                                        mask ? mask + (int64_t)t * m : nullptr, &me);
    if ((threadIdx.x & 63) == 0) {
        n_inl[t] = n;
        mean_err[t] = me;
    }
}

// ------------------------------------------------------------------------------------------------
// finalize: one wave per pair (:146-181)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ransac_finalize_kernel(
    const double* __restrict__ pts1, const double* __restrict__ pts2, int64_t ldp,
    const int64_t* __restrict__ pair_ptr, int n_samples, const double* __restrict__ Hs,
    const int32_t* __restrict__ best_it, double thr, double* __restrict__ models,
    uint8_t* __restrict__ mask, uint8_t* __restrict__ scratch_mask, int32_t* __restrict__ found,
    int32_t* __restrict__ n_final, int mlesac) {
    // one 9x9 problem per workgroup: two columns (work matrix / broadcast scratch), 2.6 KB - with the fit kernel's
    // 64-column layout (83 KB) only one workgroup fitted a CU and the pairs ran in two rounds
    __shared__ __attribute__((aligned(16))) double lds_fin[2 * 81 * 2];
    double* sG = lds_fin;
    double* sV = lds_fin + 81 * 2;
    const int p = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    const double *x1 = pts1 + r0, *y1 = pts1 + ldp + r0, *x2 = pts2 + r0, *y2 = pts2 + ldp + r0;
    uint8_t* out_mask = mask + r0;
    uint8_t* tmp_mask = scratch_mask + r0;
    const int bi = best_it[p];
    if (bi < 0) {
        for (int64_t i = lane; i < m; i += 64) out_mask[i] = 0;
        if (lane < 9) models[(int64_t)p * 9 + lane] = NAN;
        if (lane == 0) {
            found[p] = 0;
            n_final[p] = 0;
        }
        return;
    }
    Mat3 Hb;
    for (int e = 0; e < 9; ++e) Hb.m[e] = Hs[((int64_t)p * n_samples + bi) * 9 + e];
    double me;
    int nb;
    if (mlesac) {
        (void)wave_mlesac_eval(Hb, x1, y1, x2, y2, m, thr, out_mask, &nb);
        if (nb < 4) {  // isFound needs sum(bestInliers) >= sampleSize (estimateTransformationMLESAC.m:213-214)
            for (int64_t i = lane; i < m; i += 64) out_mask[i] = 0;
            if (lane < 9) models[(int64_t)p * 9 + lane] = NAN;
            if (lane == 0) {
                found[p] = 0;
                n_final[p] = 0;
            }
            return;
        }
    } else {
        nb = wave_find_inliers(Hb, x1, y1, x2, y2, m, thr, out_mask, &me);
    }
    __threadfence_block();
    // Refit on the inliers (:146-181 -> estimateHomography).  Its sums - centroids, mean distances, the Gram sums of the DLT
    // rows - run in the WAVE ORDER over the point index (round 4; oracle/ransac_oracle.c header): lane l accumulates the
    // inliers i = l, l + 64, ... in ascending order and the 64 partials meet in the xor butterfly, like the other inlier
    // sums.  Rounds 1-3 summed sequentially in index order, which made every lane of the pair's wave walk all of its
    // ~3 700 matches four times behind uniform mask tests: ~134 k serial vector instructions and 1.3 of the 3.4 ms of
    // the whole RANSAC stage for ~200 pairs, no faster for the 28 pairs of one rank of eight.
    Norm n1, n2;
    {
        double sx = 0, sy = 0, ux = 0, uy = 0;
        for (int64_t i = lane; i < m; i += 64)
            if (out_mask[i]) {
                sx = sx + x1[i];
                sy = sy + y1[i];
                ux = ux + x2[i];
                uy = uy + y2[i];
            }
        sx = wave_sum(sx);
        sy = wave_sum(sy);
        ux = wave_sum(ux);
        uy = wave_sum(uy);
        const double dn = (double)nb;
        const double cx = sx / dn, cy = sy / dn, dx2 = ux / dn, dy2 = uy / dn;
        double sd = 0, ud = 0;
        for (int64_t i = lane; i < m; i += 64)
            if (out_mask[i]) {
                const double ax = x1[i] - cx, ay = y1[i] - cy;
                sd = sd + sqrt(ax * ax + ay * ay);
                const double bx = x2[i] - dx2, by = y2[i] - dy2;
                ud = ud + sqrt(bx * bx + by * by);
            }
        sd = wave_sum(sd);
        ud = wave_sum(ud);
        n1.s = norm_scale(sd / dn, mlesac);
        n1.tx = -n1.s * cx;
        n1.ty = -n1.s * cy;
        n1.cx = cx;
        n1.cy = cy;
        n2.s = norm_scale(ud / dn, mlesac);
        n2.tx = -n2.s * dx2;
        n2.ty = -n2.s * dy2;
        n2.cx = dx2;
        n2.cy = dy2;
    }
    // Gram matrix: every lane keeps the 45 upper-triangular partial sums of ITS inliers (rows in the reference's order:
    // RANSAC all "x" rows, then all "y" rows; MLESAC per inlier its "v" row, then its "u" row), then the butterfly
    double g[45];
#pragma unroll
    for (int e = 0; e < 45; ++e) g[e] = 0;
    auto add_row = [&](int half, double x, double y, double u, double v) __attribute__((always_inline)) {
        double a[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) a[k] = dlt_entry(k, half, x, y, u, v);
        int e = 0;
#pragma unroll
        for (int pp = 0; pp < 9; ++pp)
#pragma unroll
            for (int qq = pp; qq < 9; ++qq, ++e) g[e] = g[e] + a[pp] * a[qq];
    };
    if (mlesac) {
        for (int64_t i = lane; i < m; i += 64)
            if (out_mask[i]) {
                const double x = (x1[i] - n1.cx) * n1.s, y = (y1[i] - n1.cy) * n1.s;
                const double u = (x2[i] - n2.cx) * n2.s, v = (y2[i] - n2.cy) * n2.s;
                add_row(1, x, y, u, v);
                add_row(0, x, y, u, v);
            }
    } else {
        for (int half = 0; half < 2; ++half)
            for (int64_t i = lane; i < m; i += 64)
                if (out_mask[i]) {
                    const double x = n1.s * x1[i] + n1.tx, y = n1.s * y1[i] + n1.ty;
                    const double u = n2.s * x2[i] + n2.tx, v = n2.s * y2[i] + n2.ty;
                    add_row(half, x, y, u, v);
                }
    }
    {
        int e = 0;
#pragma unroll
        for (int pp = 0; pp < 9; ++pp)
#pragma unroll
            for (int qq = pp; qq < 9; ++qq, ++e) {
                const double t = wave_sum(g[e]);
                if (lane == 0) sG[(pp * 9 + qq) * 2 + 0] = t;  // column 0 of the work matrix
            }
    }
    __syncthreads();
    Mat3 Hr;
    const bool ok = gram_to_h_wave(sG, sV, lane, n1, n2, Hr, mlesac) && (mlesac || check_model(Hr));
    bool use_refit = false;
    int nr = 0;
    if (mlesac) {  // :216-236: the refit is the answer; invalid or no inlier left -> not found
        if (ok) (void)wave_mlesac_eval(Hr, x1, y1, x2, y2, m, thr, tmp_mask, &nr);
        if (!ok || nr < 1) {
            __threadfence_block();
            for (int64_t i = lane; i < m; i += 64) out_mask[i] = 0;
            if (lane < 9) models[(int64_t)p * 9 + lane] = NAN;
            if (lane == 0) {
                found[p] = 0;
                n_final[p] = 0;
            }
            return;
        }
        use_refit = true;
    } else if (ok) {
        nr = wave_find_inliers(Hr, x1, y1, x2, y2, m, thr, tmp_mask, &me);
        use_refit = nr >= 4;
    }
    if (use_refit) {
        __threadfence_block();
        for (int64_t i = lane; i < m; i += 64) out_mask[i] = tmp_mask[i];
    }
    if (lane < 9) models[(int64_t)p * 9 + lane] = use_refit ? Hr.m[lane] : Hb.m[lane];
    if (lane == 0) {
        found[p] = 1;
        n_final[p] = use_refit ? nr : nb;
    }
}

// ------------------------------------------------------------------------------------------------
// The other transformTypes: 'affine' (:227-288), 'similarity' (:290-356), 'rigid' (:358-421), 'translation'
// (:423-452), findInliers' one-way error for them (:483-497).  svd / pinv / median are the closed forms of the
// oracle's second header (oracle/ransac_oracle.c), evaluated here in the same order.
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int tf_min_points(int type) {
    return type == APS_TFORM_AFFINE ? 3 : (type == APS_TFORM_SIMILARITY || type == APS_TFORM_RIGID) ? 2 : type == APS_TFORM_TRANSLATION ? 1 : 4;
}

// H = T2 \ Hn * T1 (left to right; Hn row-major), then the exact affine last row
__device__ __forceinline__ void denormalize_affine(const double* Hn, const Norm& n1, const Norm& n2, Mat3& H) {
    Mat3 M;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double m2 = Hn[6 + c];
        M3(M, 2, c) = m2;
        M3(M, 1, c) = (Hn[3 + c] - n2.ty * m2) / n2.s;
        M3(M, 0, c) = (Hn[c] - n2.tx * m2) / n2.s;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        M3(H, r, 0) = M3(M, r, 0) * n1.s;
        M3(H, r, 1) = M3(M, r, 1) * n1.s;
        M3(H, r, 2) = (M3(M, r, 0) * n1.tx + M3(M, r, 1) * n1.ty) + M3(M, r, 2);
    }
    M3(H, 2, 0) = 0.0;
    M3(H, 2, 1) = 0.0;
    M3(H, 2, 2) = 1.0;
}

// cyclic Jacobi on a symmetric 3x3 (row-major), the rotation rule of jacobi9
__device__ __forceinline__ void jacobi3(double* G, double* V) {
#pragma unroll
    for (int e = 0; e < 9; ++e) V[e] = (e % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double gpq = G[3 * p + q];
                const double gpp = G[3 * p + p], gqq = G[3 * q + q];
                if (!(fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq)))) {
                    rotated = true;
                    const double theta = (gqq - gpp) / (2.0 * gpq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0);
                    const double s = t * c;
                    const int k = 3 - p - q;
                    const double gkp = G[3 * k + p], gkq = G[3 * k + q];
                    const double np_ = c * gkp - s * gkq;
                    const double nq_ = s * gkp + c * gkq;
                    G[3 * k + p] = np_;
                    G[3 * p + k] = np_;
                    G[3 * k + q] = nq_;
                    G[3 * q + k] = nq_;
                    G[3 * p + p] = gpp - t * gpq;
                    G[3 * q + q] = gqq + t * gpq;
                    G[3 * p + q] = 0.0;
                    G[3 * q + p] = 0.0;
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk) {
                        const double vkp = V[3 * kk + p], vkq = V[3 * kk + q];
                        V[3 * kk + p] = c * vkp - s * vkq;
                        V[3 * kk + q] = s * vkp + c * vkq;
                    }
                }
            }
        if (!rotated) break;
    }
}

// MATLAB's median of two values
__device__ __forceinline__ double median2(double a, double b) {
    if (isnan(a) || isnan(b)) return NAN;
    if (b < a) {
        const double t = a;
        a = b;
        b = t;
    }
    const int sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
    if (sa != sb || isinf(a) || isinf(b)) return (a + b) / 2;
    return a + (b - a) / 2;
}

// what a median callback may need to form its per-point values
struct FitCtx {
    Norm n1, n2;
    double c1x, c1y, c2x, c2y;
};
enum { MED_DX = 0, MED_DY = 1, MED_RATIO = 2 };

// The estimators over an ordered point selection.  each(body) calls body(x1, y1, x2, y2) for every selected point in
// ascending order; med(which, ctx, &n_valid) returns the MATLAB median of the per-point quantity `which` over the selection
// (MED_RATIO: |pts2c| / |pts1c| over the points with |pts1c| > 1e-10; n_valid = how many there were).  Every caller of
// one fit evaluates the same expressions in the same order, so a wave may run it redundantly in all lanes.
template <class Each, class Med>
__device__ __forceinline__ bool fit_tform(int type, int n, Each&& each, Med&& med, Mat3& H) {
    FitCtx cx{};
    if (type == APS_TFORM_TRANSLATION) {
        int nv;
        const double tx = med(MED_DX, cx, &nv);
        const double ty = med(MED_DY, cx, &nv);
#pragma unroll
        for (int e = 0; e < 9; ++e) H.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
        M3(H, 0, 2) = tx;
        M3(H, 1, 2) = ty;
        return isfinite(tx) && isfinite(ty);
    }
    const double dn = (double)n;
    {  // normalizePoints (:579-610) of both sets
        double sx = 0, sy = 0, ux = 0, uy = 0;
        each([&](double a, double b, double c, double d) {
            sx = sx + a;
            sy = sy + b;
            ux = ux + c;
            uy = uy + d;
        });
        const double ax = sx / dn, ay = sy / dn, bx = ux / dn, by = uy / dn;
        double sd = 0, ud = 0;
        each([&](double a, double b, double c, double d) {
            const double dx = a - ax, dy = b - ay;
            sd = sd + sqrt(dx * dx + dy * dy);
            const double ex = c - bx, ey = d - by;
            ud = ud + sqrt(ex * ex + ey * ey);
        });
        cx.n1.s = 1.0 / (sd / dn);
        cx.n1.tx = -cx.n1.s * ax;
        cx.n1.ty = -cx.n1.s * ay;
        cx.n1.cx = ax;
        cx.n1.cy = ay;
        cx.n2.s = 1.0 / (ud / dn);
        cx.n2.tx = -cx.n2.s * bx;
        cx.n2.ty = -cx.n2.s * by;
        cx.n2.cx = bx;
        cx.n2.cy = by;
    }
    const Norm n1 = cx.n1, n2 = cx.n2;
    double Hn[9];
    if (type == APS_TFORM_AFFINE) {
        double gxx = 0, gxy = 0, gx = 0, gyy = 0, gy = 0, g1 = 0, bu0 = 0, bu1 = 0, bu2 = 0, bv0 = 0, bv1 = 0, bv2 = 0;
        each([&](double a, double b, double c, double d) {
            const double x = n1.s * a + n1.tx, y = n1.s * b + n1.ty;
            const double u = n2.s * c + n2.tx, v = n2.s * d + n2.ty;
            gxx = gxx + x * x;
            gxy = gxy + x * y;
            gx = gx + x;
            gyy = gyy + y * y;
            gy = gy + y;
            g1 = g1 + 1.0;
            bu0 = bu0 + x * u;
            bu1 = bu1 + y * u;
            bu2 = bu2 + u;
            bv0 = bv0 + x * v;
            bv1 = bv1 + y * v;
            bv2 = bv2 + v;
        });
        double G[9] = {gxx, gxy, gx, gxy, gyy, gy, gx, gy, g1}, V[9];
        jacobi3(G, V);
        double lmax = G[0];
        if (G[4] > lmax) lmax = G[4];
        if (G[8] > lmax) lmax = G[8];
        const double smax = sqrt(lmax > 0 ? lmax : 0.0);
        double cu[3], cv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double lam = G[3 * k + k];
            const double sg = sqrt(lam > 0 ? lam : 0.0);
            const bool keep = sg > 0 && !(sg < 1e-10 * smax) && lam > 64.0 * kDblEps * lmax;
            const double du = (V[k] * bu0 + V[3 + k] * bu1) + V[6 + k] * bu2;
            const double dv = (V[k] * bv0 + V[3 + k] * bv1) + V[6 + k] * bv2;
            cu[k] = keep ? du / lam : 0.0;
            cv[k] = keep ? dv / lam : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Hn[j] = (V[3 * j] * cu[0] + V[3 * j + 1] * cu[1]) + V[3 * j + 2] * cu[2];
            Hn[3 + j] = (V[3 * j] * cv[0] + V[3 * j + 1] * cv[1]) + V[3 * j + 2] * cv[2];
        }
    } else {  // similarity / rigid
        const bool rigid = type == APS_TFORM_RIGID;
        double sx = 0, sy = 0, su = 0, sv = 0;
        each([&](double a, double b, double c, double d) {
            sx = sx + (n1.s * a + n1.tx);
            sy = sy + (n1.s * b + n1.ty);
            su = su + (n2.s * c + n2.tx);
            sv = sv + (n2.s * d + n2.ty);
        });
        cx.c1x = sx / dn;
        cx.c1y = sy / dn;
        cx.c2x = su / dn;
        cx.c2y = sv / dn;
        const double c1x = cx.c1x, c1y = cx.c1y, c2x = cx.c2x, c2y = cx.c2y;
        double m11 = 0, m12 = 0, m21 = 0, m22 = 0, fax = 0, fay = 0, fbx = 0, fby = 0;
        each([&](double a, double b, double c, double d) {
            const double ax = (n1.s * a + n1.tx) - c1x, ay = (n1.s * b + n1.ty) - c1y;
            const double bx = (n2.s * c + n2.tx) - c2x, by = (n2.s * d + n2.ty) - c2y;
            m11 = m11 + bx * ax;
            m12 = m12 + bx * ay;
            m21 = m21 + by * ax;
            m22 = m22 + by * ay;
            fax = fax + ax * ax;
            fay = fay + ay * ay;
            fbx = fbx + bx * bx;
            fby = fby + by * by;
        });
        const double E = m11 + m22, A = m21 - m12;
        const double r = sqrt(E * E + A * A);
        double c = E / r, sn = A / r;
        double scale = 1.0;
        if (rigid) {
            const double F = m11 - m22, Gs = m21 + m12;
            const double Q = 0.5 * r, T = 0.5 * sqrt(F * F + Gs * Gs);
            const double sv1 = Q + T, sv2 = fabs(Q - T);
            const double cond = sv1 / (sv2 > kDblEps ? sv2 : kDblEps);
            if (cond > 1e6) {
                c = 1.0;
                sn = 0.0;
            }
        } else {
            const double sa = sqrt(fbx + fby) / sqrt(fax + fay);
            int nq = 0;
            const double mq = med(MED_RATIO, cx, &nq);
            scale = nq > 0 ? median2(sa, mq) : sa;
        }
        const double r11 = scale * c, r12 = scale * sn, r21 = scale * (-sn), r22 = scale * c;
        Hn[0] = r11;
        Hn[1] = r12;
        Hn[2] = c2x - (r11 * c1x + r12 * c1y);
        Hn[3] = r21;
        Hn[4] = r22;
        Hn[5] = c2y - (r21 * c1x + r22 * c1y);
    }
    Hn[6] = 0;
    Hn[7] = 0;
    Hn[8] = 1;
    denormalize_affine(Hn, n1, n2, H);
#pragma unroll
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    return true;
}

// the per-point quantity a median runs over (valid = it takes part)
__device__ __forceinline__ double med_value(int which, const FitCtx& cx, double a, double b, double c, double d, bool* valid) {
    *valid = true;
    if (which == MED_DX) return (c - a) + 0.0;  // (+ 0.0: no negative zero among the sort keys)
    if (which == MED_DY) return (d - b) + 0.0;
    const double ax = (cx.n1.s * a + cx.n1.tx) - cx.c1x, ay = (cx.n1.s * b + cx.n1.ty) - cx.c1y;
    const double bx = (cx.n2.s * c + cx.n2.tx) - cx.c2x, by = (cx.n2.s * d + cx.n2.ty) - cx.c2y;
    const double ra = sqrt(ax * ax + ay * ay), rb = sqrt(bx * bx + by * by);
    *valid = ra > 1e-10;
    return rb / ra;
}

// minimal-sample fit of one lane: K points in registers
template <int K>
__device__ __forceinline__ bool fit_sample(int type, const double* x1, const double* y1, const double* x2, const double* y2,
                                           Mat3& H) {
    auto each = [&](auto&& body) {
#pragma unroll
        for (int k = 0; k < K; ++k) body(x1[k], y1[k], x2[k], y2[k]);
    };
    auto med = [&](int which, const FitCtx& cx, int* nv) -> double {
        double v[K];
        int n = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            bool valid;
            const double q = med_value(which, cx, x1[k], y1[k], x2[k], y2[k], &valid);
            if (valid) {
                if (n == 0) v[0] = q;
                if (K > 1 && n == 1) v[K > 1 ? 1 : 0] = q;
                ++n;
            }
        }
        *nv = n;
        if (n == 0) return NAN;
        if (n == 1) return v[0];
        return median2(v[0], v[K > 1 ? 1 : 0]);  // K <= 2 wherever a median is taken (similarity 2, translation 1)
    };
    return fit_tform(type, K, each, med, H);
}

__global__ __launch_bounds__(64) void tform_fit_kernel(int type, const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                        int64_t ldp, const int64_t* __restrict__ pair_ptr,
                                                        const int* __restrict__ act, int n_act, int c0, int nc,
                                                        const uint32_t* __restrict__ sample_idx, int n_samples,
                                                        double* __restrict__ Hs, uint8_t* __restrict__ valid) {
    const int64_t wid = blockIdx.x * (int64_t)64 + threadIdx.x;
    if (wid >= (int64_t)n_act * nc) return;
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    const int K = tf_min_points(type);
    double x1[3], y1[3], x2[3], y2[3];
    bool ok = m >= K;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t id = k < K ? sample_idx[gid * 4 + k] : 1u;
        if (k < K && (id < 1 || (int64_t)id > m)) ok = false;
        const int64_t row = r0 + (ok ? (int64_t)id - 1 : 0);
        const bool ld = ok && k < K;
        x1[k] = ld ? pts1[row] : 0.0;
        y1[k] = ld ? pts1[ldp + row] : 0.0;
        x2[k] = ld ? pts2[row] : 0.0;
        y2[k] = ld ? pts2[ldp + row] : 0.0;
    }
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = 0.0;
    if (ok) {
        bool fin;
        if (K == 3)
            fin = fit_sample<3>(type, x1, y1, x2, y2, H);
        else if (K == 2)
            fin = fit_sample<2>(type, x1, y1, x2, y2, H);
        else
            fin = fit_sample<1>(type, x1, y1, x2, y2, H);
        ok = fin && check_model(H);
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) Hs[gid * 9 + e] = H.m[e];
    valid[gid] = ok ? 1 : 0;  // [pair][draw], like Hs (the score kernels read the chunk-local copy valid_chunk_kernel makes)
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off);
        if (o > v) v = o;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// findInliers (:444-516) for the affine family and translation: all 64 lanes of a wave, same arguments
__device__ int wave_find_inliers_t(int type, const Mat3& H, const double* __restrict__ x1, const double* __restrict__ y1,
                                   const double* __restrict__ x2, const double* __restrict__ y2, int64_t m, double thr,
                                   uint8_t* __restrict__ mask, double* mean_err) {
    const int lane = threadIdx.x & 63;
    double scale = 1.0;  // max(abs([pts1_homog(:); pts2_homog(:)])): the homogeneous ones take part (:487)
    if (type == APS_TFORM_TRANSLATION) {
        for (int64_t i = lane; i < m; i += 64) {
            if (fabs(x1[i]) > scale) scale = fabs(x1[i]);
            if (fabs(y1[i]) > scale) scale = fabs(y1[i]);
            if (fabs(x2[i]) > scale) scale = fabs(x2[i]);
            if (fabs(y2[i]) > scale) scale = fabs(y2[i]);
        }
        scale = wave_max(scale);
        thr = thr / scale;
    }
    auto error_of = [&](int64_t i) -> double {
        const double X = (M3(H, 0, 0) * x1[i] + M3(H, 0, 1) * y1[i]) + M3(H, 0, 2);
        const double Y = (M3(H, 1, 0) * x1[i] + M3(H, 1, 1) * y1[i]) + M3(H, 1, 2);
        const double W = (M3(H, 2, 0) * x1[i] + M3(H, 2, 1) * y1[i]) + M3(H, 2, 2);
        const double ex = x2[i] - X / W, ey = y2[i] - Y / W;
        double e = sqrt(ex * ex + ey * ey);
        if (type == APS_TFORM_TRANSLATION) e = e / scale;
        if (!isfinite(e)) e = INFINITY;
        if (fabs(W) < kDblEps) e = INFINITY;
        return e;
    };
    double pc = 0, pe = 0, px = 0, py = 0;
    for (int64_t i = lane; i < m; i += 64) {
        const double e = error_of(i);
        const bool in = e < thr;
        if (mask) mask[i] = in ? 1 : 0;
        if (in) {
            pc += 1.0;
            pe = pe + e;
            px = px + x1[i];
            py = py + y1[i];
        }
    }
    const double cnt = wave_sum(pc);
    const double se = wave_sum(pe), sx = wave_sum(px), sy = wave_sum(py);
    const int n = (int)cnt;
    if (type == APS_TFORM_AFFINE && n >= 3) {  // isDegenerate (:506-513, :537-574)
        const double mx = sx / cnt, my = sy / cnt;
        double pxx = 0, pxy = 0, pyy = 0;
        for (int64_t i = lane; i < m; i += 64)
            if (error_of(i) < thr) {
                const double dx = x1[i] - mx, dy = y1[i] - my;
                pxx = pxx + dx * dx;
                pxy = pxy + dx * dy;
                pyy = pyy + dy * dy;
            }
        const double sxx = wave_sum(pxx), sxy = wave_sum(pxy), syy = wave_sum(pyy);
        const double hs = 0.5 * (sxx + syy), hd = 0.5 * (sxx - syy);
        const double r = sqrt(hd * hd + sxy * sxy);
        const double l1 = hs + r;
        double l2 = hs - r;
        if (l2 < 0) l2 = 0;
        if (sqrt(l2) / sqrt(l1) < 1e-3) {
            if (mask)
                for (int64_t i = lane; i < m; i += 64) mask[i] = 0;
            *mean_err = NAN;
            return 0;
        }
    }
    *mean_err = n > 0 ? se / cnt : NAN;
    return n;
}

__device__ __forceinline__ int wave_find_inliers_any(int type, const Mat3& H, const double* x1, const double* y1,
                                                     const double* x2, const double* y2, int64_t m, double thr,
                                                     uint8_t* mask, double* mean_err) {
    return type == APS_TFORM_PROJECTIVE ? wave_find_inliers(H, x1, y1, x2, y2, m, thr, mask, mean_err)
                                        : wave_find_inliers_t(type, H, x1, y1, x2, y2, m, thr, mask, mean_err);
}

// k-th smallest (0-based) of the m sortable keys in `keys` (slots that do not take part hold ~0ull and k < #valid): an
// MSB-first radix select, one counting pass per bit, all 64 lanes.  Exact and independent of the order of the slots.
__device__ unsigned long long wave_kth_key(const unsigned long long* __restrict__ keys, int64_t m, int64_t k) {
    const int lane = threadIdx.x & 63;
    unsigned long long prefix = 0;
    for (int bit = 63; bit >= 0; --bit) {
        int c = 0;
        for (int64_t i = lane; i < m; i += 64) c += ((keys[i] >> bit) == (prefix >> bit)) ? 1 : 0;
        const int64_t zeros = wave_sum_i(c);  // keys that agree with the prefix above `bit` and have a 0 there
        if (k >= zeros) {
            k -= zeros;
            prefix |= 1ull << bit;
        }
    }
    return prefix;
}
__device__ __forceinline__ unsigned long long sort_key(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// finalize for these types: one wave per pair (:146-181)
__global__ __launch_bounds__(64) void tform_finalize_kernel(int type, const double* __restrict__ pts1,
                                                             const double* __restrict__ pts2, int64_t ldp,
                                                             const int64_t* __restrict__ pair_ptr, int n_samples,
                                                             const double* __restrict__ Hs, const int32_t* __restrict__ best_it,
                                                             double thr, double* __restrict__ models, uint8_t* __restrict__ mask,
                                                             uint8_t* __restrict__ scratch_mask,
                                                             unsigned long long* __restrict__ scratch_keys,
                                                             int32_t* __restrict__ found, int32_t* __restrict__ n_final) {
    const int p = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    const int min_pts = tf_min_points(type);
    const double *x1 = pts1 + r0, *y1 = pts1 + ldp + r0, *x2 = pts2 + r0, *y2 = pts2 + ldp + r0;
    uint8_t* out_mask = mask + r0;
    uint8_t* tmp_mask = scratch_mask + r0;
    unsigned long long* keys = scratch_keys + r0;
    const int bi = best_it[p];
    if (bi < 0) {
        for (int64_t i = lane; i < m; i += 64) out_mask[i] = 0;
        if (lane < 9) models[(int64_t)p * 9 + lane] = NAN;
        if (lane == 0) {
            found[p] = 0;
            n_final[p] = 0;
        }
        return;
    }
    Mat3 Hb;
    for (int e = 0; e < 9; ++e) Hb.m[e] = Hs[((int64_t)p * n_samples + bi) * 9 + e];
    double me;
    const int nb = wave_find_inliers_t(type, Hb, x1, y1, x2, y2, m, thr, out_mask, &me);
    __threadfence_block();
    __syncthreads();
    // the refit walks the inliers in ascending order, staged through LDS 64 at a time; every lane evaluates the same sums
    __shared__ double s_pt[4][64];
    __shared__ uint8_t s_in[64];
    auto each = [&](auto&& body) {
        for (int64_t base = 0; base < m; base += 64) {
            const int64_t i = base + lane;
            const bool have = i < m;
            s_pt[0][lane] = have ? x1[i] : 0.0;
            s_pt[1][lane] = have ? y1[i] : 0.0;
            s_pt[2][lane] = have ? x2[i] : 0.0;
            s_pt[3][lane] = have ? y2[i] : 0.0;
            s_in[lane] = have ? out_mask[i] : (uint8_t)0;
            __syncthreads();
            for (int e = 0; e < 64; ++e)
                if (s_in[e]) body(s_pt[0][e], s_pt[1][e], s_pt[2][e], s_pt[3][e]);
            __syncthreads();
        }
    };
    auto med = [&](int which, const FitCtx& cx, int* nv) -> double {
        int c_valid = 0, c_nan = 0;
        for (int64_t i = lane; i < m; i += 64) {
            unsigned long long key = ~0ull;
            if (out_mask[i]) {
                bool valid;
                const double q = med_value(which, cx, x1[i], y1[i], x2[i], y2[i], &valid);
                if (valid) {
                    ++c_valid;
                    if (isnan(q)) ++c_nan;
                    key = sort_key(q);
                }
            }
            keys[i] = key;
        }
        const int n = wave_sum_i(c_valid), n_nan = wave_sum_i(c_nan);
        __threadfence_block();
        __syncthreads();
        *nv = n;
        if (n == 0 || n_nan > 0) return NAN;
        if (n & 1) return key_value(wave_kth_key(keys, m, (n - 1) / 2));
        const double a = key_value(wave_kth_key(keys, m, n / 2 - 1)), b = key_value(wave_kth_key(keys, m, n / 2));
        return median2(a, b);
    };
    Mat3 Hr;
    const bool ok = fit_tform(type, nb, each, med, Hr) && check_model(Hr);
    bool use_refit = false;
    int nr = 0;
    if (ok) {
        nr = wave_find_inliers_t(type, Hr, x1, y1, x2, y2, m, thr, tmp_mask, &me);
        use_refit = nr >= min_pts;
    }
    if (use_refit) {
        __threadfence_block();
        __syncthreads();
        for (int64_t i = lane; i < m; i += 64) out_mask[i] = tmp_mask[i];
    }
    if (lane < 9) models[(int64_t)p * 9 + lane] = use_refit ? Hr.m[lane] : Hb.m[lane];
    if (lane == 0) {
        found[p] = 1;
        n_final[p] = use_refit ? nr : nb;
    }
}

// ------------------------------------------------------------------------------------------------
// MLESAC for the other transformationTypes (estimateTransformationMLESAC.m): estimateAffine (:389-424, null vector of
// the 2n x 7 system), estimateSimilarity (:426-458, 2n x 5), estimateRigid (:460-490, Kabsch on the raw points, closed
// form), estimateTranslation (:492-510, the mean displacement), evaluateTranslation2d (:578-598).
// ------------------------------------------------------------------------------------------------
// entry k of a constraint row: half 1 = the point's "v" row (odd rows of :404-407 / :441-444), half 0 its "u" row
__device__ __forceinline__ double mlesac_entry(int type, int k, int half, double x, double y, double u, double v) {
    if (type == APS_TFORM_AFFINE) {
        if (half) return k == 3 ? -x : k == 4 ? -y : k == 5 ? -1.0 : k == 6 ? v : 0.0;
        return k == 0 ? x : k == 1 ? y : k == 2 ? 1.0 : k == 6 ? -u : 0.0;
    }
    if (half) return k == 0 ? -y : k == 1 ? x : k == 3 ? -1.0 : k == 4 ? v : 0.0;
    return k == 0 ? x : k == 1 ? y : k == 2 ? 1.0 : k == 4 ? -u : 0.0;
}

// T from the null vector h of the affine (N = 7) / similarity (N = 5) system, denormalised: (N2 \ T) * N1, ./ T(end)
__device__ __forceinline__ bool mlesac_h_to_model(int type, const double* h, const Norm& n1, const Norm& n2, Mat3& H) {
    double Tn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    if (type == APS_TFORM_AFFINE) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Tn[k] = h[k] / h[6];
    } else {
        Tn[0] = h[0] / h[4];
        Tn[1] = h[1] / h[4];
        Tn[2] = h[2] / h[4];
        Tn[3] = -h[1] / h[4];
        Tn[4] = h[0] / h[4];
        Tn[5] = h[3] / h[4];
    }
    Mat3 M;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double m2 = Tn[6 + c];
        M3(M, 2, c) = m2;
        M3(M, 1, c) = (Tn[3 + c] - n2.ty * m2) / n2.s;
        M3(M, 0, c) = (Tn[c] - n2.tx * m2) / n2.s;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        M3(H, r, 0) = M3(M, r, 0) * n1.s;
        M3(H, r, 1) = M3(M, r, 1) * n1.s;
        M3(H, r, 2) = (M3(M, r, 0) * n1.tx + M3(M, r, 1) * n1.ty) + M3(M, r, 2);
    }
    const double d = H.m[8];
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = H.m[e] / d;
#pragma unroll
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    return true;
}

// rigid / translation over an ordered selection (`each` as in fit_tform); every caller evaluates the same expressions
template <class Each>
__device__ __forceinline__ bool mlesac_fit_closed(int type, int n, Each&& each, Mat3& H) {
    const double dn = (double)n;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
    if (type == APS_TFORM_TRANSLATION) {
        double sx = 0, sy = 0;
        each([&](double a, double b, double c, double d) {
            sx = sx + (c - a);
            sy = sy + (d - b);
        });
        M3(H, 0, 2) = sx / dn;
        M3(H, 1, 2) = sy / dn;
        return isfinite(M3(H, 0, 2)) && isfinite(M3(H, 1, 2));
    }
    double sx = 0, sy = 0, su = 0, sv = 0;
    each([&](double a, double b, double c, double d) {
        sx = sx + a;
        sy = sy + b;
        su = su + c;
        sv = sv + d;
    });
    const double c1x = sx / dn, c1y = sy / dn, c2x = su / dn, c2y = sv / dn;
    double c11 = 0, c12 = 0, c21 = 0, c22 = 0;
    each([&](double a, double b, double c, double d) {
        const double ax = a - c1x, ay = b - c1y, bx = c - c2x, by = d - c2y;
        c11 = c11 + ax * bx;
        c12 = c12 + ax * by;
        c21 = c21 + ay * bx;
        c22 = c22 + ay * by;
    });
    const double E = c11 + c22, A = c12 - c21;
    const double r = sqrt(E * E + A * A);
    const double c = E / r, sn = A / r;  // R = [c -sn; sn c]
    M3(H, 0, 0) = c;
    M3(H, 0, 1) = -sn;
    M3(H, 1, 0) = sn;
    M3(H, 1, 1) = c;
    M3(H, 0, 2) = c2x - (c * c1x + (-sn) * c1y);
    M3(H, 1, 2) = c2y - (sn * c1x + c * c1y);
#pragma unroll
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H.m[e])) return false;
    return true;
}

// Hartley-Zisserman normalisation of an ordered selection (sums in order)
template <class Each>
__device__ __forceinline__ void mlesac_normalize(int n, Each&& each, Norm& n1, Norm& n2) {
    const double dn = (double)n;
    double sx = 0, sy = 0, ux = 0, uy = 0;
    each([&](double a, double b, double c, double d) {
        sx = sx + a;
        sy = sy + b;
        ux = ux + c;
        uy = uy + d;
    });
    n1.cx = sx / dn;
    n1.cy = sy / dn;
    n2.cx = ux / dn;
    n2.cy = uy / dn;
    double sd = 0, ud = 0;
    each([&](double a, double b, double c, double d) {
        const double dx = a - n1.cx, dy = b - n1.cy;
        sd = sd + sqrt(dx * dx + dy * dy);
        const double ex = c - n2.cx, ey = d - n2.cy;
        ud = ud + sqrt(ex * ex + ey * ey);
    });
    n1.s = norm_scale(sd / dn, 1);
    n1.tx = -n1.s * n1.cx;
    n1.ty = -n1.s * n1.cy;
    n2.s = norm_scale(ud / dn, 1);
    n2.tx = -n2.s * n2.cx;
    n2.ty = -n2.s * n2.cy;
}

// one lane per (pair, draw): K = 3 / 2 / 2 / 1 sample points; the N x N Gram problem of a lane in its LDS column
template <int N>
__device__ __forceinline__ bool mlesac_fit_sample_null(int type, int K, const double* x1, const double* y1, const double* x2,
                                                       const double* y2, double* sG, double* sV, int lane, Mat3& H) {
    constexpr int S = 64;
    auto each = [&](auto&& body) {
        for (int k = 0; k < K; ++k) body(x1[k], y1[k], x2[k], y2[k]);
    };
    Norm n1, n2;
    mlesac_normalize(K, each, n1, n2);
    for (int a = 0; a < N; ++a)
        for (int b = a; b < N; ++b) GE(a, b) = 0.0;
    for (int k = 0; k < K; ++k)
        for (int half = 1; half >= 0; --half) {
            const double x = (x1[k] - n1.cx) * n1.s, y = (y1[k] - n1.cy) * n1.s;
            const double u = (x2[k] - n2.cx) * n2.s, v = (y2[k] - n2.cy) * n2.s;
            double a[N];
#pragma unroll
            for (int e = 0; e < N; ++e) a[e] = mlesac_entry(type, e, half, x, y, u, v);
#pragma unroll
            for (int pp = 0; pp < N; ++pp)
#pragma unroll
                for (int qq = pp; qq < N; ++qq) GE(pp, qq) = GE(pp, qq) + a[pp] * a[qq];
        }
    for (int p = 0; p < N; ++p)
        for (int q = 0; q < p; ++q) GE(p, q) = GE(q, p);
    jacobi9<S, N>(sG, sV, lane);
    int kmin = 0;
    for (int k = 1; k < N; ++k)
        if (GE(k, k) < GE(kmin, kmin)) kmin = k;
    double h[N];
    for (int k = 0; k < N; ++k) h[k] = VE(k, kmin);
    return mlesac_h_to_model(type, h, n1, n2, H);
}

__global__ __launch_bounds__(64) void mlesac_tform_fit_kernel(int type, const double* __restrict__ pts1,
                                                               const double* __restrict__ pts2, int64_t ldp,
                                                               const int64_t* __restrict__ pair_ptr,
                                                               const int* __restrict__ act, int n_act, int c0, int nc,
                                                               const uint32_t* __restrict__ sample_idx, int n_samples,
                                                               double* __restrict__ Hs, uint8_t* __restrict__ valid) {
    extern __shared__ __attribute__((aligned(16))) double lds_fit[];
    double* sG = lds_fit;
    double* sV = lds_fit + 81 * 64;
    const int lane = threadIdx.x;
    const int64_t wid = blockIdx.x * (int64_t)64 + lane;
    if (wid >= (int64_t)n_act * nc) return;  // no barriers below
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    const int K = tf_min_points(type);
    double x1[3], y1[3], x2[3], y2[3];
    bool ok = m >= K;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t id = k < K ? sample_idx[gid * 4 + k] : 1u;
        if (k < K && (id < 1 || (int64_t)id > m)) ok = false;
        const int64_t row = r0 + (ok ? (int64_t)id - 1 : 0);
        const bool ld = ok && k < K;
        x1[k] = ld ? pts1[row] : 0.0;
        y1[k] = ld ? pts1[ldp + row] : 0.0;
        x2[k] = ld ? pts2[row] : 0.0;
        y2[k] = ld ? pts2[ldp + row] : 0.0;
    }
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = 0.0;
    if (ok) {
        if (type == APS_TFORM_AFFINE)
            ok = mlesac_fit_sample_null<7>(type, K, x1, y1, x2, y2, sG, sV, lane, H);
        else if (type == APS_TFORM_SIMILARITY)
            ok = mlesac_fit_sample_null<5>(type, K, x1, y1, x2, y2, sG, sV, lane, H);
        else
            ok = mlesac_fit_closed(type, K, [&](auto&& body) {
                for (int k = 0; k < K; ++k) body(x1[k], y1[k], x2[k], y2[k]);
            }, H);
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) Hs[gid * 9 + e] = H.m[e];
    valid[gid] = ok ? 1 : 0;  // [pair][draw], like Hs (the score kernels read the chunk-local copy valid_chunk_kernel makes)
}

// evaluateModel over evaluateTransform2d, or evaluateTranslation2d for 'translation' (no division, no |w| test)
__device__ double wave_mlesac_eval_any(int type, const Mat3& H, const double* __restrict__ x1, const double* __restrict__ y1,
                                       const double* __restrict__ x2, const double* __restrict__ y2, int64_t m, double thr,
                                       uint8_t* __restrict__ mask, int* n_inl) {
    if (type != APS_TFORM_TRANSLATION) return wave_mlesac_eval(H, x1, y1, x2, y2, m, thr, mask, n_inl);
    const int lane = threadIdx.x & 63;
    double ps = 0, pc = 0;
    for (int64_t i = lane; i < m; i += 64) {
        const double dx = (x1[i] + M3(H, 0, 2)) - x2[i], dy = (y1[i] + M3(H, 1, 2)) - y2[i];
        double d = sqrt(dx * dx + dy * dy);
        if (d > thr) d = thr;
        const bool in = d < thr;
        if (mask) mask[i] = in ? 1 : 0;
        ps = ps + d;
        if (in) pc += 1.0;
    }
    *n_inl = (int)wave_sum(pc);
    return wave_sum(ps);
}

__global__ __launch_bounds__(256) void mlesac_tform_score_kernel(
    int type, const double* __restrict__ pts1, const double* __restrict__ pts2, int64_t ldp,
    const int64_t* __restrict__ pair_ptr, const int* __restrict__ act, int n_act, int c0, int nc, int n_samples,
    const double* __restrict__ Hs, const uint8_t* __restrict__ valid, double thr, int32_t* __restrict__ n_inl,
    double* __restrict__ acc_dis) {
    const int64_t wid = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (wid >= (int64_t)n_act * nc) return;
    const int lane = threadIdx.x & 63;
    if (!valid[wid]) {
        if (lane == 0) {
            n_inl[wid] = 0;
            acc_dis[wid] = NAN;
        }
        return;
    }
    const int p = act[wid / nc];
    const int64_t gid = (int64_t)p * n_samples + c0 + (int)(wid % nc);
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    Mat3 H;
#pragma unroll
    for (int e = 0; e < 9; ++e) H.m[e] = Hs[gid * 9 + e];
    int n;
    const double acc = wave_mlesac_eval_any(type, H, pts1 + r0, pts1 + ldp + r0, pts2 + r0, pts2 + ldp + r0, m, thr, nullptr, &n);
    if (lane == 0) {
        n_inl[wid] = n;
        acc_dis[wid] = acc;
    }
}

// finalize: one wave per pair (:213-241): the best draw's inliers, refit on them, re-evaluate; the refit is the answer
__global__ __launch_bounds__(64) void mlesac_tform_finalize_kernel(int type, const double* __restrict__ pts1,
                                                                    const double* __restrict__ pts2, int64_t ldp,
                                                                    const int64_t* __restrict__ pair_ptr, int n_samples,
                                                                    const double* __restrict__ Hs,
                                                                    const int32_t* __restrict__ best_it, double thr,
                                                                    double* __restrict__ models, uint8_t* __restrict__ mask,
                                                                    uint8_t* __restrict__ scratch_mask,
                                                                    int32_t* __restrict__ found, int32_t* __restrict__ n_final) {
    __shared__ __attribute__((aligned(16))) double lds_fin[2 * 81 * 2];
    double* sG = lds_fin;
    double* sV = lds_fin + 81 * 2;
    const int p = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t r0 = pair_ptr[p];
    const int64_t m = pair_ptr[p + 1] - r0;
    const int min_pts = tf_min_points(type);
    const double *x1 = pts1 + r0, *y1 = pts1 + ldp + r0, *x2 = pts2 + r0, *y2 = pts2 + ldp + r0;
    uint8_t* out_mask = mask + r0;
    uint8_t* tmp_mask = scratch_mask + r0;
    auto not_found = [&]() {
        __threadfence_block();
        __syncthreads();
        for (int64_t i = lane; i < m; i += 64) out_mask[i] = 0;
        if (lane < 9) models[(int64_t)p * 9 + lane] = NAN;
        if (lane == 0) {
            found[p] = 0;
            n_final[p] = 0;
        }
    };
    const int bi = best_it[p];
    if (bi < 0) {
        not_found();
        return;
    }
    Mat3 Hb;
    for (int e = 0; e < 9; ++e) Hb.m[e] = Hs[((int64_t)p * n_samples + bi) * 9 + e];
    int nb;
    (void)wave_mlesac_eval_any(type, Hb, x1, y1, x2, y2, m, thr, out_mask, &nb);
    if (nb < min_pts) {
        not_found();
        return;
    }
    __threadfence_block();
    __syncthreads();
    __shared__ double s_pt[4][64];
    __shared__ uint8_t s_in[64];
    auto each = [&](auto&& body) {
        for (int64_t base = 0; base < m; base += 64) {
            const int64_t i = base + lane;
            const bool have = i < m;
            s_pt[0][lane] = have ? x1[i] : 0.0;
            s_pt[1][lane] = have ? y1[i] : 0.0;
            s_pt[2][lane] = have ? x2[i] : 0.0;
            s_pt[3][lane] = have ? y2[i] : 0.0;
            s_in[lane] = have ? out_mask[i] : (uint8_t)0;
            __syncthreads();
            for (int e = 0; e < 64; ++e)
                if (s_in[e]) body(s_pt[0][e], s_pt[1][e], s_pt[2][e], s_pt[3][e]);
            __syncthreads();
        }
    };
    Mat3 Hr;
    bool ok;
    if (type == APS_TFORM_RIGID || type == APS_TFORM_TRANSLATION) {
        ok = mlesac_fit_closed(type, nb, each, Hr);
    } else {
        const int N = type == APS_TFORM_AFFINE ? 7 : 5;
        Norm n1, n2;
        mlesac_normalize(nb, each, n1, n2);
        // lane e owns the upper-triangular Gram entry (pp, qq) of the N x N system; rows per inlier: "v" then "u"
        int pp = 0, qq = 0;
        {
            int e = lane < N * (N + 1) / 2 ? lane : 0, row = 0;
            while (e >= N - row) {
                e -= N - row;
                ++row;
            }
            pp = row;
            qq = row + e;
        }
        double g = 0;
        each([&](double a, double b, double c, double d) {
            const double x = (a - n1.cx) * n1.s, y = (b - n1.cy) * n1.s;
            const double u = (c - n2.cx) * n2.s, v = (d - n2.cy) * n2.s;
            g = g + mlesac_entry(type, pp, 1, x, y, u, v) * mlesac_entry(type, qq, 1, x, y, u, v);
            g = g + mlesac_entry(type, pp, 0, x, y, u, v) * mlesac_entry(type, qq, 0, x, y, u, v);
        });
#define G2(p, q) sG[((p) * 9 + (q)) * 2]
#define V2(p, q) sV[((p) * 9 + (q)) * 2]
        if (lane < N * (N + 1) / 2) G2(pp, qq) = g;
        __syncthreads();
        if (lane == 0)
            for (int a = 0; a < N; ++a)
                for (int b = 0; b < a; ++b) G2(a, b) = G2(b, a);
        __syncthreads();
        if (N == 7)
            jacobi9_wave<7>(sG, sV, lane);
        else
            jacobi9_wave<5>(sG, sV, lane);
        int kmin = 0;
        for (int k = 1; k < N; ++k)
            if (G2(k, k) < G2(kmin, kmin)) kmin = k;
        double h[7];
        for (int k = 0; k < 7; ++k) h[k] = k < N ? V2(k, kmin) : 0.0;
#undef G2
#undef V2
        ok = mlesac_h_to_model(type, h, n1, n2, Hr);
    }
    int nr = 0;
    if (ok) (void)wave_mlesac_eval_any(type, Hr, x1, y1, x2, y2, m, thr, tmp_mask, &nr);
    if (!ok || nr < 1) {
        not_found();
        return;
    }
    __threadfence_block();
    __syncthreads();
    for (int64_t i = lane; i < m; i += 64) out_mask[i] = tmp_mask[i];
    if (lane < 9) models[(int64_t)p * 9 + lane] = Hr.m[lane];
    if (lane == 0) {
        found[p] = 1;
        n_final[p] = nr;
    }
}

// ------------------------------------------------------------------------------------------------
// seeded 4-subsets (stand-in for randperm(numPoints, 4), :96)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double mix_uniform(unsigned long long seed, unsigned long long key, unsigned long long ctr) {
    unsigned long long x = seed * 0x9E3779B97F4A7C15ull + key * 0xD1B54A32D192ED03ull + ctr * 0x8CB92BA72F3D8DD7ull +
                           0x2545F4914F6CDD1Dull;
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

__global__ void draw_samples_kernel(const int64_t* __restrict__ counts, const unsigned long long* __restrict__ keys,
                                    int n_pairs, int n_samples, unsigned long long seed,
                                    uint32_t* __restrict__ out) {
    const int64_t gid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gid >= (int64_t)n_pairs * n_samples) return;
    const int p = (int)(gid / n_samples), it = (int)(gid % n_samples);
    const long long n = counts[p];
    uint32_t* o = out + gid * 4;
    const unsigned long long key = keys ? keys[p] : (unsigned long long)p;
    // partial Fisher-Yates without a table: the k-th draw picks among the n-k values not chosen yet, so it skips over
    // the earlier picks in ascending order.  A pair with fewer than 4 matches gets n distinct entries, then ones.
    long long srt[4];  // the picks so far, ascending
    for (int k = 0; k < 4; ++k) {
        if (k >= n) {
            o[k] = 1u;
            continue;
        }
        long long v = (long long)(mix_uniform(seed, key, 4ull * it + k) * (double)(n - k));
        v = v < n - k - 1 ? v : n - k - 1;
        for (int j = 0; j < k; ++j) v = v + (v >= srt[j]);
        o[k] = (uint32_t)(v + 1);
        int j = k;
        while (j > 0 && srt[j - 1] > v) {
            srt[j] = srt[j - 1];
            --j;
        }
        srt[j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// imageMatching.m:121-135 on the device: one thread per match of the work list
__global__ void gather_match_points_kernel(const unsigned long long* __restrict__ tab, int n_img, int n_work,
                                           const int32_t* __restrict__ idx_a, const int32_t* __restrict__ idx_b,
                                           int64_t total, double* __restrict__ pa, double* __restrict__ pb, int64_t ldp) {
    const int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (m >= total) return;
    const unsigned long long* kp = tab;
    const unsigned long long* cnt = tab + n_img;
    const unsigned long long* lst = cnt + n_img;
    const unsigned long long* wp = lst + n_work;
    const unsigned long long* im = wp + n_work + 1;
    int lo = 0, hi = n_work - 1;  // the work pair whose slice holds match m
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)wp[mid] <= m)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int64_t pos = (int64_t)lst[lo] + (m - (int64_t)wp[lo]);
    const int ia = (int)(im[lo] & 0xffffffffull), ib = (int)(im[lo] >> 32);
    const int64_t ra = (int64_t)idx_a[pos] - 1, rb = (int64_t)idx_b[pos] - 1;
    const double* ka = reinterpret_cast<const double*>((uintptr_t)kp[ia]);
    const double* kb = reinterpret_cast<const double*>((uintptr_t)kp[ib]);
    const bool oka = ra >= 0 && ra < (int64_t)cnt[ia], okb = rb >= 0 && rb < (int64_t)cnt[ib];
    pa[m] = oka ? ka[2 * ra] : NAN;
    pa[ldp + m] = oka ? ka[2 * ra + 1] : NAN;
    pb[m] = okb ? kb[2 * rb] : NAN;
    pb[ldp + m] = okb ? kb[2 * rb + 1] : NAN;
}

static void check_opts(const aps_ransac_opts& o) {
    APS_REQUIRE(o.tform_type >= APS_TFORM_PROJECTIVE && o.tform_type <= APS_TFORM_TRANSLATION, APS_E_TYPE,
                "unknown transformationType %d", o.tform_type);
    APS_REQUIRE(o.method == APS_ROBUST_RANSAC || o.method == APS_ROBUST_MLESAC, APS_E_ARG, "unknown robust estimator %d", o.method);
    APS_REQUIRE(o.max_iter > 0, APS_E_ARG, "maxIter must be positive");
    APS_REQUIRE(o.max_distance > 0, APS_E_ARG, "maxDistance must be positive");
    APS_REQUIRE(o.confidence > 0 && o.confidence < 100, APS_E_ARG, "inliersConfidence must be in (0,100)");
}

// The sequential part of the loop (:94-143) over pre-scored draws, resumable: the draws of a pair arrive in chunks
// and the loop state is carried from one chunk to the next, so the outcome is that of one pass over all of them.
static thread_local int g_draws_exhausted = 0;  // pairs of this thread's last call whose draws ran out (see aps.h)

struct Replay {
    int trial = 1, skip = 0, it = 0, best_it = -1, best_n = 0, limit = 0;
    double best = INFINITY;
    bool done = false;
};

// valid/n_inl/mean_err hold draws [st.it, upto) of this pair
static void replay_loop(Replay& st, const uint8_t* valid, const int32_t* n_inl, const double* mean_err, int upto,
                        int64_t m, const aps_ransac_opts& o) {
    const int min_pts = tf_min_points(o.tform_type);
    const int max_skip = o.max_iter * 10;
    const int base = st.it;
    while (st.trial <= st.limit && st.skip < max_skip && st.it < upto) {
        const int cur = st.it++ - base;
        if (!valid[cur]) {
            ++st.skip;
            continue;
        }
        const int n = n_inl[cur];
        if (n >= min_pts) {
            const double me = mean_err[cur];
            if (n > st.best_n || (n == st.best_n && me < st.best)) {
                st.best_n = n;
                st.best = me;
                st.best_it = base + cur;
                const double ratio = (double)n / (double)m;
                if (ratio > 0) {
                    const double need = std::ceil(std::log(1 - o.confidence / 100) /
                                                  std::log(1 - std::pow(ratio, min_pts)));
                    if (need < (double)st.limit) st.limit = (int)need;
                }
            }
        }
        ++st.trial;
    }
    st.done = !(st.trial <= st.limit && st.skip < max_skip);
}

// vision.internal.ransac.computeLoopNumber restated (toolbox-internal, unpinned)
static int mlesac_loop_number(int sample_size, double confidence, int64_t num_pts, int inlier_num) {
    const double pr = std::pow((double)inlier_num / (double)num_pts, (double)sample_size);
    if (pr < 2.220446049250313e-16) return 2147483647;
    const double n = std::ceil(std::log10(1.0 - 0.01 * confidence) / std::log10(1.0 - pr));
    if (!(n < 2147483647.0)) return 2147483647;
    return n < 0 ? 0 : (int)n;
}

// The sequential part of mlesac() (estimateTransformationMLESAC.m:157-211) over pre-scored draws (resumable, as above)
static void replay_mlesac(Replay& st, const uint8_t* valid, const int32_t* n_inl, const double* acc_dis, int upto,
                          int64_t m, const aps_ransac_opts& o) {
    const int max_skip = 10000;  // setDefaultParams: maxIterations (1000) * 10; the caller cannot override it (:72)
    const int base = st.it;
    while (st.trial <= st.limit && st.skip < max_skip && st.it < upto) {
        const int cur = st.it++ - base;
        if (!valid[cur]) {
            ++st.skip;
            continue;
        }
        if (acc_dis[cur] < st.best) {
            st.best = acc_dis[cur];
            st.best_it = base + cur;
            st.limit = std::min(st.limit, mlesac_loop_number(tf_min_points(o.tform_type), o.confidence, m, n_inl[cur]));
        }
        ++st.trial;
    }
    st.done = !(st.trial <= st.limit && st.skip < max_skip);
}

static void ransac_batch(const double* d_p1, const double* d_p2, int64_t ldp,
                         const std::vector<int64_t>& h_ptr, const uint32_t* d_samples, int n_samples,
                         const aps_ransac_opts& o, double* d_models, uint8_t* d_mask, int32_t* d_found,
                         int32_t* d_ninl, std::vector<int>* trials_out) {
    const int n_pairs = (int)h_ptr.size() - 1;
    if (n_pairs <= 0) return;
    const int mlesac = o.method == APS_ROBUST_MLESAC ? 1 : 0;
    const int type = o.tform_type, min_pts = tf_min_points(type);
    const int64_t total_rows = h_ptr.back();
    const int64_t nh = (int64_t)n_pairs * n_samples;
    Ws<int64_t> d_ptr(n_pairs + 1);
    APS_HIP(hipMemcpyAsync(d_ptr, h_ptr.data(), (n_pairs + 1) * sizeof(int64_t), hipMemcpyHostToDevice,
                           stream()));
    Ws<double> Hs(nh * 9), merr(nh);
    Ws<uint8_t> valid(nh), valid_loc(nh), scratch(std::max<int64_t>(total_rows, 1));
    Ws<int32_t> ninl(nh), best(n_pairs), d_act(n_pairs);
    Ws<unsigned long long> med_keys(type == APS_TFORM_PROJECTIVE ? 1 : std::max<int64_t>(total_rows, 1));
    const size_t lds_bytes = 2 * 81 * 64 * sizeof(double);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        APS_HIP(hipFuncSetAttribute((const void*)ransac_fit_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        APS_HIP(hipFuncSetAttribute((const void*)ransac_finalize_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        APS_HIP(hipFuncSetAttribute((const void*)mlesac_tform_fit_kernel,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    // The reference's loop ends adaptively (a pair with 80 % inliers needs 13 trials, not 564), and only its
    // sequential part knows when.  So the draws are fitted and scored in growing chunks: after each chunk the host
    // advances every live pair's loop and only the pairs that have not ended yet get the next chunk.  The first
    // chunk is sized to one round of the fit kernel (one 64-lane workgroup per CU).
    std::vector<Replay> st(n_pairs);
    std::vector<int> act;
    for (int p = 0; p < n_pairs; ++p) {
        const int64_t m = h_ptr[p + 1] - h_ptr[p];
        st[p].limit = o.max_iter;
        if (mlesac) st[p].best = o.max_distance * (double)m;
        if (m < min_pts)
            st[p].done = true;
        else
            act.push_back(p);
    }
    std::vector<uint8_t> h_valid;
    std::vector<int32_t> h_ninl, h_best(n_pairs);
    std::vector<double> h_merr;
    g_draws_exhausted = 0;
    // (Round 5.)  The FIRST scored chunk is capped at 96 draws per pair - a pair with >= 50 % inliers ends within it - and a
    // batch small enough for one round of the fit kernel (one rank of eight: 28 pairs x 564 draws) is FITTED whole in the
    // first iteration: a fit launch costs its ~0.35 ms of latency whatever its size, a score launch costs by the draws.
    // Before, such a batch fitted and scored all 564 draws of every pair in one chunk (fit 0.36 + score 0.52 ms per rank);
    // the results do not depend on the chunking.
    int c0 = 0, nc_prev = 0, fitted = 0;
    while (!act.empty() && c0 < n_samples) {
        const int n_act = (int)act.size();
        int nc = std::max(std::max(16, 2 * nc_prev), nc_prev == 0 ? std::min(96, 16384 / n_act) : 16384 / n_act);
        nc = std::min(nc, n_samples - c0);
        const int64_t nw = (int64_t)n_act * nc;
        APS_HIP(hipMemcpyAsync(d_act, act.data(), n_act * sizeof(int), hipMemcpyHostToDevice, stream()));
        if (c0 + nc > fitted) {
            const int nf = (int64_t)n_act * (n_samples - fitted) <= 16384 ? n_samples - fitted : c0 + nc - fitted;
            const int64_t nwf = (int64_t)n_act * nf;
            Prof prof("ransac_fit");
            if (type == APS_TFORM_PROJECTIVE)
                ransac_fit_kernel<<<cdiv(nwf, 64), 64, lds_bytes, stream()>>>(d_p1, d_p2, ldp, d_ptr, d_act, n_act, fitted, nf,
                                                                               d_samples, n_samples, Hs, valid, mlesac);
            else if (mlesac)
                mlesac_tform_fit_kernel<<<cdiv(nwf, 64), 64, lds_bytes, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, d_act, n_act,
                                                                                     fitted, nf, d_samples, n_samples, Hs, valid);
            else
                tform_fit_kernel<<<cdiv(nwf, 64), 64, 0, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, d_act, n_act, fitted, nf,
                                                                      d_samples, n_samples, Hs, valid);
            fitted += nf;
        }
        check_launch("ransac_fit_kernel");
        {
            Prof prof("ransac_score");
            valid_chunk_kernel<<<cdiv(nw, 256), 256, 0, stream()>>>(d_act, n_act, c0, nc, n_samples, valid, valid_loc);
            if (mlesac && type != APS_TFORM_PROJECTIVE)
                mlesac_tform_score_kernel<<<cdiv(nw, 4), 256, 0, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, d_act, n_act, c0, nc,
                                                                              n_samples, Hs, valid_loc, o.max_distance, ninl, merr);
            else if (mlesac)
                mlesac_score_kernel<<<cdiv(nw, 4), 256, 0, stream()>>>(d_p1, d_p2, ldp, d_ptr, d_act, n_act, c0, nc,
                                                                        n_samples, Hs, valid_loc, o.max_distance, ninl, merr);
            else
                ransac_score_kernel<<<cdiv(nw, 4), 256, 0, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, d_act, n_act, c0, nc,
                                                                        n_samples, Hs, valid_loc, o.max_distance, ninl, merr);
        }
        check_launch("ransac_score_kernel");
        h_valid.resize(nw);
        h_ninl.resize(nw);
        h_merr.resize(nw);
        APS_HIP(hipMemcpyAsync(h_valid.data(), valid_loc, nw, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipMemcpyAsync(h_ninl.data(), ninl, nw * sizeof(int32_t), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipMemcpyAsync(h_merr.data(), merr, nw * sizeof(double), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        std::vector<int> next;
        for (int a = 0; a < n_act; ++a) {
            const int p = act[a];
            (mlesac ? replay_mlesac : replay_loop)(st[p], h_valid.data() + (int64_t)a * nc, h_ninl.data() + (int64_t)a * nc,
                                                   h_merr.data() + (int64_t)a * nc, c0 + nc, h_ptr[p + 1] - h_ptr[p], o);
            if (!st[p].done) next.push_back(p);
        }
        act.swap(next);
        c0 += nc;
        nc_prev = nc;
    }
    // the reference keeps drawing until trial > maxTrials or skipTrials reaches its cap (:94); here the draws are an
    // input, and running out of them first means the sequential loop would have gone on
    g_draws_exhausted = (int)act.size();
    if (trials_out) trials_out->assign(n_pairs, 0);
    for (int p = 0; p < n_pairs; ++p) {
        h_best[p] = st[p].best_it;
        if (trials_out) (*trials_out)[p] = st[p].it;
    }
    APS_HIP(hipMemcpyAsync(best, h_best.data(), n_pairs * sizeof(int32_t), hipMemcpyHostToDevice,
                           stream()));
    {
        Prof prof("ransac_finalize");
        if (type == APS_TFORM_PROJECTIVE)
            ransac_finalize_kernel<<<n_pairs, 64, 0, stream()>>>(d_p1, d_p2, ldp, d_ptr, n_samples, Hs, best, o.max_distance,
                                                                  d_models, d_mask, scratch, d_found, d_ninl, mlesac);
        else if (mlesac)
            mlesac_tform_finalize_kernel<<<n_pairs, 64, 0, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, n_samples, Hs, best,
                                                                        o.max_distance, d_models, d_mask, scratch, d_found,
                                                                        d_ninl);
        else
            tform_finalize_kernel<<<n_pairs, 64, 0, stream()>>>(type, d_p1, d_p2, ldp, d_ptr, n_samples, Hs, best,
                                                                 o.max_distance, d_models, d_mask, scratch, med_keys, d_found,
                                                                 d_ninl);
    }
    check_launch("ransac_finalize_kernel");
    APS_HIP(hipStreamSynchronize(stream()));
}

}  // namespace aps

using namespace aps;

extern "C" {

int aps_ransac_score(const double* Hs, int n_hyp, const double* p1, const double* p2, int64_t m,
                     int64_t ldp, double thr, int tform_type, int32_t* n_inl, double* mean_err,
                     uint8_t* mask) {
    return guarded([&] {
        APS_REQUIRE(tform_type >= APS_TFORM_PROJECTIVE && tform_type <= APS_TFORM_TRANSLATION, APS_E_TYPE,
                    "unknown transformationType %d", tform_type);
        APS_REQUIRE(n_hyp >= 0 && m >= 0, APS_E_ARG, "negative size");
        APS_REQUIRE(ldp >= m, APS_E_DIM, "ldp < m");
        APS_REQUIRE(n_hyp == 0 || (Hs && n_inl && mean_err), APS_E_ARG, "NULL argument");
        APS_REQUIRE(m == 0 || (p1 && p2), APS_E_ARG, "NULL points");
        ctx();
        if (n_hyp == 0) return;
        In<double> dH(Hs, (size_t)n_hyp * 9), d1(p1, (size_t)ldp + m), d2(p2, (size_t)ldp + m);
        Out<int32_t> on(n_inl, n_hyp);
        Out<double> oe(mean_err, n_hyp);
        Out<uint8_t> om(mask, (size_t)n_hyp * m);
        ransac_score_explicit_kernel<<<cdiv(n_hyp, 4), 256, 0, stream()>>>(
            tform_type, d1, d2, ldp, m, dH, n_hyp, thr, on, oe, om.present() ? om.get() : nullptr);
        check_launch("ransac_score_explicit_kernel");
        on.commit();
        oe.commit();
        om.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_ransac_draws_exhausted(void) { return g_draws_exhausted; }

int aps_gather_match_points(const double* const* kp, const int64_t* kp_count, int n_img, const int32_t* idx_a,
                            const int32_t* idx_b, const int64_t* list_start, const int64_t* work_ptr,
                            const int32_t* img_a, const int32_t* img_b, int n_work, double* pts_a, double* pts_b,
                            int64_t ldp) {
    return guarded([&] {
        APS_REQUIRE(n_img >= 0 && n_work >= 0, APS_E_ARG, "negative count");
        if (n_work == 0) return;
        APS_REQUIRE(kp && kp_count && list_start && work_ptr && img_a && img_b, APS_E_ARG, "NULL table");
        const int64_t total = work_ptr[n_work];
        APS_REQUIRE(total >= 0 && ldp >= total, APS_E_DIM, "ldp < total");
        if (total == 0) return;
        APS_REQUIRE(idx_a && idx_b && pts_a && pts_b, APS_E_ARG, "NULL list / output");
        APS_REQUIRE(is_device_ptr(idx_a) && is_device_ptr(idx_b) && is_device_ptr(pts_a) && is_device_ptr(pts_b), APS_E_ARG,
                    "the match lists and the point arrays must be device memory");
        for (int q = 0; q < n_work; ++q)
            APS_REQUIRE(img_a[q] >= 0 && img_a[q] < n_img && img_b[q] >= 0 && img_b[q] < n_img && work_ptr[q + 1] >= work_ptr[q],
                        APS_E_ARG, "bad work pair %d", q);
        ctx();
        // one table upload: [kp pointers | kp counts | list_start | work_ptr | img_a, img_b]
        std::vector<unsigned long long> tab((size_t)2 * n_img + (size_t)2 * n_work + 1 + (size_t)n_work);
        size_t o = 0;
        for (int i = 0; i < n_img; ++i) tab[o++] = (unsigned long long)(uintptr_t)kp[i];
        for (int i = 0; i < n_img; ++i) tab[o++] = (unsigned long long)kp_count[i];
        for (int q = 0; q < n_work; ++q) tab[o++] = (unsigned long long)list_start[q];
        for (int q = 0; q <= n_work; ++q) tab[o++] = (unsigned long long)work_ptr[q];
        for (int q = 0; q < n_work; ++q) tab[o++] = ((unsigned long long)(uint32_t)img_b[q] << 32) | (uint32_t)img_a[q];
        Ws<unsigned long long> dtab(tab.size());
        APS_HIP(hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream()));
        gather_match_points_kernel<<<cdiv(total, 256), 256, 0, stream()>>>(dtab, n_img, n_work, idx_a, idx_b, total, pts_a, pts_b, ldp);
        check_launch("gather_match_points_kernel");
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_ransac_draw_samples(const int64_t* counts, const uint64_t* keys, int n_pairs, int n_samples,
                            uint64_t seed, uint32_t* sample_idx) {
    return guarded([&] {
        APS_REQUIRE(n_pairs >= 0 && n_samples > 0, APS_E_ARG, "bad n_pairs/n_samples");
        if (n_pairs == 0) return;
        APS_REQUIRE(counts && sample_idx, APS_E_ARG, "NULL argument");
        ctx();
        In<int64_t> dc(counts, n_pairs);
        In<unsigned long long> dk(reinterpret_cast<const unsigned long long*>(keys), keys ? n_pairs : 0);
        Out<uint32_t> o(sample_idx, (size_t)4 * n_samples * n_pairs);
        const int64_t total = (int64_t)n_pairs * n_samples;
        draw_samples_kernel<<<cdiv(total, 256), 256, 0, stream()>>>(dc, keys ? dk.get() : nullptr, n_pairs, n_samples,
                                                                    seed, o);
        check_launch("draw_samples_kernel");
        o.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_ransac_homography_batch(const double* pts1, const double* pts2, int64_t ldp,
                                const int64_t* pair_ptr, int n_pairs, const uint32_t* sample_idx,
                                int n_samples, const aps_ransac_opts* opts, double* models,
                                uint8_t* mask, int32_t* found, int32_t* n_inl) {
    return guarded([&] {
        APS_REQUIRE(opts != nullptr, APS_E_ARG, "opts is NULL");
        check_opts(*opts);
        APS_REQUIRE(n_pairs >= 0 && n_samples > 0, APS_E_ARG, "bad n_pairs/n_samples");
        if (n_pairs == 0) return;
        APS_REQUIRE(pair_ptr && sample_idx && models && found && n_inl, APS_E_ARG, "NULL argument");
        ctx();
        std::vector<int64_t> h_ptr(n_pairs + 1);
        if (is_device_ptr(pair_ptr))
            APS_HIP(hipMemcpy(h_ptr.data(), pair_ptr, (n_pairs + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        else
            std::copy(pair_ptr, pair_ptr + n_pairs + 1, h_ptr.begin());
        APS_REQUIRE(h_ptr[0] == 0, APS_E_ARG, "pair_ptr[0] must be 0");
        for (int p = 0; p < n_pairs; ++p)
            APS_REQUIRE(h_ptr[p + 1] >= h_ptr[p], APS_E_ARG, "pair_ptr must be non-decreasing");
        const int64_t total = h_ptr.back();
        APS_REQUIRE(ldp >= total, APS_E_DIM, "ldp < total rows");
        APS_REQUIRE(total == 0 || (pts1 && pts2 && mask), APS_E_ARG, "NULL points/mask");
        In<double> d1(pts1, (size_t)ldp + total), d2(pts2, (size_t)ldp + total);
        In<uint32_t> ds(sample_idx, (size_t)4 * n_samples * n_pairs);
        Out<double> om(models, (size_t)n_pairs * 9);
        Out<uint8_t> omask(mask, (size_t)std::max<int64_t>(total, 1));
        Out<int32_t> of(found, n_pairs), on(n_inl, n_pairs);
        ransac_batch(d1, d2, ldp, h_ptr, ds, n_samples, *opts, om, omask, of, on, nullptr);
        om.commit();
        omask.commit(total);
        of.commit();
        on.commit();
    });
}

int aps_ransac_homography(const double* p1, const double* p2, int64_t m, int64_t ldp,
                          const uint32_t* sample_idx, int n_samples, const aps_ransac_opts* opts,
                          double* model, uint8_t* inlier_mask, int* is_found, int* trials_used) {
    return guarded([&] {
        APS_REQUIRE(opts != nullptr, APS_E_ARG, "opts is NULL");
        check_opts(*opts);
        APS_REQUIRE(m >= 0 && n_samples > 0, APS_E_ARG, "bad m/n_samples");
        APS_REQUIRE(ldp >= m, APS_E_DIM, "ldp < m");
        APS_REQUIRE(model && is_found && sample_idx, APS_E_ARG, "NULL argument");
        APS_REQUIRE(m == 0 || (p1 && p2 && inlier_mask), APS_E_ARG, "NULL points/mask");
        ctx();
        std::vector<int64_t> h_ptr = {0, m};
        In<double> d1(p1, (size_t)ldp + m), d2(p2, (size_t)ldp + m);
        In<uint32_t> ds(sample_idx, (size_t)4 * n_samples);
        Out<double> om(model, 9);
        Out<uint8_t> omask(inlier_mask, (size_t)std::max<int64_t>(m, 1));
        Ws<int32_t> df(1), dn(1);
        std::vector<int> trials;
        ransac_batch(d1, d2, ldp, h_ptr, ds, n_samples, *opts, om, omask, df, dn, &trials);
        int32_t f = 0;
        APS_HIP(hipMemcpy(&f, df, sizeof f, hipMemcpyDeviceToHost));
        *is_found = f;
        if (trials_used) *trials_used = trials.empty() ? 0 : trials[0];
        om.commit();
        omask.commit(m);
    });
}

}  // extern "C"
