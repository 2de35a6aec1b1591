/*
 * ransac_oracle.c — CPU restatement of PP/imageMatching/estimateTransformationRANSAC.m (projective).
 *
 * TEST INFRASTRUCTURE ONLY (see match_oracle.c).  PARITY UNPINNED: the reference has no tests and
 * relies on MATLAB's svd / mldivide / rcond / det (LAPACK, closed build) and on the unseeded global
 * RNG inside parfor workers (:96).  What is fixed here, and mirrored by the HIP path:
 *   - the random 4-subsets are an INPUT (one column per loop iteration of :94-143);
 *   - svd(A) -> V(:,end) (:214-216) is computed as the eigenvector of the smallest eigenvalue of the
 *     9x9 Gram matrix A'A by cyclic Jacobi (same subspace; Gram sums in row order);
 *   - H \ x (:472) is evaluated as adj(H)*x: only the ratio of homogeneous coordinates is used;
 *   - rcond(H) (:532) is the exact 1-norm value 1/(|H|_1 |H^-1|_1); det by first-row cofactors;
 *   - sums over inliers (mean error :117, centroid and second moments of isDegenerate :559-567) use
 *     the "wave order": 64 lane-strided partial sums (element i -> lane i%64, ascending i), combined
 *     by the xor butterfly 32,16,8,4,2,1;
 *   - (round 4) so do the sums of the projective REFIT on the inliers (:146-181 -> estimateHomography: centroids, mean
 *     distances and the Gram sums of the DLT rows; point i adds to partial i % 64).  MATLAB's mean / sum and A'*A leave the
 *     order open, a sequential order made the refit one wave's serial walk over every match of a pair (1.3 of the 3.4 ms
 *     of the RANSAC stage), and the wave order is the one the other inlier sums already use.  The MINIMAL-sample fits
 *     (four points) keep plain sequential sums;
 *   - svd of the centred n x 2 inlier matrix (:562) is the closed form from its 2x2 scatter matrix.
 * Everything is IEEE double with no contraction (-ffp-contract=off), so the device reproduces it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define DBL_EPS 2.220446049250313e-16

/* ---- wave-order reduction ------------------------------------------------------------------- */
static double wave_reduce(double p[64]) {
    double q[64];
    for (int off = 32; off > 0; off >>= 1) {
        for (int l = 0; l < 64; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof q);
    }
    return p[0];
}

/* ---- 3x3 helpers (column-major: H[r + 3c]) ---------------------------------------------------- */
#define H_(r, c) H[(r) + 3 * (c)]
static void adjugate3(const double* H, double* A) {
    A[0 + 3 * 0] = H_(1, 1) * H_(2, 2) - H_(1, 2) * H_(2, 1);
    A[0 + 3 * 1] = H_(0, 2) * H_(2, 1) - H_(0, 1) * H_(2, 2);
    A[0 + 3 * 2] = H_(0, 1) * H_(1, 2) - H_(0, 2) * H_(1, 1);
    A[1 + 3 * 0] = H_(1, 2) * H_(2, 0) - H_(1, 0) * H_(2, 2);
    A[1 + 3 * 1] = H_(0, 0) * H_(2, 2) - H_(0, 2) * H_(2, 0);
    A[1 + 3 * 2] = H_(0, 2) * H_(1, 0) - H_(0, 0) * H_(1, 2);
    A[2 + 3 * 0] = H_(1, 0) * H_(2, 1) - H_(1, 1) * H_(2, 0);
    A[2 + 3 * 1] = H_(0, 1) * H_(2, 0) - H_(0, 0) * H_(2, 1);
    A[2 + 3 * 2] = H_(0, 0) * H_(1, 1) - H_(0, 1) * H_(1, 0);
}
static double det3(const double* H) {
    const double c0 = H_(1, 1) * H_(2, 2) - H_(1, 2) * H_(2, 1);
    const double c1 = H_(1, 0) * H_(2, 2) - H_(1, 2) * H_(2, 0);
    const double c2 = H_(1, 0) * H_(2, 1) - H_(1, 1) * H_(2, 0);
    return (H_(0, 0) * c0 - H_(0, 1) * c1) + H_(0, 2) * c2;
}
static double norm1_3(const double* H) {
    double m = 0;
    for (int c = 0; c < 3; ++c) {
        const double s = (fabs(H_(0, c)) + fabs(H_(1, c))) + fabs(H_(2, c));
        if (s > m) m = s;
    }
    return m;
}

/* checkModel (:518-535) */
static int check_model(const double* H) {
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0;
    const double d = det3(H);
    if (!(fabs(d) > DBL_EPS)) return 0;
    double A[9], Inv[9];
    adjugate3(H, A);
    for (int e = 0; e < 9; ++e) Inv[e] = A[e] / d;
    const double rc = 1.0 / (norm1_3(H) * norm1_3(Inv));
    return rc > DBL_EPS;
}

/* ---- cyclic Jacobi on a symmetric 9x9 (row-major G[9*p+q]); returns V (columns = eigenvectors) */
static void jacobi9(double* G, double* V) {
    for (int p = 0; p < 9; ++p)
        for (int q = 0; q < 9; ++q) V[9 * p + q] = (p == q) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double gpq = G[9 * p + q];
                const double gpp = G[9 * p + p], gqq = G[9 * q + q];
                if (fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq))) continue;
                rotated = 1;
                const double theta = (gqq - gpp) / (2.0 * gpq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < 9; ++k) {
                    if (k == p || k == q) continue;
                    const double gkp = G[9 * k + p], gkq = G[9 * k + q];
                    const double np_ = c * gkp - s * gkq;
                    const double nq_ = s * gkp + c * gkq;
                    G[9 * k + p] = np_; G[9 * p + k] = np_;
                    G[9 * k + q] = nq_; G[9 * q + k] = nq_;
                }
                G[9 * p + p] = gpp - t * gpq;
                G[9 * q + q] = gqq + t * gpq;
                G[9 * p + q] = 0.0;
                G[9 * q + p] = 0.0;
                for (int k = 0; k < 9; ++k) {
                    const double vkp = V[9 * k + p], vkq = V[9 * k + q];
                    V[9 * k + p] = c * vkp - s * vkq;
                    V[9 * k + q] = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
}

/* normalizePoints (:579-610): sums over the SELECTED points - sequential in index order, or (wave != 0: the refit on the
 * inliers) in the wave order over the point index: point i adds to partial i % 64, the partials meet in wave_reduce */
static void normalize_sel(const double* x, const double* y, const int64_t* sel, int64_t n,
                          double* scale, double* tx, double* ty, int wave) {
    double sx = 0, sy = 0;
    if (wave) {
        double px[64], py[64];
        for (int l = 0; l < 64; ++l) px[l] = py[l] = 0;
        for (int64_t e = 0; e < n; ++e) {
            const int l = (int)(sel[e] & 63);
            px[l] = px[l] + x[sel[e]];
            py[l] = py[l] + y[sel[e]];
        }
        sx = wave_reduce(px);
        sy = wave_reduce(py);
    } else {
        for (int64_t e = 0; e < n; ++e) { sx = sx + x[sel[e]]; sy = sy + y[sel[e]]; }
    }
    const double cx = sx / (double)n, cy = sy / (double)n;
    double sd = 0;
    if (wave) {
        double pd[64];
        for (int l = 0; l < 64; ++l) pd[l] = 0;
        for (int64_t e = 0; e < n; ++e) {
            const double dx = x[sel[e]] - cx, dy = y[sel[e]] - cy;
            const int l = (int)(sel[e] & 63);
            pd[l] = pd[l] + sqrt(dx * dx + dy * dy);
        }
        sd = wave_reduce(pd);
    } else {
        for (int64_t e = 0; e < n; ++e) {
            const double dx = x[sel[e]] - cx, dy = y[sel[e]] - cy;
            sd = sd + sqrt(dx * dx + dy * dy);
        }
    }
    const double s = 1.0 / (sd / (double)n);
    *scale = s;
    *tx = -s * cx; /* T(1,3) = -scale*centroid(1) */
    *ty = -s * cy;
}

/* the upper triangle of a 9 x 9 Gram matrix accumulated in the wave order: PG[k][l], k = the (p, q >= p) entry */
typedef struct { double v[45][64]; } gram_partials;
static void gram_partials_add(gram_partials* P, int lane, const double* a) {
    int k = 0;
    for (int p = 0; p < 9; ++p)
        for (int q = p; q < 9; ++q, ++k) P->v[k][lane] = P->v[k][lane] + a[p] * a[q];
}
static void gram_partials_reduce(gram_partials* P, double* G) {
    int k = 0;
    for (int p = 0; p < 9; ++p)
        for (int q = p; q < 9; ++q, ++k) G[9 * p + q] = wave_reduce(P->v[k]);
}

/* estimateHomography (:188-225) on the points listed in sel (n >= 4).  H column-major.
 * Returns 0 if the result is not finite. */
static int fit_homography(const double* x1, const double* y1, const double* x2, const double* y2,
                          const int64_t* sel, int64_t n, double* H, int wave) {
    double s1, t1x, t1y, s2, t2x, t2y;
    normalize_sel(x1, y1, sel, n, &s1, &t1x, &t1y, wave);
    normalize_sel(x2, y2, sel, n, &s2, &t2x, &t2y, wave);
    double G[81];
    for (int e = 0; e < 81; ++e) G[e] = 0;
    gram_partials* PG = wave ? (gram_partials*)calloc(1, sizeof(gram_partials)) : NULL;
    /* rows of A in the reference's order: first all "x" rows, then all "y" rows (:209-212) */
    for (int half = 0; half < 2; ++half)
        for (int64_t e = 0; e < n; ++e) {
            const double x = s1 * x1[sel[e]] + t1x, y = s1 * y1[sel[e]] + t1y;
            const double u = s2 * x2[sel[e]] + t2x, v = s2 * y2[sel[e]] + t2y;
            double a[9];
            if (half == 0) {
                a[0] = -x; a[1] = -y; a[2] = -1; a[3] = 0; a[4] = 0; a[5] = 0;
                a[6] = x * u; a[7] = y * u; a[8] = u;
            } else {
                a[0] = 0; a[1] = 0; a[2] = 0; a[3] = -x; a[4] = -y; a[5] = -1;
                a[6] = x * v; a[7] = y * v; a[8] = v;
            }
            if (wave) {
                gram_partials_add(PG, (int)(sel[e] & 63), a);
                continue;
            }
            for (int p = 0; p < 9; ++p)
                for (int q = p; q < 9; ++q) G[9 * p + q] = G[9 * p + q] + a[p] * a[q];
        }
    if (wave) {
        gram_partials_reduce(PG, G);
        free(PG);
    }
    for (int p = 0; p < 9; ++p)
        for (int q = 0; q < p; ++q) G[9 * p + q] = G[9 * q + p];
    double V[81];
    jacobi9(G, V);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (G[9 * k + k] < G[9 * kmin + kmin]) kmin = k;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = V[9 * k + kmin];
    /* H_norm = reshape(h,3,3)' -> H_norm(r,c) = h[3r+c];  Hn = H_norm / H_norm(3,3) */
    double Hn[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Hn[r + 3 * c] = h[3 * r + c] / h[8];
    /* M = T2 \ Hn by back substitution (T2 = [s2 0 t2x; 0 s2 t2y; 0 0 1]) */
    double M[9];
    for (int c = 0; c < 3; ++c) {
        const double m2 = Hn[2 + 3 * c];
        M[2 + 3 * c] = m2;
        M[1 + 3 * c] = (Hn[1 + 3 * c] - t2y * m2) / s2;
        M[0 + 3 * c] = (Hn[0 + 3 * c] - t2x * m2) / s2;
    }
    /* H = M * T1, T1 = [s1 0 t1x; 0 s1 t1y; 0 0 1] */
    for (int r = 0; r < 3; ++r) {
        H[r + 3 * 0] = M[r + 3 * 0] * s1;
        H[r + 3 * 1] = M[r + 3 * 1] * s1;
        H[r + 3 * 2] = (M[r + 3 * 0] * t1x + M[r + 3 * 1] * t1y) + M[r + 3 * 2];
    }
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0;
    return 1;
}

/* symmetric transfer error of one correspondence (:469-481,:499-503) */
static double transfer_error(const double* H, const double* A, double x1, double y1, double x2,
                             double y2) {
    const double X = (H_(0, 0) * x1 + H_(0, 1) * y1) + H_(0, 2);
    const double Y = (H_(1, 0) * x1 + H_(1, 1) * y1) + H_(1, 2);
    const double W = (H_(2, 0) * x1 + H_(2, 1) * y1) + H_(2, 2);
    const double tx = X / W, ty = Y / W;
    const double IX = (A[0] * x2 + A[3] * y2) + A[6];
    const double IY = (A[1] * x2 + A[4] * y2) + A[7];
    const double IW = (A[2] * x2 + A[5] * y2) + A[8];
    const double ix = IX / IW, iy = IY / IW;
    const double ex = x2 - tx, ey = y2 - ty, fx = x1 - ix, fy = y1 - iy;
    const double d1 = ex * ex + ey * ey;
    const double d2 = fx * fx + fy * fy;
    double e = sqrt(d1 + d2);
    if (!isfinite(e)) e = INFINITY;
    if (fabs(W) < DBL_EPS) e = INFINITY;
    return e;
}

/* findInliers (:444-516).  mask may be NULL.  Returns the inlier count; *mean_err = mean error over
 * inliers in wave order (NaN if none). */
static int find_inliers(const double* H, const double* x1, const double* y1, const double* x2,
                        const double* y2, int64_t m, double thr, uint8_t* mask, double* mean_err) {
    double A[9];
    adjugate3(H, A);
    double pc[64], pe[64], px[64], py[64];
    for (int l = 0; l < 64; ++l) pc[l] = pe[l] = px[l] = py[l] = 0;
    for (int64_t i = 0; i < m; ++i) {
        const double e = transfer_error(H, A, x1[i], y1[i], x2[i], y2[i]);
        const int in = e < thr;
        if (mask) mask[i] = (uint8_t)in;
        if (in) {
            const int l = (int)(i & 63);
            pc[l] += 1.0;
            pe[l] = pe[l] + e;
            px[l] = px[l] + x1[i];
            py[l] = py[l] + y1[i];
        }
    }
    const double cnt = wave_reduce(pc);
    const double se = wave_reduce(pe), sx = wave_reduce(px), sy = wave_reduce(py);
    int n = (int)cnt;
    if (n >= 4) { /* isDegenerate on pts1(inliers) (:506-513, :537-574) */
        const double mx = sx / cnt, my = sy / cnt;
        double pxx[64], pxy[64], pyy[64];
        for (int l = 0; l < 64; ++l) pxx[l] = pxy[l] = pyy[l] = 0;
        for (int64_t i = 0; i < m; ++i) {
            const double e = transfer_error(H, A, x1[i], y1[i], x2[i], y2[i]);
            if (e < thr) {
                const int l = (int)(i & 63);
                const double dx = x1[i] - mx, dy = y1[i] - my;
                pxx[l] = pxx[l] + dx * dx;
                pxy[l] = pxy[l] + dx * dy;
                pyy[l] = pyy[l] + dy * dy;
            }
        }
        const double sxx = wave_reduce(pxx), sxy = wave_reduce(pxy), syy = wave_reduce(pyy);
        const double hs = 0.5 * (sxx + syy), hd = 0.5 * (sxx - syy);
        const double r = sqrt(hd * hd + sxy * sxy);
        const double l1 = hs + r;
        double l2 = hs - r;
        if (l2 < 0) l2 = 0;
        const double s1 = sqrt(l1), s2 = sqrt(l2);
        if (s2 / s1 < 1e-3) { /* :567 — NaN (0/0) compares false like MATLAB */
            if (mask) memset(mask, 0, (size_t)m);
            *mean_err = NAN;
            return 0;
        }
    }
    *mean_err = n > 0 ? se / cnt : NAN;
    return n;
}

/* ---- exported: batched scoring ------------------------------------------------------------------
 * Hs: 3x3xT column-major pages; p1, p2: m x 2 column-major with leading dimension ldp. */
ORC_API void orc_ransac_score(const double* Hs, int n_hyp, const double* p1, const double* p2,
                              int64_t m, int64_t ldp, double thr, int32_t* n_inl, double* mean_err,
                              uint8_t* mask) {
    for (int t = 0; t < n_hyp; ++t)
        n_inl[t] = find_inliers(Hs + 9 * t, p1, p1 + ldp, p2, p2 + ldp, m, thr,
                                mask ? mask + (size_t)t * m : NULL, mean_err + t);
}

ORC_API int orc_fit_homography(const double* p1, const double* p2, int64_t ldp, const int64_t* sel,
                               int64_t n, double* H) {
    return fit_homography(p1, p1 + ldp, p2, p2 + ldp, sel, n, H, 0);
}
/* the same fit with the REFIT's wave-order sums (what orc_ransac_homography runs on the inliers after the loop) */
ORC_API int orc_fit_homography_refit(const double* p1, const double* p2, int64_t ldp, const int64_t* sel,
                                     int64_t n, double* H) {
    return fit_homography(p1, p1 + ldp, p2, p2 + ldp, sel, n, H, 1);
}

ORC_API int orc_check_model(const double* H) { return check_model(H); }

/* ---- exported: the whole loop (:54-183) ---------------------------------------------------------
 * sample_idx: 4 x n_samples, 1-based, column-major; one column per loop iteration. */
ORC_API void orc_ransac_homography(const double* p1, const double* p2, int64_t m, int64_t ldp,
                                   const uint32_t* sample_idx, int n_samples, double max_distance,
                                   double confidence, int max_iter, double* model,
                                   uint8_t* inlier_mask, int* is_found, int* trials_used) {
    const double *x1 = p1, *y1 = p1 + ldp, *x2 = p2, *y2 = p2 + ldp;
    const int min_pts = 4;
    memset(inlier_mask, 0, (size_t)m);
    for (int e = 0; e < 9; ++e) model[e] = NAN;
    *is_found = 0;
    if (trials_used) *trials_used = 0;
    if (m < min_pts) return; /* :71-76 */

    int max_trials = max_iter;
    const int max_skip = max_iter * 10;
    int trial = 1, skip = 0, it = 0, best_n = 0, have_best = 0;
    double best_err = INFINITY, bestH[9];
    uint8_t* cur = (uint8_t*)malloc((size_t)m);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)m, 1);
    while (trial <= max_trials && skip < max_skip && it < n_samples) { /* :94 */
        int64_t sel[4];
        for (int k = 0; k < 4; ++k) sel[k] = (int64_t)sample_idx[4 * it + k] - 1;
        ++it;
        double H[9];
        if (!fit_homography(x1, y1, x2, y2, sel, 4, H, 0) || !check_model(H)) { /* :101-108,:138-141 */
            ++skip;
            continue;
        }
        double me;
        const int n = find_inliers(H, x1, y1, x2, y2, m, max_distance, cur, &me);
        if (n >= min_pts) { /* :114-134 */
            if (n > best_n || (n == best_n && me < best_err)) {
                best_n = n;
                best_err = me;
                have_best = 1;
                memcpy(bestH, H, sizeof bestH);
                memcpy(best_mask, cur, (size_t)m);
                const double ratio = (double)n / (double)m;
                if (ratio > 0) {
                    const double need = ceil(log(1 - confidence / 100) / log(1 - pow(ratio, min_pts)));
                    if (need < (double)max_trials) max_trials = (int)need; /* min(maxTrials, ...) */
                }
            }
        }
        ++trial;
    }
    if (trials_used) *trials_used = it;

    if (have_best && best_n >= min_pts) { /* :146-176 */
        int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)best_n);
        int64_t c = 0;
        for (int64_t i = 0; i < m; ++i)
            if (best_mask[i]) sel[c++] = i;
        double R[9];
        const int ok = fit_homography(x1, y1, x2, y2, sel, c, R, 1) && check_model(R); /* the refit: wave-order sums */
        free(sel);
        if (ok) {
            double me;
            const int n = find_inliers(R, x1, y1, x2, y2, m, max_distance, cur, &me);
            if (n >= min_pts) {
                memcpy(model, R, sizeof R);
                memcpy(inlier_mask, cur, (size_t)m);
            } else {
                memcpy(model, bestH, sizeof bestH);
                memcpy(inlier_mask, best_mask, (size_t)m);
            }
        } else {
            memcpy(model, bestH, sizeof bestH);
            memcpy(inlier_mask, best_mask, (size_t)m);
        }
        *is_found = 1;
    }
    free(cur);
    free(best_mask);
}

/* ================================================================================================
 * MLESAC (PP/imageMatching/estimateTransformationMLESAC.m, projective).  PARITY UNPINNED, like RANSAC
 * above, plus: vision.internal.ransac.computeLoopNumber (:201) is toolbox-internal; it is restated as
 * N = ceil(log10(1 - 0.01*confidence) / log10(1 - (inliers/points)^4)), intmax when the power is < eps.
 * Fixed here and mirrored by the HIP path:
 *   - draws are an INPUT (one column per iteration of :157-211);
 *   - estimateHomography (:345-387): Hartley-Zisserman normalisation scale sqrt(2)/meanDist (:653-657),
 *     Gram sums of the DLT rows in the reference's row order (per point: the "v" row, then the "u" row),
 *     null vector by the same cyclic Jacobi, T = (N2 \ (Hn / Hn(3,3))) * N1, then T ./ T(end) (:713-714);
 *   - evaluateTransform2d (:534-562): hypot is evaluated as sqrt(dx*dx + dy*dy); |w| < eps -> inf;
 *   - sum(dis) after truncation (:283-285) in wave order.
 * ================================================================================================ */
static void normalize_sel_hz(const double* x, const double* y, const int64_t* sel, int64_t n,
                             double* scale, double* tx, double* ty, double* cxo, double* cyo, int wave) {
    double sx = 0, sy = 0;
    if (wave) { /* the refit on the inliers: wave-order sums over the point index (see normalize_sel) */
        double px[64], py[64];
        for (int l = 0; l < 64; ++l) px[l] = py[l] = 0;
        for (int64_t e = 0; e < n; ++e) {
            const int l = (int)(sel[e] & 63);
            px[l] = px[l] + x[sel[e]];
            py[l] = py[l] + y[sel[e]];
        }
        sx = wave_reduce(px);
        sy = wave_reduce(py);
    } else {
        for (int64_t e = 0; e < n; ++e) { sx = sx + x[sel[e]]; sy = sy + y[sel[e]]; }
    }
    const double cx = sx / (double)n, cy = sy / (double)n;
    double sd = 0;
    if (wave) {
        double pd[64];
        for (int l = 0; l < 64; ++l) pd[l] = 0;
        for (int64_t e = 0; e < n; ++e) {
            const double dx = x[sel[e]] - cx, dy = y[sel[e]] - cy;
            const int l = (int)(sel[e] & 63);
            pd[l] = pd[l] + sqrt(dx * dx + dy * dy);
        }
        sd = wave_reduce(pd);
    } else {
        for (int64_t e = 0; e < n; ++e) {
            const double dx = x[sel[e]] - cx, dy = y[sel[e]] - cy;
            sd = sd + sqrt(dx * dx + dy * dy);
        }
    }
    const double md = sd / (double)n;
    const double s = md > 0 ? sqrt(2.0) / md : 1.0;
    *scale = s;
    *tx = -s * cx;
    *ty = -s * cy;
    *cxo = cx; /* the normalised points are (p - centroid) * scale (:667-671: p has no homogeneous row), */
    *cyo = cy; /* the matrix [s 0 -s*cx; ...] only enters the denormalisation                           */
}

static int fit_homography_mlesac(const double* x1, const double* y1, const double* x2, const double* y2,
                                 const int64_t* sel, int64_t n, double* H, int wave) {
    double s1, t1x, t1y, s2, t2x, t2y, c1x, c1y, c2x, c2y;
    normalize_sel_hz(x1, y1, sel, n, &s1, &t1x, &t1y, &c1x, &c1y, wave);
    normalize_sel_hz(x2, y2, sel, n, &s2, &t2x, &t2y, &c2x, &c2y, wave);
    double G[81];
    for (int e = 0; e < 81; ++e) G[e] = 0;
    gram_partials* PG = wave ? (gram_partials*)calloc(1, sizeof(gram_partials)) : NULL;
    for (int64_t e = 0; e < n; ++e)
        for (int half = 1; half >= 0; --half) { /* rows 2i-1 (v) and 2i (u) of :368-373; a a' is sign-blind */
            const double x = (x1[sel[e]] - c1x) * s1, y = (y1[sel[e]] - c1y) * s1;
            const double u = (x2[sel[e]] - c2x) * s2, v = (y2[sel[e]] - c2y) * s2;
            double a[9];
            if (half == 0) {
                a[0] = -x; a[1] = -y; a[2] = -1; a[3] = 0; a[4] = 0; a[5] = 0;
                a[6] = x * u; a[7] = y * u; a[8] = u;
            } else {
                a[0] = 0; a[1] = 0; a[2] = 0; a[3] = -x; a[4] = -y; a[5] = -1;
                a[6] = x * v; a[7] = y * v; a[8] = v;
            }
            if (wave) {
                gram_partials_add(PG, (int)(sel[e] & 63), a);
                continue;
            }
            for (int p = 0; p < 9; ++p)
                for (int q = p; q < 9; ++q) G[9 * p + q] = G[9 * p + q] + a[p] * a[q];
        }
    if (wave) {
        gram_partials_reduce(PG, G);
        free(PG);
    }
    for (int p = 0; p < 9; ++p)
        for (int q = 0; q < p; ++q) G[9 * p + q] = G[9 * q + p];
    double V[81];
    jacobi9(G, V);
    int kmin = 0;
    for (int k = 1; k < 9; ++k)
        if (G[9 * k + k] < G[9 * kmin + kmin]) kmin = k;
    double h[9];
    for (int k = 0; k < 9; ++k) h[k] = V[9 * k + kmin];
    double Hn[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Hn[r + 3 * c] = h[3 * r + c] / h[8];
    double M[9];
    for (int c = 0; c < 3; ++c) {
        const double m2 = Hn[2 + 3 * c];
        M[2 + 3 * c] = m2;
        M[1 + 3 * c] = (Hn[1 + 3 * c] - t2y * m2) / s2;
        M[0 + 3 * c] = (Hn[0 + 3 * c] - t2x * m2) / s2;
    }
    double T[9];
    for (int r = 0; r < 3; ++r) {
        T[r + 3 * 0] = M[r + 3 * 0] * s1;
        T[r + 3 * 1] = M[r + 3 * 1] * s1;
        T[r + 3 * 2] = (M[r + 3 * 0] * t1x + M[r + 3 * 1] * t1y) + M[r + 3 * 2];
    }
    for (int e = 0; e < 9; ++e) H[e] = T[e] / T[8];
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0; /* checkTForm (:721-738) */
    return 1;
}

static double oneway_dist(const double* H, double x1, double y1, double x2, double y2) {
    const double X = (H_(0, 0) * x1 + H_(0, 1) * y1) + H_(0, 2);
    const double Y = (H_(1, 0) * x1 + H_(1, 1) * y1) + H_(1, 2);
    const double W = (H_(2, 0) * x1 + H_(2, 1) * y1) + H_(2, 2);
    const double dx = X / W - x2, dy = Y / W - y2;
    double d = sqrt(dx * dx + dy * dy);
    if (fabs(W) < DBL_EPS) d = INFINITY;
    return d;
}

/* evaluateModel (:258-295): truncated distances, their sum (wave order) and the inlier count */
static double mlesac_eval(const double* H, const double* x1, const double* y1, const double* x2,
                          const double* y2, int64_t m, double thr, uint8_t* mask, int* n_inl) {
    double ps[64], pc[64];
    for (int l = 0; l < 64; ++l) ps[l] = pc[l] = 0;
    for (int64_t i = 0; i < m; ++i) {
        double d = oneway_dist(H, x1[i], y1[i], x2[i], y2[i]);
        if (d > thr) d = thr; /* NaN stays NaN and poisons the sum, as in MATLAB */
        const int in = d < thr;
        if (mask) mask[i] = (uint8_t)in;
        const int l = (int)(i & 63);
        ps[l] = ps[l] + d;
        if (in) pc[l] += 1.0;
    }
    *n_inl = (int)wave_reduce(pc);
    return wave_reduce(ps);
}

static int mlesac_loop_number(double confidence, int64_t num_pts, int inlier_num) {
    const double pr = pow((double)inlier_num / (double)num_pts, 4.0);
    if (pr < DBL_EPS) return 2147483647;
    const double num = log10(1.0 - 0.01 * confidence), den = log10(1.0 - pr);
    const double n = ceil(num / den);
    if (!(n < 2147483647.0)) return 2147483647;
    return n < 0 ? 0 : (int)n;
}

ORC_API double orc_mlesac_eval(const double* H, const double* p1, const double* p2, int64_t m, int64_t ldp,
                               double thr, uint8_t* mask, int* n_inl) {
    return mlesac_eval(H, p1, p1 + ldp, p2, p2 + ldp, m, thr, mask, n_inl);
}

ORC_API int orc_fit_homography_mlesac(const double* p1, const double* p2, int64_t ldp, const int64_t* sel,
                                      int64_t n, double* H) {
    return fit_homography_mlesac(p1, p1 + ldp, p2, p2 + ldp, sel, n, H, 0);
}

ORC_API void orc_mlesac_homography(const double* p1, const double* p2, int64_t m, int64_t ldp,
                                   const uint32_t* sample_idx, int n_samples, double max_distance,
                                   double confidence, int max_num_trials, double* model,
                                   uint8_t* inlier_mask, int* is_found, int* trials_used) {
    const double *x1 = p1, *y1 = p1 + ldp, *x2 = p2, *y2 = p2 + ldp;
    memset(inlier_mask, 0, (size_t)m);
    for (int e = 0; e < 9; ++e) model[e] = NAN;
    *is_found = 0;
    if (trials_used) *trials_used = 0;
    if (m < 4) return;
    int num_trials = max_num_trials;
    const int max_skip = 10000; /* setDefaultParams: maxIterations(1000) * 10, not overridable (:72) */
    int idx = 1, skip = 0, it = 0, have_best = 0;
    double best_dis = max_distance * (double)m, bestH[9];
    uint8_t* cur = (uint8_t*)malloc((size_t)m);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)m, 1);
    while (idx <= num_trials && skip < max_skip && it < n_samples) {
        int64_t sel[4];
        for (int k = 0; k < 4; ++k) sel[k] = (int64_t)sample_idx[4 * it + k] - 1;
        ++it;
        double H[9];
        if (!fit_homography_mlesac(x1, y1, x2, y2, sel, 4, H, 0)) {
            ++skip;
            continue;
        }
        int n;
        const double acc = mlesac_eval(H, x1, y1, x2, y2, m, max_distance, cur, &n);
        if (acc < best_dis) {
            best_dis = acc;
            have_best = 1;
            memcpy(bestH, H, sizeof bestH);
            memcpy(best_mask, cur, (size_t)m);
            const int num = mlesac_loop_number(confidence, m, n);
            if (num < num_trials) num_trials = num;
        }
        ++idx;
    }
    if (trials_used) *trials_used = it;
    int64_t c = 0;
    for (int64_t i = 0; i < m; ++i) c += best_mask[i];
    if (have_best && c >= 4) { /* :216-241, recomputeModelFromInliers = true */
        int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)c);
        int64_t k = 0;
        for (int64_t i = 0; i < m; ++i)
            if (best_mask[i]) sel[k++] = i;
        double R[9];
        const int ok = fit_homography_mlesac(x1, y1, x2, y2, sel, c, R, 1); /* the refit: wave-order sums */
        free(sel);
        int n = 0;
        if (ok) mlesac_eval(R, x1, y1, x2, y2, m, max_distance, cur, &n);
        if (ok && n > 0) {
            memcpy(model, R, sizeof R);
            memcpy(inlier_mask, cur, (size_t)m);
            *is_found = 1;
        }
    }
    free(cur);
    free(best_mask);
}

/* ================================================================================================
 * The other transformTypes of estimateTransformationRANSAC.m: 'affine' (:227-288), 'similarity'
 * (:290-356), 'rigid' (:358-421), 'translation' (:423-452); findInliers' one-way error for them
 * (:483-497); minimal samples 3 / 2 / 2 / 1 (getTransformParams :612-660).  PARITY UNPINNED like the
 * projective path.  Fixed here and mirrored by the HIP path:
 *   - draws: the FIRST minPoints entries of every 4-column draw are the sample;
 *   - estimateAffine's svd-based pseudo-inverse of the block matrix [P 0; 0 P] (P = [x y 1]) is evaluated on
 *     the 3x3 Gram matrix P'P by cyclic Jacobi: h = sum_k v_k (v_k' P' b) / lambda_k over the kept k.  A
 *     singular value is kept when sqrt(lambda_k) >= 1e-10 sqrt(lambda_max) (:263-267); a lambda_k at the
 *     rounding level of the Gram sums (<= 64 eps lambda_max) stands for a singular value the reference's
 *     svd reports below that threshold (exactly collinear samples) and is dropped;
 *   - [U,S,V] = svd(M) of the 2x2 cross-covariance M = pts2c' * pts1c and R = V*diag(1,det(V*U'))*U'
 *     (:322-323, :391-403) in closed form: with E = M11 + M22, A = M21 - M12, r = hypot(E, A):
 *     R = [E A; -A E] / r (the transpose of the least-squares rotation: the reference's V and U are swapped
 *     with respect to Kabsch's formula, and that is what is restated); singular values Q + T and |Q - T|,
 *     Q = hypot(E, A)/2, T = hypot(M11 - M22, M21 + M12)/2;  rigid's second svd (:406-407, re-orthogonalising
 *     an orthogonal matrix) is a rounding-level no-op and is dropped;
 *   - median (:335,:339,:445-446): exact order statistic; even counts a + (b - a)/2 (MATLAB's meanof), or
 *     (a + b)/2 when the signs differ or one is infinite; NaN if any element is NaN;
 *   - sums of a fit run over the selected points in ascending order; the Frobenius norms (:328) add all x
 *     squares, then all y squares.
 * ================================================================================================ */
enum { TF_PROJECTIVE = 0, TF_AFFINE = 1, TF_SIMILARITY = 2, TF_RIGID = 3, TF_TRANSLATION = 4 };

static int tf_min_points(int type) {
    switch (type) {
        case TF_AFFINE: return 3;
        case TF_SIMILARITY: case TF_RIGID: return 2;
        case TF_TRANSLATION: return 1;
        default: return 4;
    }
}

static int cmp_double(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
/* MATLAB median of n >= 1 values (v is sorted in place) */
static double median_inplace(double* v, int64_t n) {
    for (int64_t i = 0; i < n; ++i)
        if (isnan(v[i])) return NAN;
    qsort(v, (size_t)n, sizeof(double), cmp_double);
    if (n & 1) return v[(n - 1) / 2];
    const double a = v[n / 2 - 1], b = v[n / 2];
    const int sa = (a > 0) - (a < 0), sb = (b > 0) - (b < 0);
    if (sa != sb || isinf(a) || isinf(b)) return (a + b) / 2;
    return a + (b - a) / 2;
}

/* H = T2 \ Hn * T1 (left to right), Hn row-major here; then the exact affine last row (:284-287 etc.) */
static void denormalize_affine(const double Hn[9], double s1, double t1x, double t1y, double s2, double t2x,
                               double t2y, double* H) {
    double M[9];
    for (int c = 0; c < 3; ++c) {
        const double m2 = Hn[3 * 2 + c];
        M[2 + 3 * c] = m2;
        M[1 + 3 * c] = (Hn[3 * 1 + c] - t2y * m2) / s2;
        M[0 + 3 * c] = (Hn[3 * 0 + c] - t2x * m2) / s2;
    }
    for (int r = 0; r < 3; ++r) {
        H[r + 3 * 0] = M[r + 3 * 0] * s1;
        H[r + 3 * 1] = M[r + 3 * 1] * s1;
        H[r + 3 * 2] = (M[r + 3 * 0] * t1x + M[r + 3 * 1] * t1y) + M[r + 3 * 2];
    }
    H[2 + 3 * 0] = 0.0;
    H[2 + 3 * 1] = 0.0;
    H[2 + 3 * 2] = 1.0;
}

/* cyclic Jacobi on a symmetric 3x3 (row-major), same rotation rule as jacobi9 */
static void jacobi3(double* G, double* V) {
    for (int p = 0; p < 3; ++p)
        for (int q = 0; q < 3; ++q) V[3 * p + q] = (p == q) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double gpq = G[3 * p + q];
                const double gpp = G[3 * p + p], gqq = G[3 * q + q];
                if (fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq))) continue;
                rotated = 1;
                const double theta = (gqq - gpp) / (2.0 * gpq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                const int k = 3 - p - q; /* the one index that is neither p nor q */
                const double gkp = G[3 * k + p], gkq = G[3 * k + q];
                const double np_ = c * gkp - s * gkq;
                const double nq_ = s * gkp + c * gkq;
                G[3 * k + p] = np_; G[3 * p + k] = np_;
                G[3 * k + q] = nq_; G[3 * q + k] = nq_;
                G[3 * p + p] = gpp - t * gpq;
                G[3 * q + q] = gqq + t * gpq;
                G[3 * p + q] = 0.0;
                G[3 * q + p] = 0.0;
                for (int kk = 0; kk < 3; ++kk) {
                    const double vkp = V[3 * kk + p], vkq = V[3 * kk + q];
                    V[3 * kk + p] = c * vkp - s * vkq;
                    V[3 * kk + q] = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
}

/* estimateAffine (:227-288) */
static int fit_affine(const double* x1, const double* y1, const double* x2, const double* y2, const int64_t* sel,
                      int64_t n, double* H) {
    double s1, t1x, t1y, s2, t2x, t2y;
    normalize_sel(x1, y1, sel, n, &s1, &t1x, &t1y, 0);
    normalize_sel(x2, y2, sel, n, &s2, &t2x, &t2y, 0);
    double gxx = 0, gxy = 0, gx = 0, gyy = 0, gy = 0, g1 = 0, bu[3] = {0, 0, 0}, bv[3] = {0, 0, 0};
    for (int64_t e = 0; e < n; ++e) {
        const double x = s1 * x1[sel[e]] + t1x, y = s1 * y1[sel[e]] + t1y;
        const double u = s2 * x2[sel[e]] + t2x, v = s2 * y2[sel[e]] + t2y;
        gxx = gxx + x * x; gxy = gxy + x * y; gx = gx + x;
        gyy = gyy + y * y; gy = gy + y; g1 = g1 + 1.0;
        bu[0] = bu[0] + x * u; bu[1] = bu[1] + y * u; bu[2] = bu[2] + u;
        bv[0] = bv[0] + x * v; bv[1] = bv[1] + y * v; bv[2] = bv[2] + v;
    }
    double G[9] = {gxx, gxy, gx, gxy, gyy, gy, gx, gy, g1}, V[9];
    jacobi3(G, V);
    double lmax = G[0];
    if (G[4] > lmax) lmax = G[4];
    if (G[8] > lmax) lmax = G[8];
    const double smax = sqrt(lmax > 0 ? lmax : 0.0);
    double cu[3], cv[3];
    for (int k = 0; k < 3; ++k) {
        const double lam = G[3 * k + k];
        const double sg = sqrt(lam > 0 ? lam : 0.0);
        const int keep = sg > 0 && !(sg < 1e-10 * smax) && lam > 64.0 * DBL_EPS * lmax;
        const double du = (V[3 * 0 + k] * bu[0] + V[3 * 1 + k] * bu[1]) + V[3 * 2 + k] * bu[2];
        const double dv = (V[3 * 0 + k] * bv[0] + V[3 * 1 + k] * bv[1]) + V[3 * 2 + k] * bv[2];
        cu[k] = keep ? du / lam : 0.0;
        cv[k] = keep ? dv / lam : 0.0;
    }
    double Hn[9];
    for (int j = 0; j < 3; ++j) {
        Hn[3 * 0 + j] = (V[3 * j + 0] * cu[0] + V[3 * j + 1] * cu[1]) + V[3 * j + 2] * cu[2];
        Hn[3 * 1 + j] = (V[3 * j + 0] * cv[0] + V[3 * j + 1] * cv[1]) + V[3 * j + 2] * cv[2];
    }
    Hn[6] = 0; Hn[7] = 0; Hn[8] = 1;
    denormalize_affine(Hn, s1, t1x, t1y, s2, t2x, t2y, H);
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0;
    return 1;
}

/* estimateSimilarity (:290-356) and estimateRigid (:358-421) */
static int fit_sim_rigid(const double* x1, const double* y1, const double* x2, const double* y2, const int64_t* sel,
                         int64_t n, int rigid, double* H) {
    double s1, t1x, t1y, s2, t2x, t2y;
    normalize_sel(x1, y1, sel, n, &s1, &t1x, &t1y, 0);
    normalize_sel(x2, y2, sel, n, &s2, &t2x, &t2y, 0);
    double sx = 0, sy = 0, su = 0, sv = 0;
    for (int64_t e = 0; e < n; ++e) {
        sx = sx + (s1 * x1[sel[e]] + t1x); sy = sy + (s1 * y1[sel[e]] + t1y);
        su = su + (s2 * x2[sel[e]] + t2x); sv = sv + (s2 * y2[sel[e]] + t2y);
    }
    const double dn = (double)n;
    const double c1x = sx / dn, c1y = sy / dn, c2x = su / dn, c2y = sv / dn;
    double m11 = 0, m12 = 0, m21 = 0, m22 = 0, fax = 0, fay = 0, fbx = 0, fby = 0;
    double* q = rigid ? NULL : (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int64_t nq = 0;
    for (int64_t e = 0; e < n; ++e) {
        const double ax = (s1 * x1[sel[e]] + t1x) - c1x, ay = (s1 * y1[sel[e]] + t1y) - c1y;
        const double bx = (s2 * x2[sel[e]] + t2x) - c2x, by = (s2 * y2[sel[e]] + t2y) - c2y;
        m11 = m11 + bx * ax; m12 = m12 + bx * ay; m21 = m21 + by * ax; m22 = m22 + by * ay;
        if (!rigid) {
            fax = fax + ax * ax; fay = fay + ay * ay; fbx = fbx + bx * bx; fby = fby + by * by;
            const double ra = sqrt(ax * ax + ay * ay), rb = sqrt(bx * bx + by * by);
            if (ra > 1e-10) q[nq++] = rb / ra;
        }
    }
    const double E = m11 + m22, A = m21 - m12;
    const double r = sqrt(E * E + A * A);
    double c = E / r, sn = A / r;
    double scale = 1.0;
    if (rigid) {
        const double F = m11 - m22, Gs = m21 + m12;
        const double Q = 0.5 * r, T = 0.5 * sqrt(F * F + Gs * Gs);
        const double sv1 = Q + T, sv2 = fabs(Q - T);
        const double cond = sv1 / (sv2 > DBL_EPS ? sv2 : DBL_EPS);
        if (cond > 1e6) { c = 1.0; sn = 0.0; } /* :397-399 */
    } else {
        const double sa = sqrt(fbx + fby) / sqrt(fax + fay);
        if (nq > 0) {
            double two[2] = {sa, median_inplace(q, nq)};
            scale = median_inplace(two, 2);
        } else {
            scale = sa;
        }
        free(q);
    }
    /* R = [c sn; -sn c];  t = centroid2' - s*R*centroid1' with (s*R) formed first */
    const double r11 = scale * c, r12 = scale * sn, r21 = scale * (-sn), r22 = scale * c;
    const double tx = c2x - (r11 * c1x + r12 * c1y), ty = c2y - (r21 * c1x + r22 * c1y);
    const double Hn[9] = {r11, r12, tx, r21, r22, ty, 0, 0, 1};
    denormalize_affine(Hn, s1, t1x, t1y, s2, t2x, t2y, H);
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0;
    return 1;
}

/* estimateTranslation (:423-452) */
static int fit_translation(const double* x1, const double* y1, const double* x2, const double* y2, const int64_t* sel,
                           int64_t n, double* H) {
    double* d = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int64_t e = 0; e < n; ++e) d[e] = x2[sel[e]] - x1[sel[e]];
    const double tx = median_inplace(d, n);
    for (int64_t e = 0; e < n; ++e) d[e] = y2[sel[e]] - y1[sel[e]];
    const double ty = median_inplace(d, n);
    free(d);
    const double Hc[9] = {1, 0, 0, 0, 1, 0, tx, ty, 1};
    memcpy(H, Hc, sizeof Hc);
    return isfinite(tx) && isfinite(ty);
}

/* refit != 0: the fit on the inliers after the loop (only the projective type distinguishes it: wave-order sums) */
static int fit_tform_r(int type, const double* x1, const double* y1, const double* x2, const double* y2,
                       const int64_t* sel, int64_t n, double* H, int refit) {
    switch (type) {
        case TF_AFFINE: return fit_affine(x1, y1, x2, y2, sel, n, H);
        case TF_SIMILARITY: return fit_sim_rigid(x1, y1, x2, y2, sel, n, 0, H);
        case TF_RIGID: return fit_sim_rigid(x1, y1, x2, y2, sel, n, 1, H);
        case TF_TRANSLATION: return fit_translation(x1, y1, x2, y2, sel, n, H);
        default: return fit_homography(x1, y1, x2, y2, sel, n, H, refit);
    }
}
static int fit_tform(int type, const double* x1, const double* y1, const double* x2, const double* y2,
                     const int64_t* sel, int64_t n, double* H) {
    return fit_tform_r(type, x1, y1, x2, y2, sel, n, H, 0);
}

/* findInliers (:444-516) for the affine family and translation; projective goes to find_inliers above */
static int find_inliers_tform(int type, const double* H, const double* x1, const double* y1, const double* x2,
                              const double* y2, int64_t m, double thr, uint8_t* mask, double* mean_err) {
    if (type == TF_PROJECTIVE) return find_inliers(H, x1, y1, x2, y2, m, thr, mask, mean_err);
    double scale = 1.0; /* max(abs([pts1_homog(:); pts2_homog(:)])): the homogeneous ones are part of it (:487) */
    if (type == TF_TRANSLATION) {
        for (int64_t i = 0; i < m; ++i) {
            if (fabs(x1[i]) > scale) scale = fabs(x1[i]);
            if (fabs(y1[i]) > scale) scale = fabs(y1[i]);
            if (fabs(x2[i]) > scale) scale = fabs(x2[i]);
            if (fabs(y2[i]) > scale) scale = fabs(y2[i]);
        }
        thr = thr / scale;
    }
    double pc[64], pe[64], px[64], py[64];
    for (int l = 0; l < 64; ++l) pc[l] = pe[l] = px[l] = py[l] = 0;
    uint8_t* in_all = (uint8_t*)malloc((size_t)(m > 0 ? m : 1));
    for (int64_t i = 0; i < m; ++i) {
        const double X = (H_(0, 0) * x1[i] + H_(0, 1) * y1[i]) + H_(0, 2);
        const double Y = (H_(1, 0) * x1[i] + H_(1, 1) * y1[i]) + H_(1, 2);
        const double W = (H_(2, 0) * x1[i] + H_(2, 1) * y1[i]) + H_(2, 2);
        const double ex = x2[i] - X / W, ey = y2[i] - Y / W;
        double e = sqrt(ex * ex + ey * ey);
        if (type == TF_TRANSLATION) e = e / scale;
        if (!isfinite(e)) e = INFINITY;
        if (fabs(W) < DBL_EPS) e = INFINITY;
        const int in = e < thr;
        in_all[i] = (uint8_t)in;
        if (mask) mask[i] = (uint8_t)in;
        if (in) {
            const int l = (int)(i & 63);
            pc[l] += 1.0;
            pe[l] = pe[l] + e;
            px[l] = px[l] + x1[i];
            py[l] = py[l] + y1[i];
        }
    }
    const double cnt = wave_reduce(pc);
    const double se = wave_reduce(pe), sx = wave_reduce(px), sy = wave_reduce(py);
    const int n = (int)cnt;
    if (type == TF_AFFINE && n >= 3) { /* isDegenerate (:506-513, :537-574) */
        const double mx = sx / cnt, my = sy / cnt;
        double pxx[64], pxy[64], pyy[64];
        for (int l = 0; l < 64; ++l) pxx[l] = pxy[l] = pyy[l] = 0;
        for (int64_t i = 0; i < m; ++i)
            if (in_all[i]) {
                const int l = (int)(i & 63);
                const double dx = x1[i] - mx, dy = y1[i] - my;
                pxx[l] = pxx[l] + dx * dx;
                pxy[l] = pxy[l] + dx * dy;
                pyy[l] = pyy[l] + dy * dy;
            }
        const double sxx = wave_reduce(pxx), sxy = wave_reduce(pxy), syy = wave_reduce(pyy);
        const double hs = 0.5 * (sxx + syy), hd = 0.5 * (sxx - syy);
        const double r = sqrt(hd * hd + sxy * sxy);
        const double l1 = hs + r;
        double l2 = hs - r;
        if (l2 < 0) l2 = 0;
        if (sqrt(l2) / sqrt(l1) < 1e-3) {
            if (mask) memset(mask, 0, (size_t)m);
            free(in_all);
            *mean_err = NAN;
            return 0;
        }
    }
    free(in_all);
    *mean_err = n > 0 ? se / cnt : NAN;
    return n;
}

ORC_API int orc_tform_min_points(int type) { return tf_min_points(type); }

ORC_API int orc_fit_tform(int type, const double* p1, const double* p2, int64_t ldp, const int64_t* sel, int64_t n,
                          double* H) {
    return fit_tform(type, p1, p1 + ldp, p2, p2 + ldp, sel, n, H);
}

ORC_API void orc_ransac_score_tform(int type, const double* Hs, int n_hyp, const double* p1, const double* p2,
                                    int64_t m, int64_t ldp, double thr, int32_t* n_inl, double* mean_err,
                                    uint8_t* mask) {
    for (int t = 0; t < n_hyp; ++t)
        n_inl[t] = find_inliers_tform(type, Hs + 9 * t, p1, p1 + ldp, p2, p2 + ldp, m, thr,
                                      mask ? mask + (size_t)t * m : NULL, mean_err + t);
}

/* The whole loop (:54-183) for any transformType.  sample_idx: 4 x n_samples, 1-based, column-major; the first
 * minPoints entries of a column are the sample. */
ORC_API void orc_ransac_tform(int type, const double* p1, const double* p2, int64_t m, int64_t ldp,
                              const uint32_t* sample_idx, int n_samples, double max_distance, double confidence,
                              int max_iter, double* model, uint8_t* inlier_mask, int* is_found, int* trials_used) {
    const double *x1 = p1, *y1 = p1 + ldp, *x2 = p2, *y2 = p2 + ldp;
    const int min_pts = tf_min_points(type);
    memset(inlier_mask, 0, (size_t)m);
    for (int e = 0; e < 9; ++e) model[e] = NAN;
    *is_found = 0;
    if (trials_used) *trials_used = 0;
    if (m < min_pts) return;
    int max_trials = max_iter;
    const int max_skip = max_iter * 10;
    int trial = 1, skip = 0, it = 0, best_n = 0, have_best = 0;
    double best_err = INFINITY, bestH[9];
    uint8_t* cur = (uint8_t*)malloc((size_t)m);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)m, 1);
    while (trial <= max_trials && skip < max_skip && it < n_samples) {
        int64_t sel[4];
        int in_range = 1;
        for (int k = 0; k < min_pts; ++k) {
            sel[k] = (int64_t)sample_idx[4 * it + k] - 1;
            if (sel[k] < 0 || sel[k] >= m) in_range = 0;
        }
        ++it;
        double H[9];
        if (!in_range || !fit_tform(type, x1, y1, x2, y2, sel, min_pts, H) || !check_model(H)) {
            ++skip;
            continue;
        }
        double me;
        const int n = find_inliers_tform(type, H, x1, y1, x2, y2, m, max_distance, cur, &me);
        if (n >= min_pts) {
            if (n > best_n || (n == best_n && me < best_err)) {
                best_n = n;
                best_err = me;
                have_best = 1;
                memcpy(bestH, H, sizeof bestH);
                memcpy(best_mask, cur, (size_t)m);
                const double ratio = (double)n / (double)m;
                if (ratio > 0) {
                    const double need = ceil(log(1 - confidence / 100) / log(1 - pow(ratio, min_pts)));
                    if (need < (double)max_trials) max_trials = (int)need;
                }
            }
        }
        ++trial;
    }
    if (trials_used) *trials_used = it;
    if (have_best && best_n >= min_pts) {
        int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)best_n);
        int64_t c = 0;
        for (int64_t i = 0; i < m; ++i)
            if (best_mask[i]) sel[c++] = i;
        double R[9];
        const int ok = fit_tform_r(type, x1, y1, x2, y2, sel, c, R, 1) && check_model(R);
        free(sel);
        int use_refit = 0;
        if (ok) {
            double me;
            use_refit = find_inliers_tform(type, R, x1, y1, x2, y2, m, max_distance, cur, &me) >= min_pts;
        }
        memcpy(model, use_refit ? R : bestH, sizeof bestH);
        memcpy(inlier_mask, use_refit ? cur : best_mask, (size_t)m);
        *is_found = 1;
    }
    free(cur);
    free(best_mask);
}

/* ================================================================================================
 * MLESAC for the other transformTypes (estimateTransformationMLESAC.m): estimateAffine (:389-424, the null vector
 * of the 2n x 7 system), estimateSimilarity (:426-458, 2n x 5), estimateRigid (:460-490, Kabsch on the raw
 * points), estimateTranslation (:492-510, the MEAN displacement), evaluateTransform2d / evaluateTranslation2d
 * (:534-598), sampleSize 3 / 2 / 2 / 1 (:83-92).  Fixed here as in the projective MLESAC path, plus:
 *   - the null vectors by cyclic Jacobi on the 7x7 / 5x5 Gram matrices (rows in the reference's order: per point
 *     its "v" row, then its "u" row);
 *   - estimateRigid's svd(C), C = P1c' * P2c, R = V*diag(1, sign(det(U*V')))*U' in closed form:
 *     R = [E -A; A E] / hypot(E, A), E = C11 + C22, A = C12 - C21 (here the reference has Kabsch's order: this is
 *     the least-squares rotation);
 *   - computeLoopNumber with the sample size as the exponent.
 * ================================================================================================ */
static void jacobi_n(double* G, double* V, int N) { /* row stride N; the rotation rule of jacobi9 */
    for (int p = 0; p < N; ++p)
        for (int q = 0; q < N; ++q) V[N * p + q] = (p == q) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < N - 1; ++p)
            for (int q = p + 1; q < N; ++q) {
                const double gpq = G[N * p + q];
                const double gpp = G[N * p + p], gqq = G[N * q + q];
                if (fabs(gpq) <= 1e-300 || fabs(gpq) <= 1e-18 * sqrt(fabs(gpp * gqq))) continue;
                rotated = 1;
                const double theta = (gqq - gpp) / (2.0 * gpq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0);
                const double s = t * c;
                for (int k = 0; k < N; ++k) {
                    if (k == p || k == q) continue;
                    const double gkp = G[N * k + p], gkq = G[N * k + q];
                    const double np_ = c * gkp - s * gkq;
                    const double nq_ = s * gkp + c * gkq;
                    G[N * k + p] = np_; G[N * p + k] = np_;
                    G[N * k + q] = nq_; G[N * q + k] = nq_;
                }
                G[N * p + p] = gpp - t * gpq;
                G[N * q + q] = gqq + t * gpq;
                G[N * p + q] = 0.0;
                G[N * q + p] = 0.0;
                for (int k = 0; k < N; ++k) {
                    const double vkp = V[N * k + p], vkq = V[N * k + q];
                    V[N * k + p] = c * vkp - s * vkq;
                    V[N * k + q] = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
}

/* denormalizeTform (:705-716): (N2 \ T) * N1, then ./ T(end); Tn row-major */
static int denormalize_mlesac(const double Tn[9], double s1, double t1x, double t1y, double s2, double t2x, double t2y,
                              double* H) {
    double M[9], T[9];
    for (int c = 0; c < 3; ++c) {
        const double m2 = Tn[6 + c];
        M[2 + 3 * c] = m2;
        M[1 + 3 * c] = (Tn[3 + c] - t2y * m2) / s2;
        M[0 + 3 * c] = (Tn[c] - t2x * m2) / s2;
    }
    for (int r = 0; r < 3; ++r) {
        T[r + 3 * 0] = M[r + 3 * 0] * s1;
        T[r + 3 * 1] = M[r + 3 * 1] * s1;
        T[r + 3 * 2] = (M[r + 3 * 0] * t1x + M[r + 3 * 1] * t1y) + M[r + 3 * 2];
    }
    for (int e = 0; e < 9; ++e) H[e] = T[e] / T[8];
    for (int e = 0; e < 9; ++e)
        if (!isfinite(H[e])) return 0;
    return 1;
}

static int fit_tform_mlesac_r(int type, const double* x1, const double* y1, const double* x2, const double* y2,
                              const int64_t* sel, int64_t n, double* H, int refit) {
    if (type == TF_PROJECTIVE) return fit_homography_mlesac(x1, y1, x2, y2, sel, n, H, refit);
    const double dn = (double)n;
    if (type == TF_TRANSLATION) { /* mean(points2 - points1) */
        double sx = 0, sy = 0;
        for (int64_t e = 0; e < n; ++e) {
            sx = sx + (x2[sel[e]] - x1[sel[e]]);
            sy = sy + (y2[sel[e]] - y1[sel[e]]);
        }
        const double Hc[9] = {1, 0, 0, 0, 1, 0, sx / dn, sy / dn, 1};
        memcpy(H, Hc, sizeof Hc);
        return isfinite(H[6]) && isfinite(H[7]);
    }
    if (type == TF_RIGID) {
        double sx = 0, sy = 0, su = 0, sv = 0;
        for (int64_t e = 0; e < n; ++e) {
            sx = sx + x1[sel[e]]; sy = sy + y1[sel[e]];
            su = su + x2[sel[e]]; sv = sv + y2[sel[e]];
        }
        const double c1x = sx / dn, c1y = sy / dn, c2x = su / dn, c2y = sv / dn;
        double c11 = 0, c12 = 0, c21 = 0, c22 = 0; /* C = normPoints1' * normPoints2 */
        for (int64_t e = 0; e < n; ++e) {
            const double ax = x1[sel[e]] - c1x, ay = y1[sel[e]] - c1y;
            const double bx = x2[sel[e]] - c2x, by = y2[sel[e]] - c2y;
            c11 = c11 + ax * bx; c12 = c12 + ax * by; c21 = c21 + ay * bx; c22 = c22 + ay * by;
        }
        const double E = c11 + c22, A = c12 - c21;
        const double r = sqrt(E * E + A * A);
        const double c = E / r, sn = A / r; /* R = [c -sn; sn c] */
        const double tx = c2x - (c * c1x + (-sn) * c1y), ty = c2y - (sn * c1x + c * c1y);
        const double Hc[9] = {c, sn, 0, -sn, c, 0, tx, ty, 1};
        memcpy(H, Hc, sizeof Hc);
        for (int e = 0; e < 9; ++e)
            if (!isfinite(H[e])) return 0;
        return 1;
    }
    double s1, t1x, t1y, s2, t2x, t2y, c1x, c1y, c2x, c2y;
    normalize_sel_hz(x1, y1, sel, n, &s1, &t1x, &t1y, &c1x, &c1y, 0);
    normalize_sel_hz(x2, y2, sel, n, &s2, &t2x, &t2y, &c2x, &c2y, 0);
    const int N = type == TF_AFFINE ? 7 : 5;
    double G[49], V[49];
    for (int e = 0; e < 49; ++e) G[e] = 0;
    for (int64_t e = 0; e < n; ++e)
        for (int half = 1; half >= 0; --half) { /* the "v" row (odd rows of the constraints), then the "u" row */
            const double x = (x1[sel[e]] - c1x) * s1, y = (y1[sel[e]] - c1y) * s1;
            const double u = (x2[sel[e]] - c2x) * s2, v = (y2[sel[e]] - c2y) * s2;
            double a[7];
            if (type == TF_AFFINE) {
                if (half) { a[0] = 0; a[1] = 0; a[2] = 0; a[3] = -x; a[4] = -y; a[5] = -1; a[6] = v; }
                else { a[0] = x; a[1] = y; a[2] = 1; a[3] = 0; a[4] = 0; a[5] = 0; a[6] = -u; }
            } else {
                if (half) { a[0] = -y; a[1] = x; a[2] = 0; a[3] = -1; a[4] = v; }
                else { a[0] = x; a[1] = y; a[2] = 1; a[3] = 0; a[4] = -u; }
            }
            for (int p = 0; p < N; ++p)
                for (int q = p; q < N; ++q) G[N * p + q] = G[N * p + q] + a[p] * a[q];
        }
    for (int p = 0; p < N; ++p)
        for (int q = 0; q < p; ++q) G[N * p + q] = G[N * q + p];
    jacobi_n(G, V, N);
    int kmin = 0;
    for (int k = 1; k < N; ++k)
        if (G[N * k + k] < G[N * kmin + kmin]) kmin = k;
    double h[7];
    for (int k = 0; k < N; ++k) h[k] = V[N * k + kmin];
    double Tn[9] = {0, 0, 0, 0, 0, 0, 0, 0, 1};
    if (type == TF_AFFINE) {
        for (int k = 0; k < 6; ++k) Tn[k] = h[k] / h[6];
    } else {
        Tn[0] = h[0] / h[4]; Tn[1] = h[1] / h[4]; Tn[2] = h[2] / h[4];
        Tn[3] = -h[1] / h[4]; Tn[4] = h[0] / h[4]; Tn[5] = h[3] / h[4];
    }
    return denormalize_mlesac(Tn, s1, t1x, t1y, s2, t2x, t2y, H);
}

static double mlesac_eval_tform(int type, const double* H, const double* x1, const double* y1, const double* x2,
                                const double* y2, int64_t m, double thr, uint8_t* mask, int* n_inl) {
    if (type != TF_TRANSLATION) return mlesac_eval(H, x1, y1, x2, y2, m, thr, mask, n_inl);
    double ps[64], pc[64];
    for (int l = 0; l < 64; ++l) ps[l] = pc[l] = 0;
    for (int64_t i = 0; i < m; ++i) { /* evaluateTranslation2d (:578-598) */
        const double dx = (x1[i] + H_(0, 2)) - x2[i], dy = (y1[i] + H_(1, 2)) - y2[i];
        double d = sqrt(dx * dx + dy * dy);
        if (d > thr) d = thr;
        const int in = d < thr;
        if (mask) mask[i] = (uint8_t)in;
        const int l = (int)(i & 63);
        ps[l] = ps[l] + d;
        if (in) pc[l] += 1.0;
    }
    *n_inl = (int)wave_reduce(pc);
    return wave_reduce(ps);
}

static int mlesac_loop_number_k(int k, double confidence, int64_t num_pts, int inlier_num) {
    const double pr = pow((double)inlier_num / (double)num_pts, (double)k);
    if (pr < DBL_EPS) return 2147483647;
    const double num = log10(1.0 - 0.01 * confidence), den = log10(1.0 - pr);
    const double n = ceil(num / den);
    if (!(n < 2147483647.0)) return 2147483647;
    return n < 0 ? 0 : (int)n;
}

static int fit_tform_mlesac(int type, const double* x1, const double* y1, const double* x2, const double* y2,
                            const int64_t* sel, int64_t n, double* H) {
    return fit_tform_mlesac_r(type, x1, y1, x2, y2, sel, n, H, 0);
}

ORC_API int orc_fit_tform_mlesac(int type, const double* p1, const double* p2, int64_t ldp, const int64_t* sel, int64_t n,
                                 double* H) {
    return fit_tform_mlesac(type, p1, p1 + ldp, p2, p2 + ldp, sel, n, H);
}

ORC_API double orc_mlesac_eval_tform(int type, const double* H, const double* p1, const double* p2, int64_t m, int64_t ldp,
                                     double thr, uint8_t* mask, int* n_inl) {
    return mlesac_eval_tform(type, H, p1, p1 + ldp, p2, p2 + ldp, m, thr, mask, n_inl);
}

/* mlesac() (:94-254) for any transformationType; the first sampleSize entries of a 4-column draw are the sample */
ORC_API void orc_mlesac_tform(int type, const double* p1, const double* p2, int64_t m, int64_t ldp,
                              const uint32_t* sample_idx, int n_samples, double max_distance, double confidence,
                              int max_num_trials, double* model, uint8_t* inlier_mask, int* is_found, int* trials_used) {
    const double *x1 = p1, *y1 = p1 + ldp, *x2 = p2, *y2 = p2 + ldp;
    const int k = tf_min_points(type);
    memset(inlier_mask, 0, (size_t)m);
    for (int e = 0; e < 9; ++e) model[e] = NAN;
    *is_found = 0;
    if (trials_used) *trials_used = 0;
    if (m < k) return;
    int num_trials = max_num_trials;
    const int max_skip = 10000;
    int idx = 1, skip = 0, it = 0, have_best = 0;
    double best_dis = max_distance * (double)m, bestH[9];
    uint8_t* cur = (uint8_t*)malloc((size_t)m);
    uint8_t* best_mask = (uint8_t*)calloc((size_t)m, 1);
    while (idx <= num_trials && skip < max_skip && it < n_samples) {
        int64_t sel[4];
        int in_range = 1;
        for (int j = 0; j < k; ++j) {
            sel[j] = (int64_t)sample_idx[4 * it + j] - 1;
            if (sel[j] < 0 || sel[j] >= m) in_range = 0;
        }
        ++it;
        double H[9];
        if (!in_range || !fit_tform_mlesac(type, x1, y1, x2, y2, sel, k, H)) {
            ++skip;
            continue;
        }
        int n;
        const double acc = mlesac_eval_tform(type, H, x1, y1, x2, y2, m, max_distance, cur, &n);
        if (acc < best_dis) {
            best_dis = acc;
            have_best = 1;
            memcpy(bestH, H, sizeof bestH);
            memcpy(best_mask, cur, (size_t)m);
            const int num = mlesac_loop_number_k(k, confidence, m, n);
            if (num < num_trials) num_trials = num;
        }
        ++idx;
    }
    if (trials_used) *trials_used = it;
    int64_t c = 0;
    for (int64_t i = 0; i < m; ++i) c += best_mask[i];
    if (have_best && c >= k) {
        int64_t* sel = (int64_t*)malloc(sizeof(int64_t) * (size_t)c);
        int64_t kk = 0;
        for (int64_t i = 0; i < m; ++i)
            if (best_mask[i]) sel[kk++] = i;
        double R[9];
        const int ok = fit_tform_mlesac_r(type, x1, y1, x2, y2, sel, c, R, 1);
        free(sel);
        int n = 0;
        if (ok) mlesac_eval_tform(type, R, x1, y1, x2, y2, m, max_distance, cur, &n);
        if (ok && n > 0) {
            memcpy(model, R, sizeof R);
            memcpy(inlier_mask, cur, (size_t)m);
            *is_found = 1;
        }
    }
    free(cur);
    free(best_mask);
}
