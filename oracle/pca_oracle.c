/*
 * pca_oracle.c — CPU restatement of nearest2ApproxFloatFast / doBlock (row a6 of SURVEY.md section 8).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  libaps_hip.so never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests or fixtures, MATLAB is not available, and `pca`, `mean`
 * and the BLAS product behind `Ablk * Bfull.'` are closed toolbox / library code whose floating-point
 * evaluation order is unspecified.  The order is FIXED here (below) and the HIP path is built to the
 * same order, so that indices and distances compare bit for bit; what the toolbox would return can
 * differ in the last bits of the basis and therefore in near-tied neighbours.
 *
 * Files followed ("PP/" = /root/reference/Procedural Program/):
 *   PP/featureMatching/matchFeaturesScratch.m:476-490   cast, PCA on B (mean 'omitnan', pca(...,'NumComponents',k),
 *                                                        both sets projected with B's mean and basis), L2 normalisation
 *   PP/featureMatching/matchFeaturesScratch.m:552-570   doBlock: G = Ablk*Bfull.', [sim1,id1] = max, G(id1) = -inf,
 *                                                        sim2 = max, d = 2 - 2*sim
 * (the block loop :499-528 only partitions the rows of A: every row's result is independent of the block size)
 *
 * Fixed evaluation order:
 *   mean        per column: f64 sums of chunks of 256 consecutive rows (ascending), chunk sums added ascending, NaN
 *               skipped and not counted; mu = (float)(sum / count)
 *   covariance  C(i,j): per chunk of 256 rows the f32 chain acc = fmaf(x(r,i)-mu(i), x(r,j)-mu(j), acc), r ascending
 *               from acc = 0; chunk values added in f64, ascending; divided by (n - 1)          [pca: cov of centred data]
 *   axes        eigenvectors of C by cyclic Jacobi rotations in f64 (p ascending, q ascending, fixed formulas and stop
 *               rule), ordered by descending eigenvalue, each signed so that its largest-magnitude entry is positive
 *               [pca's documented sign convention]; cast to f32
 *   projection  y(c) = chain over k ascending of fmaf(x(k)-mu(k), coeff(k,c), y(c))
 *   normalise   s = s + y(c)*y(c) over c ascending (separate product), y / (sqrtf(s) + eps('single'))
 *   similarity  G(i,j) = chain over c ascending of fmaf(b(j,c), a(i,c), G)
 *   top two     first maximum (lowest index on ties), then the maximum of the rest (NaN ignored, as max does)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
#define PCA_CHUNK 256

static const float EPS_F32 = 1.1920928955078125e-07f;

/* :480  muB = mean(B, 1, 'omitnan') */
static void column_means(const float* B, int64_t n, int dim, float* mu) {
    const int64_t n_chunks = (n + PCA_CHUNK - 1) / PCA_CHUNK;
    for (int c = 0; c < dim; ++c) {
        double s = 0.0;
        long long cnt = 0;
        for (int64_t b = 0; b < n_chunks; ++b) {
            double ps = 0.0;
            const int64_t r1 = (b + 1) * PCA_CHUNK < n ? (b + 1) * PCA_CHUNK : n;
            for (int64_t r = b * PCA_CHUNK; r < r1; ++r) {
                const float v = B[r * dim + c];
                if (v == v) {
                    ps += (double)v;
                    ++cnt;
                }
            }
            s += ps;
        }
        mu[c] = (float)(s / (double)cnt);
    }
}

/* the covariance pca() diagonalises: centred rows, divided by n - 1 */
static void covariance(const float* B, int64_t n, int dim, const float* mu, double* cov) {
    const int64_t n_chunks = (n + PCA_CHUNK - 1) / PCA_CHUNK;
    float* bc = (float*)malloc((size_t)PCA_CHUNK * dim * sizeof(float));
    float* part = (float*)malloc((size_t)dim * dim * sizeof(float));
    memset(cov, 0, (size_t)dim * dim * sizeof(double));
    for (int64_t b = 0; b < n_chunks; ++b) {
        const int64_t r0 = b * PCA_CHUNK, r1 = r0 + PCA_CHUNK < n ? r0 + PCA_CHUNK : n;
        for (int64_t r = r0; r < r1; ++r)
            for (int k = 0; k < dim; ++k) bc[(r - r0) * dim + k] = B[r * dim + k] - mu[k];
#pragma omp parallel for schedule(static)
        for (int i = 0; i < dim; ++i)
            for (int j = 0; j < dim; ++j) {
                float acc = 0.f;
                for (int64_t r = 0; r < r1 - r0; ++r) acc = fmaf(bc[r * dim + i], bc[r * dim + j], acc);
                part[i * dim + j] = acc;
            }
        for (int e = 0; e < dim * dim; ++e) cov[e] += (double)part[e];
    }
    const double inv = 1.0 / (double)(n - 1 > 1 ? n - 1 : 1);
    for (int e = 0; e < dim * dim; ++e) cov[e] *= inv;
    free(bc);
    free(part);
}

/* symmetric eigen-problem, cyclic Jacobi; A (n x n) is destroyed, its diagonal ends as the eigenvalues, V holds the vectors */
static void jacobi_eigh(double* A, int n, double* V) {
    memset(V, 0, (size_t)n * n * sizeof(double));
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < n; ++p) {
            diag += A[(size_t)p * n + p] * A[(size_t)p * n + p];
            for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
        }
        if (off <= 1e-36 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = cs * akp - sn * akq;
                    A[(size_t)k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = cs * apk - sn * aqk;
                    A[(size_t)q * n + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = cs * vkp - sn * vkq;
                    V[(size_t)k * n + q] = sn * vkp + cs * vkq;
                }
            }
    }
}

/* :481-482  coeff = pca(B - muB, 'NumComponents', k): coeff[k * ncomp + c] */
ORC_API void orc_pca_basis(const float* B, int64_t n, int dim, int ncomp, float* mu, float* coeff, double* cov_out) {
    double* cov = (double*)malloc((size_t)dim * dim * sizeof(double));
    double* V = (double*)malloc((size_t)dim * dim * sizeof(double));
    int* order = (int*)malloc((size_t)dim * sizeof(int));
    column_means(B, n, dim, mu);
    covariance(B, n, dim, mu, cov);
    if (cov_out) memcpy(cov_out, cov, (size_t)dim * dim * sizeof(double));
    jacobi_eigh(cov, dim, V);
    for (int i = 0; i < dim; ++i) order[i] = i;
    for (int i = 1; i < dim; ++i) { /* stable insertion sort, descending eigenvalue */
        const int o = order[i];
        int j = i - 1;
        while (j >= 0 && cov[(size_t)order[j] * dim + order[j]] < cov[(size_t)o * dim + o]) {
            order[j + 1] = order[j];
            --j;
        }
        order[j + 1] = o;
    }
    /* pca() of n observations returns min(n - 1, NumComponents) columns (centred data has rank <= n - 1; MATLAB's default
     * 'Economy' form, matchFeaturesScratch.m:481-482 passes no other): the remaining columns of coeff stay zero, which projects
     * every row to exactly 0 there - the sums, norms and cosines of a basis without those columns (round 6) */
    memset(coeff, 0, (size_t)dim * ncomp * sizeof(float));
    for (int c = 0; c < ncomp && c < n - 1; ++c) {
        const int col = order[c];
        int big = 0;
        for (int k = 1; k < dim; ++k)
            if (fabs(V[(size_t)k * dim + col]) > fabs(V[(size_t)big * dim + col])) big = k;
        const double sgn = V[(size_t)big * dim + col] < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < dim; ++k) coeff[(size_t)k * ncomp + c] = (float)(sgn * V[(size_t)k * dim + col]);
    }
    free(cov);
    free(V);
    free(order);
}

/* :483-484, :488-489  (X - muB) * coeff, then rows / (sqrt(sum(rows.^2, 2)) + eps('single')); ncomp == 0: no projection */
static void project_normalise(const float* X, int64_t n, int dim, const float* mu, const float* coeff, int ncomp, float* Y) {
    const int nc = ncomp > 0 ? ncomp : dim;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        float* y = Y + r * nc;
        if (ncomp > 0) {
            for (int c = 0; c < nc; ++c) {
                float acc = 0.f;
                for (int k = 0; k < dim; ++k) acc = fmaf(X[r * dim + k] - mu[k], coeff[(size_t)k * ncomp + c], acc);
                y[c] = acc;
            }
        } else {
            for (int c = 0; c < nc; ++c) y[c] = X[r * dim + c];
        }
        float s = 0.f;
        for (int c = 0; c < nc; ++c) {
            const float p = y[c] * y[c];
            s = s + p;
        }
        const float nrm = sqrtf(s) + EPS_F32;
        for (int c = 0; c < nc; ++c) y[c] = y[c] / nrm;
    }
}

/* nearest2ApproxFloatFast (:442-528) with doBlock (:552-570); A n1 x dim, B n2 x dim row-major; idx2 1-based */
ORC_API void orc_pca2nn(const float* A, int64_t n1, const float* B, int64_t n2, int dim, int n_components, int use_pca, uint32_t* idx2,
                        float* d1, float* d2, float* mu_out, float* coeff_out) {
    const int project = use_pca && dim > n_components; /* :478 */
    const int ncomp = project ? n_components : 0, nc = project ? n_components : dim;
    float* mu = (float*)calloc((size_t)dim, sizeof(float));
    float* coeff = (float*)calloc((size_t)dim * (ncomp > 0 ? ncomp : 1), sizeof(float));
    if (project) orc_pca_basis(B, n2, dim, ncomp, mu, coeff, NULL);
    if (project && mu_out) memcpy(mu_out, mu, (size_t)dim * sizeof(float));
    if (project && coeff_out) memcpy(coeff_out, coeff, (size_t)dim * ncomp * sizeof(float));
    float* YA = (float*)malloc((size_t)n1 * nc * sizeof(float));
    float* YB = (float*)malloc((size_t)n2 * nc * sizeof(float));
    project_normalise(A, n1, dim, mu, coeff, ncomp, YA);
    project_normalise(B, n2, dim, mu, coeff, ncomp, YB);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t i = 0; i < n1; ++i) {
        const float* a = YA + i * nc;
        float sim1 = -INFINITY, sim2 = -INFINITY;
        int64_t id1 = -1;
        /* [sim1, id1] = max(G, [], 2): the first maximum; G(id1) = -inf; sim2 = max of the rest.  One pass: a value equal to
         * the running maximum is "the rest" */
        for (int64_t j = 0; j < n2; ++j) {
            const float* b = YB + j * nc;
            float g = 0.f;
            for (int c = 0; c < nc; ++c) g = fmaf(b[c], a[c], g);
            if (g > sim1) {
                sim2 = sim1;
                sim1 = g;
                id1 = j;
            } else if (g > sim2) {
                sim2 = g;
            }
        }
        idx2[i] = id1 < 0 ? 1u : (uint32_t)(id1 + 1);
        d1[i] = 2.0f - 2.0f * sim1;
        d2[i] = 2.0f - 2.0f * sim2;
    }
    free(mu);
    free(coeff);
    free(YA);
    free(YB);
}
