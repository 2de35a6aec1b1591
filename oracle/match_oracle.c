/*
 * match_oracle.c — CPU restatement of the reference's float/binary descriptor matching.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker.  libaps_hip.so never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors, MATLAB
 * is not available, and flann_knn.cpp needs OpenCV 4.12 which is not in the image.  This restatement
 * follows the cited lines; where MATLAB leaves the floating-point evaluation order unspecified
 * (BLAS sgemm, sum) the order is FIXED here and documented, and the HIP path is built to the same
 * order so that indices compare bit-exactly.
 *
 * Files followed ("PP/" = /root/reference/Procedural Program/):
 *   PP/featureMatching/matchFeaturesScratch.m:105-110,170-211,217-234,322-366
 *   PP/featureMatching/featureMatchingPairwise.m:48-63
 *   PP/featureMatching/featureMatchingGlobal.m:80-97,123-161
 *   PP/mex/nearest2HammingExhaustiveMEX.cpp:16-80 (and the OMP twin, same arithmetic)
 *   PP/mex/flann_knn.cpp:118-253 (interface only; exact kNN restated in place of FLANN)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef __FMA__
#include <immintrin.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* eps('single') */
static const float EPS_F32 = 1.1920928955078125e-07f;

/* ---- normalizeRowsL2 (matchFeaturesScratch.m:217-234) ----------------------------------------
 * n = sqrt(sum(X.^2,2)) + eps('single'); Xn = X./n.  Fixed order: k ascending, s = s + x*x with a
 * separately rounded product (this file is compiled with -ffp-contract=off). */
ORC_API void orc_normalize_rows(float* X, int64_t n, int dim) {
    for (int64_t i = 0; i < n; ++i) {
        float* x = X + i * dim;
        float s = 0.f;
        for (int k = 0; k < dim; ++k) {
            const float p = x[k] * x[k];
            s = s + p;
        }
        const float nrm = sqrtf(s) + EPS_F32;
        for (int k = 0; k < dim; ++k) x[k] = x[k] / nrm;
    }
}

static void row_sq(const float* X, int64_t n, int dim, float* sq) {
    for (int64_t i = 0; i < n; ++i) {
        const float* x = X + i * dim;
        float s = 0.f;
        for (int k = 0; k < dim; ++k) {
            const float p = x[k] * x[k];
            s = s + p;
        }
        sq[i] = s;
    }
}

/* ---- nearest2SSDExhaustive (matchFeaturesScratch.m:322-366) ----------------------------------
 * A: n1 x dim, B: n2 x dim, row-major.  D2 = a2 + b2.' - 2*G (:353) evaluated left to right in f32;
 * G(i,j) = k-ascending fma chain from 0 (the order fixed for the unspecified sgemm order).
 * [best,idx] = min(D2,[],2) takes the FIRST index on ties (:356); second = min after masking (:357-358).
 * The row blocking of :343 does not change any value and is not restated.
 * idx2 is 1-based; n2 == 0 gives idx 0 and inf distances. */
ORC_API void orc_match_2nn_ssd(const float* A, int64_t n1, const float* B, int64_t n2, int dim,
                               uint32_t* idx2, float* d1, float* d2) {
    float* a2 = (float*)malloc(sizeof(float) * (size_t)(n1 > 0 ? n1 : 1));
    float* b2 = (float*)malloc(sizeof(float) * (size_t)(n2 > 0 ? n2 : 1));
    row_sq(A, n1, dim, a2);
    row_sq(B, n2, dim, b2);
    /* Bt[k][j]: lets 8 columns share one fma instruction; the per-(i,j) chain is unchanged */
    const int64_t n2p = (n2 + 7) & ~(int64_t)7;
    float* Bt = (float*)calloc((size_t)dim * (size_t)(n2p > 0 ? n2p : 8), sizeof(float));
    for (int64_t j = 0; j < n2; ++j)
        for (int k = 0; k < dim; ++k) Bt[(size_t)k * n2p + j] = B[j * dim + k];

#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        const float* a = A + i * dim;
        float best = INFINITY, second = INFINITY;
        int64_t ibest = -1;
        for (int64_t j0 = 0; j0 < n2; j0 += 8) {
            float g[8];
#ifdef __FMA__
            __m256 acc = _mm256_setzero_ps();
            for (int k = 0; k < dim; ++k)
                acc = _mm256_fmadd_ps(_mm256_set1_ps(a[k]), _mm256_loadu_ps(Bt + (size_t)k * n2p + j0),
                                      acc);
            _mm256_storeu_ps(g, acc);
#else
            for (int u = 0; u < 8; ++u) {
                float acc = 0.f;
                for (int k = 0; k < dim; ++k) acc = fmaf(a[k], Bt[(size_t)k * n2p + j0 + u], acc);
                g[u] = acc;
            }
#endif
            const int64_t lim = (n2 - j0) < 8 ? (n2 - j0) : 8;
            for (int64_t u = 0; u < lim; ++u) {
                const float s = a2[i] + b2[j0 + u];
                const float t = 2.0f * g[u];
                const float d = s - t;
                if (d < best) { /* strict: first index wins ties */
                    second = best;
                    best = d;
                    ibest = j0 + u;
                } else if (d < second) {
                    second = d;
                }
            }
        }
        if (n2 > 0 && ibest < 0) ibest = 0; /* all-inf/NaN row: MATLAB's min returns index 1 */
        idx2[i] = n2 > 0 ? (uint32_t)(ibest + 1) : 0u;
        d1[i] = best;
        d2[i] = second;
    }
    free(Bt);
    free(a2);
    free(b2);
}

/* stable merge sort of an index permutation by ascending key (MATLAB sort is stable) */
static void stable_sort_by_key(const double* key, int64_t* perm, int64_t* tmp, int64_t n) {
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (key[perm[b]] < key[perm[a]]) ? perm[b++] : perm[a++];
            while (a < mid) tmp[o++] = perm[a++];
            while (b < hi) tmp[o++] = perm[b++];
        }
        memcpy(perm, tmp, sizeof(int64_t) * (size_t)n);
    }
}

/* ---- matchFeaturesScratch, float + 'Exhaustive' (matchFeaturesScratch.m:81-215) --------------
 * normalize: 0 never, 1 always, 2 = :105 rule (max|A|>2 || max|B|>2 -> normalise both).
 * F1/F2 row-major, not modified.  Outputs 1-based; returns K. */
ORC_API int64_t orc_match_features(const float* F1, int64_t n1, const float* F2, int64_t n2, int dim,
                                   double max_ratio, double match_threshold, int unique,
                                   int normalize, uint32_t* out1, uint32_t* out2, float* metric) {
    if (n1 == 0 || n2 == 0) return 0;
    float* A = (float*)malloc(sizeof(float) * (size_t)n1 * dim);
    float* B = (float*)malloc(sizeof(float) * (size_t)n2 * dim);
    memcpy(A, F1, sizeof(float) * (size_t)n1 * dim);
    memcpy(B, F2, sizeof(float) * (size_t)n2 * dim);
    int do_norm = normalize == 1;
    if (normalize == 2) {
        float m = 0.f;
        for (int64_t e = 0; e < n1 * dim; ++e) { float v = fabsf(A[e]); if (v > m) m = v; }
        for (int64_t e = 0; e < n2 * dim; ++e) { float v = fabsf(B[e]); if (v > m) m = v; }
        do_norm = m > 2.f; /* :105 */
    }
    if (do_norm) {
        orc_normalize_rows(A, n1, dim);
        orc_normalize_rows(B, n2, dim);
    }
    uint32_t* idx2 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)n1);
    float* d1 = (float*)malloc(sizeof(float) * (size_t)n1);
    float* d2 = (float*)malloc(sizeof(float) * (size_t)n1);
    orc_match_2nn_ssd(A, n1, B, n2, dim, idx2, d1, d2);

    /* :170-178 — d1/d2 live in DOUBLE arrays in the reference (inf(N1,1), :344), so the ratio test is
     * a double comparison: dBest <= (MaxRatio*MaxRatio) * dSecond */
    const double r2 = max_ratio * max_ratio;
    int64_t* i1 = (int64_t*)malloc(sizeof(int64_t) * (size_t)n1);
    double* dk = (double*)malloc(sizeof(double) * (size_t)n1);
    int64_t m = 0;
    for (int64_t i = 0; i < n1; ++i) {
        const double b = (double)d1[i], s = (double)d2[i];
        const int keep = (b <= r2 * s) && (b <= match_threshold) && isfinite(b) && isfinite(s);
        if (keep) {
            i1[m] = i;
            dk[m] = b;
            ++m;
        }
    }
    int64_t K = 0;
    if (unique && m > 0) {
        /* :186-207 — stable ascending sort, then first-come greedy with used1/used2 */
        int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)m);
        int64_t* tmp = (int64_t*)malloc(sizeof(int64_t) * (size_t)m);
        for (int64_t e = 0; e < m; ++e) perm[e] = e;
        stable_sort_by_key(dk, perm, tmp, m);
        uint8_t* used1 = (uint8_t*)calloc((size_t)n1, 1);
        uint8_t* used2 = (uint8_t*)calloc((size_t)n2, 1);
        for (int64_t e = 0; e < m; ++e) {
            const int64_t a = i1[perm[e]];
            const int64_t b = (int64_t)idx2[a] - 1;
            if (!used1[a] && !used2[b]) {
                used1[a] = used2[b] = 1;
                out1[K] = (uint32_t)(a + 1);
                out2[K] = (uint32_t)(b + 1);
                metric[K] = (float)dk[perm[e]];
                ++K;
            }
        }
        free(perm); free(tmp); free(used1); free(used2);
    } else {
        for (int64_t e = 0; e < m; ++e) {
            out1[K] = (uint32_t)(i1[e] + 1);
            out2[K] = idx2[i1[e]];
            metric[K] = (float)dk[e];
            ++K;
        }
    }
    free(A); free(B); free(idx2); free(d1); free(d2); free(i1); free(dk);
    return K;
}

/* ---- exact kNN in place of flann_knn_win (flann_knn.cpp:229-250 output contract) ---------------
 * squared L2 via the same canonical expansion as above, ascending distance, ties -> lower index.
 * idx 1-based row-major fq x k; missing neighbours (k > ft): idx 0, dist inf. */
ORC_API void orc_knn(const float* train, int64_t ft, const float* query, int64_t fq, int dim, int k,
                     uint32_t* idx, float* dist) {
    float* t2 = (float*)malloc(sizeof(float) * (size_t)(ft > 0 ? ft : 1));
    float* q2 = (float*)malloc(sizeof(float) * (size_t)(fq > 0 ? fq : 1));
    row_sq(train, ft, dim, t2);
    row_sq(query, fq, dim, q2);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < fq; ++i) {
        const float* a = query + i * dim;
        float* bd = dist + i * k;
        uint32_t* bi = idx + i * k;
        for (int u = 0; u < k; ++u) { bd[u] = INFINITY; bi[u] = 0; }
        for (int64_t j = 0; j < ft; ++j) {
            const float* b = train + j * dim;
            float acc = 0.f;
            for (int kk = 0; kk < dim; ++kk) acc = fmaf(a[kk], b[kk], acc);
            const float s = q2[i] + t2[j];
            const float t = 2.0f * acc;
            const float d = s - t;
            /* insert keeping ascending (d, j); strict < keeps the lower index first on ties */
            int pos = k;
            while (pos > 0 && (bi[pos - 1] == 0 || d < bd[pos - 1])) --pos;
            if (pos < k) {
                for (int u = k - 1; u > pos; --u) { bd[u] = bd[u - 1]; bi[u] = bi[u - 1]; }
                bd[pos] = d;
                bi[pos] = (uint32_t)(j + 1);
            }
        }
    }
    free(t2);
    free(q2);
}

/* ---- featureMatchingGlobal per-query filter (featureMatchingGlobal.m:123-161) -----------------
 * nn_idx/nn_dist: f x k row-major (1-based idx, 0 = none).  img_idx/local_idx 1-based.
 * Appends [li lj] to pair (min(qi,j), max(qi,j)) in query order.  Outputs are written as a list of
 * (pair_i, pair_j, li, lj) rows in processing order; returns the number of rows. */
ORC_API int64_t orc_global_filter(const uint32_t* nn_idx, const float* nn_dist, int64_t f, int k,
                                  const uint32_t* img_idx, const uint32_t* local_idx, double ratio,
                                  uint32_t* out /* rows of 4 */) {
    int64_t n = 0;
    uint32_t ni[64];
    float nd[64];
    for (int64_t q = 0; q < f; ++q) {
        const uint32_t qi = img_idx[q];
        int c = 0;
        for (int u = 0; u < k && u < 64; ++u) {
            const uint32_t id = nn_idx[q * k + u];
            if (id == 0) continue;                 /* missing neighbour */
            if (id == (uint32_t)(q + 1)) continue; /* :130 remove self */
            if (img_idx[id - 1] == qi) continue;   /* :135 same image */
            ni[c] = id;
            nd[c] = nn_dist[q * k + u];
            ++c;
        }
        if (c < 2) continue; /* :140 */
        /* :145 single arithmetic: neighDist(1)/max(neighDist(2),eps('single')) > ratioThr */
        const float den = nd[1] > EPS_F32 ? nd[1] : EPS_F32;
        const float r = nd[0] / den;
        if (r > (float)ratio) continue; /* single vs double scalar: MATLAB compares in single */
        const uint32_t j = img_idx[ni[0] - 1];
        const uint32_t li = local_idx[q], lj = local_idx[ni[0] - 1];
        if (qi < j) {
            out[4 * n + 0] = qi; out[4 * n + 1] = j; out[4 * n + 2] = li; out[4 * n + 3] = lj;
        } else {
            out[4 * n + 0] = j; out[4 * n + 1] = qi; out[4 * n + 2] = lj; out[4 * n + 3] = li;
        }
        ++n;
    }
    return n;
}

/* ---- nearest2HammingExhaustiveMEX (nearest2HammingExhaustiveMEX.cpp:16-80) --------------------
 * Row-major here (A: n1 x nb bytes); the arithmetic and tie rules are those of :52-74. */
ORC_API void orc_hamming_2nn(const uint8_t* A, int64_t n1, const uint8_t* B, int64_t n2, int nb,
                             uint32_t* idx2, float* d1, float* d2) {
    uint8_t lut[256];
    for (int v = 0; v < 256; ++v) { /* :8-14 */
        uint8_t c = 0, x = (uint8_t)v;
        while (x) { c += (x & 1); x >>= 1; }
        lut[v] = c;
    }
    if (n2 == 0) { /* :42-45 */
        for (int64_t i = 0; i < n1; ++i) { idx2[i] = 0; d1[i] = NAN; d2[i] = NAN; }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n1; ++i) {
        uint16_t best = 0xFFFF, second = 0xFFFF;
        int64_t ibest = -1, isecond = -1;
        for (int64_t j = 0; j < n2; ++j) {
            uint16_t hsum = 0;
            for (int b = 0; b < nb; ++b) hsum += (uint16_t)lut[A[i * nb + b] ^ B[j * nb + b]];
            if (hsum < best) { /* :63-65 */
                second = best; isecond = ibest;
                best = hsum; ibest = j;
            } else if ((hsum <= second) && (j != ibest)) { /* :66-68 */
                second = hsum; isecond = j;
            }
        }
        if (n2 == 1 || isecond == -1) { /* :71-74 */
            second = (uint16_t)(nb * 8);
            isecond = ibest;
        }
        idx2[i] = (uint32_t)(ibest + 1);
        d1[i] = (float)best;
        d2[i] = (float)second;
    }
}

/* The uint8 branches of flann_knn.cpp (:199-223 BFMatcher(NORM_HAMMING).knnMatch, :235-240 LSH index) as ONE exact
 * Hamming k-NN: ascending distance, ties -> lower train index; missing neighbours: index 0, distance Inf (:217-218).
 * idx, dist: fq x k row-major. */
ORC_API void orc_knn_hamming(const uint8_t* train, int64_t ft, const uint8_t* query, int64_t fq, int nb, int k,
                             uint32_t* idx, float* dist) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < fq; ++i) {
        int64_t best_i[16];
        unsigned best_d[16];
        int n = 0;
        for (int64_t j = 0; j < ft; ++j) {
            unsigned d = 0;
            for (int b = 0; b < nb; ++b) d += (unsigned)__builtin_popcount((unsigned)(query[i * nb + b] ^ train[j * nb + b]));
            int pos = n;
            while (pos > 0 && d < best_d[pos - 1]) --pos; /* strictly smaller moves ahead: equal keeps the lower index first */
            if (pos >= k) continue;
            const int last = n < k ? n : k - 1;
            for (int e = last; e > pos; --e) {
                best_d[e] = best_d[e - 1];
                best_i[e] = best_i[e - 1];
            }
            best_d[pos] = d;
            best_i[pos] = j;
            if (n < k) ++n;
        }
        for (int e = 0; e < k; ++e) {
            idx[i * k + e] = e < n ? (uint32_t)(best_i[e] + 1) : 0u;
            dist[i * k + e] = e < n ? (float)best_d[e] : INFINITY;
        }
    }
}
