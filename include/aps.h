/*
 * aps.h — C ABI of libaps_hip.so, the MI355X (gfx950) stitching hot path.
 *
 * This is the drop-in boundary for AutoPanoStitch's hot path.  The reference's only FFI is the
 * MATLAB mex gateway (PP/mex/flann_knn.cpp:118-119, PP/mex/nearest2HammingExhaustiveMEX.cpp:16);
 * every entry point below is what a mex shim for the cited reference function binds (the shims and
 * the shadowing .m wrappers are in matlab/, described in INTEGRATION.md).  "PP/" abbreviates
 * "/root/reference/Procedural Program/".
 *
 * Conventions (all functions):
 *   - extern "C", return int status: APS_OK (0) or a negative APS_E_* code; a human-readable
 *     message for the calling thread is available from aps_last_error().  Nothing throws or
 *     long-jumps across the boundary (the mex shim turns a non-zero status into
 *     mexErrMsgIdAndTxt("aps:<kind>", aps_last_error()), as flann_knn.cpp:130-166 does).
 *   - Pointers are plain pointers; each may point to HOST memory (what a mex shim holds) or to
 *     DEVICE memory (what a resident pipeline holds).  The library asks the HIP runtime which it
 *     is (hipPointerGetAttributes) and stages host buffers through HBM itself.  No torch types.
 *   - The caller owns every buffer; outputs are caller-allocated; the library keeps no pointer
 *     after return.  Data-dependent output sizes use (capacity in, count out); if the capacity
 *     is too small the call fails with APS_E_CAP and *count holds the needed size.
 *   - Matrices carry an explicit layout + leading dimension.  APS_COLMAJOR is MATLAB's layout
 *     (element (i,k) at p[i + k*ld], flann_knn.cpp:111-112); APS_ROWMAJOR is C/numpy/torch
 *     (element (i,k) at p[i*ld + k]).
 *   - Indices crossing the boundary are 1-based uint32 like the reference's mex outputs
 *     (flann_knn.cpp:214,248; nearest2HammingExhaustiveMEX.cpp:76); 0 means "none".
 *   - Thread-safe and re-entrant: per-thread HIP stream and workspace; no global mutable state
 *     besides the device selection of the calling thread.
 *   - There is NO CPU fallback in this library.  Without a usable gfx950 device every compute
 *     entry point returns APS_E_DEVICE.
 */
#ifndef APS_H_
#define APS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APS_VERSION 100 /* 0.1.0 */

/* ---- status codes -------------------------------------------------------------------------- */
enum {
    APS_OK = 0,
    APS_E_ARG = -1,      /* bad argument value (null pointer, negative size, unknown enum)       */
    APS_E_DIM = -2,      /* dimension mismatch (flann_knn:dim, hamm2nn:cols)                      */
    APS_E_TYPE = -3,     /* unsupported element type / layout                                     */
    APS_E_OOM = -4,      /* device or host allocation failed (renderPanorama.m:245-266 semantics) */
    APS_E_DEVICE = -5,   /* no gfx950 device / HIP runtime error                                  */
    APS_E_INTERNAL = -6, /* invariant violated inside the library                                 */
    APS_E_CAP = -7       /* output capacity too small; needed size reported through *count        */
};

enum { APS_COLMAJOR = 0, APS_ROWMAJOR = 1 };

/* ---- library / device ---------------------------------------------------------------------- */
int aps_version(void);
const char* aps_last_error(void);
/* Number of usable gfx950 devices (0 if none; never fails). */
int aps_device_count(void);
/* Select the device for the calling thread.  The selection also becomes the process-wide default that threads which
 * never called aps_set_device start on (worker pools, parpool('Threads') workers); before any call the default is env
 * APS_DEVICE, else 0. */
int aps_set_device(int device);
/* Select the device for the calling thread ONLY (the process-wide default is left alone): what a worker thread of a
 * process that drives several GPUs calls before its first entry point, so that it does not inherit whichever device
 * another thread selected last. */
int aps_set_thread_device(int device);
/* Recreate the calling thread's own stream at a priority of the device's range: level > 0 the highest, < 0 the lowest,
 * 0 the middle (the default).  For a host that runs two stages side by side from two threads and wants the short launches
 * of one not to queue behind the other's kernels (the runtime keeps separate hardware queues per priority level). */
int aps_set_thread_stream_priority(int level);
/* The device the calling thread is bound to (binding it to the default first if it has none); < 0 on error. */
int aps_get_device(void);
/* Run the calling thread's work on an existing HIP stream (e.g. torch's current stream);
 * NULL restores the library's own per-thread stream. */
int aps_set_stream(void* hip_stream);
/* Block until the calling thread's stream has drained. */
int aps_synchronize(void);
/* Release the calling thread's cached device workspace. */
int aps_release_workspace(void);
/* Per-thread event timer on the library's stream (used by bench.py for in-stream kernel timing):
 * aps_timer_begin() records an event, aps_timer_end() records another, synchronises and returns
 * the elapsed milliseconds between them. */
int aps_timer_begin(void);
int aps_timer_end(float* ms);
/* Per-kernel timing on the stream the kernels are launched on (what bench.py's `roofline` entry uses):
 * while enabled, every launch site of a named hot kernel is bracketed by HIP events.
 * aps_profile_get sums the elapsed time of all recorded launches whose name equals `name`
 * (e.g. "match2nn", "sift_blur", "warp_layer", "mb_blur", ...) and returns the launch count.
 * on = 1: every launch site; on = 2: only the per-batch sites (matching, RANSAC, coverage, crop, BA, gain) - the
 * per-image and per-tile chains issue thousands of launches per stitch and their event records cost a few percent;
 * on = 0: off. */
int aps_profile_enable(int on);
int aps_profile_reset(void);
int aps_profile_get(const char* name, double* total_ms, int* launches);
/* The recorded launches of `name` one by one, in launch order: ms[0 .. min(*count, cap) - 1]; *count = how many there are.
 * (bench.py: the dominant kernel's duration per step - its median and minimum, not only the mean.) */
int aps_profile_series(const char* name, double* ms, int cap, int* count);
/* Writes a ';'-separated list of the kernel names recorded since the last reset into buf. */
int aps_profile_names(char* buf, int buf_len);

/* ============================================================================================
 * (1) Descriptor matching — PP/featureMatching/matchFeaturesScratch.m
 * ============================================================================================ */

/* a5: nearest2SSDExhaustive (matchFeaturesScratch.m:322-366).
 *   D2(i,j) = (a2(i) + b2(j)) - 2*G(i,j),  G = A*B'  in f32;
 *   idx2(i) = argmin_j D2(i,j) (first index on ties, :356), d1 = that minimum,
 *   d2 = min over j != idx2(i) (inf if n2 == 1).
 * Arithmetic contract (the canonical order the oracle restates): G(i,j) is the k-ascending f32 fma
 * chain acc = fmaf(A(i,k), B(j,k), acc) from acc = 0 — exactly what v_mfma_f32_32x32x2_f32 computes;
 * a2/b2 are k-ascending sums s = s + x*x (separate multiply and add).
 * idx2: 1-based uint32[n1]; d1, d2: f32[n1]. */
int aps_match_2nn_ssd(const float* A, int64_t n1, int64_t lda, const float* B, int64_t n2,
                      int64_t ldb, int dim, int layout, uint32_t* idx2, float* d1, float* d2);

/* a6: nearest2ApproxFloatFast + doBlock (matchFeaturesScratch.m:442-573), the 'pca2nn' back end of Method = 'Approximate':
 *   muB = mean(B,1,'omitnan'); coeff = pca(B - muB, 'NumComponents', n_components); both sets centred with muB and
 *   projected (:480-484; skipped when use_pca == 0 or dim <= n_components, :478); rows / (sqrt(sum(row.^2)) + eps('single'))
 *   (:488-489); G = A*B'; idx2 = first argmax_j G(i,j), the second similarity = max of the rest; d = 2 - 2*sim (:552-570).
 * Arithmetic contract (the canonical order the oracle restates; MATLAB leaves it open):
 *   mean / covariance sums per chunk of 256 rows in ascending row order (the covariance chunk as the f32 fma chain of
 *   v_mfma_f32_32x32x2_f32), chunk partials added in f64 in ascending order, covariance / (n2 - 1);
 *   principal axes = eigenvectors of that covariance by cyclic Jacobi rotations in f64 on the HOST (fixed order), sorted by
 *   descending eigenvalue, each signed so that its largest-magnitude entry is positive (pca's convention), cast to f32;
 *   projection = k-ascending f32 fma chain; squared norm s = s + y*y over ascending components;
 *   G(i,j) = component-ascending f32 fma chain (one MFMA chain).
 *   Fewer than n_components + 1 rows in B: pca() returns min(n2 - 1, n_components) columns (the centred data has rank
 *   <= n2 - 1); the other columns of the basis are ZERO here, which projects every row of A and of B to exactly 0 on them -
 *   the same norms, similarities and distances as the reference's narrower basis (n2 == 1: no column at all, d1 = 2).
 * idx2: 1-based uint32[n1]; d1, d2: f32[n1] (d2 = +inf when n2 == 1).  mu_out (f32[dim]) and coeff_out (f32[dim x
 * n_components], row-major; columns min(n2 - 1, n_components) .. are zero) are optional (NULL) and filled only when the
 * projection runs.  n1, n2 >= 1. */
int aps_match_pca2nn(const float* A, int64_t n1, int64_t lda, const float* B, int64_t n2, int64_t ldb, int dim, int layout,
                     int n_components, int use_pca, uint32_t* idx2, float* d1, float* d2, float* mu_out, float* coeff_out);

/* Options of the a4 driver (matchFeaturesScratch.m:59-78 name/value pairs). */
typedef struct aps_match_opts {
    double max_ratio;       /* 'MaxRatio'       (inputs.m:59  Ratiothreshold = 0.6).  f64 like MATLAB's scalars:
                               the test is d1 <= MaxRatio^2 * d2 with MaxRatio^2 evaluated in double
                               (matchFeaturesScratch.m:170-173); a float field would move the boundary. */
    double match_threshold; /* 'MatchThreshold' (inputs.m:55  Matchingthreshold = 1.5; raw SSD), f64   */
    int unique;            /* 'Unique'         (featureMatchingPairwise.m:113: true)               */
    int normalize;         /* 0 never, 1 always, 2 = the reference's rule: L2-normalise both sets
                              iff max|A| > 2 or max|B| > 2 (matchFeaturesScratch.m:105-110)        */
} aps_match_opts;

/* a4 + a5: matchFeaturesScratch(F1, F2, 'Method','Exhaustive', ...) for float descriptors
 * (matchFeaturesScratch.m:81-215): conditional row normalisation x./(sqrt(sum(x.^2))+eps('single'))
 * (:232-233), exhaustive 2-NN, keep iff d1 <= r^2*d2 (:173-174) and d1 <= MatchThreshold (:177) and
 * both finite (:178), then greedy one-to-one by ascending d with stable order (:186-207).
 * Outputs: idx1/idx2 1-based uint32[cap], metric f32[cap], *count = K.  cap >= n1 always suffices. */
int aps_match_features(const float* F1, int64_t n1, int64_t ld1, const float* F2, int64_t n2,
                       int64_t ld2, int dim, int layout, const aps_match_opts* opts,
                       uint32_t* idx1, uint32_t* idx2, float* metric, int64_t cap, int64_t* count);

/* Diagnostics of the calling thread's most recent filtered matching call (aps_match_features / _pairwise / _pairs):
 * rows = (A row, pair) combinations the int8 screening pre-pass looked at (0 when it did not run: APS_MATCH_NO_SCREEN,
 * APS_MATCH_MODE=f32), survivors = those it could NOT prove to fail the ratio / threshold filter and handed to the exact
 * path.  The screen never changes a result (matchFeaturesScratch.m:170-178 is evaluated on exact distances for every
 * row that can pass it); the counters exist for benchmarks and for tests that guard against a silently disabled screen. */
int aps_match_screen_stats(int64_t* rows, int64_t* survivors);

/* ... and of the same call: jobs = the (A set, B set) jobs the screening pass ran, exact_jobs = those it ran on EXACT int8
 * codes.  A set whose every row is a vector of integers 0 .. 255 divided by its f32 norm, bit for bit (SIFT descriptors as
 * OpenCV quantises them and the SIFT stage hands them over; detected per row, no hint needed), is coded as u - 128 without any
 * rounding, and a job of two such sets bounds its distances from exact integer dot products: the only slack left is the
 * spread of the column set's norms.  Any other job takes the rounded codes with their error terms.  Same results either
 * way; APS_MATCH_NO_EXACT=1 switches the exact codes off (A/B).  For benchmarks and tests. */
int aps_match_screen_exact_jobs(int64_t* jobs, int64_t* exact_jobs);

/* Diagnostics: the eight statistics words the matcher's proofs take from ONE descriptor set (n x 128 f32, `normalize` != 0:
 * rows L2-normalised first as matchFeaturesScratch.m:232-233 does), as the preparation kernels compute them (maxima over
 * all rows, folded from per-workgroup maxima): stats[0] = max ||x||^2 (canonical k-ascending f32 sum), [1] = max ||x - f16(x)||
 * (rounded up), [2], [3] = residuals of the f16 norm pieces (0 for ordinary data), [4] = max x, [5] = max column-side int8
 * residual norm (rounded up), [6] = min ||x||^2, [7] = min x.  Host output.  For tests: a statistic that misses a row makes
 * a bound unsound without changing any result on ordinary data. */
int aps_match_set_stats(const float* X, int64_t n, int64_t ld, int layout, int normalize, float* stats);

/* Diagnostics of the co-residency rule: registers per lane and the workgroup bound of the int8
 * screening kernel as the LOADED code object declares them (hipFuncGetAttributes).  shape = 16 (v_mfma_i32_16x16x64_i8, the
 * default) or 32 (v_mfma_i32_32x32x32_i8, APS_SCREEN_SHAPE=32); bounds_pass != 0: the pooled matcher's instantiation.  The
 * kernels must hold 256 registers at 512 threads per workgroup so that no other kernel's waves share a SIMD with int8-MFMA
 * waves; the library refuses to launch them otherwise (APS_E_INTERNAL). */
int aps_match_screen_kernel_regs(int shape, int bounds_pass, int* num_regs, int* max_threads_per_block);

/* a3: featureMatchingPairwise (featureMatchingPairwise.m:48-63): all upper-triangular image pairs
 * in the reference's order (column-major linear index of triu(.,1): (1,2),(1,3),(2,3),(1,4),...),
 * each through aps_match_features' rule, in ONE batched launch sequence.
 *   desc[i]  : descriptors of image i, counts[i] x dim, row stride ld[i] (layout as above)
 *   pair_ptr : int64[n_pairs+1] CSR offsets into idx_i/idx_j/metric, n_pairs = n_img*(n_img-1)/2
 *   idx_i/idx_j : 1-based feature indices in image i / j (i < j), metric: SSD
 * cap >= sum over pairs of counts[i] always suffices. */
int aps_match_pairwise(const float* const* desc, const int64_t* counts, const int64_t* ld,
                       int n_img, int dim, int layout, const aps_match_opts* opts,
                       int64_t* pair_ptr, uint32_t* idx_i, uint32_t* idx_j, float* metric,
                       int64_t cap, int64_t* count);

/* The same for an explicit list of image pairs (0-based image ids, pair_a[p] != pair_b[p]); this is what
 * lets the n x n pair matrix of featureMatchingPairwise.m:48 be tiled across GPUs: every rank holds all
 * descriptors (after the all-gather) and matches its own slice of the pair list.  For each pair the rows
 * of image pair_a[p] are matched against image pair_b[p]; idx_a/idx_b are 1-based feature indices. */
int aps_match_pairs(const float* const* desc, const int64_t* counts, const int64_t* ld, int n_img,
                    int dim, int layout, const int32_t* pair_a, const int32_t* pair_b, int64_t n_pairs,
                    const aps_match_opts* opts, int64_t* pair_ptr, uint32_t* idx_a, uint32_t* idx_b,
                    float* metric, int64_t cap, int64_t* count);

/* a8 kNN: [idx, dist] = flann_knn_win(train, query, k, 'flann', trees, checks) for float
 * descriptors (flann_knn.cpp:118-253; caller featureMatchingGlobal.m:108-117), with an EXACT
 * search in place of OpenCV's randomized kd-forest: squared-L2, ascending, ties -> lower index.
 * idx: 1-based uint32 Fq x k, dist: f32 Fq x k, both in `layout` with leading dimension ldo. */
int aps_knn_global(const float* train, int64_t ft, int64_t ldt, const float* query, int64_t fq,
                   int64_t ldq, int dim, int layout, int k, uint32_t* idx, float* dist,
                   int64_t ldo);
/* The same search for featureMatchingGlobal's own use (featureMatchingGlobal.m:106-147: the pool against itself, then the
 * per-query filter at ratioThr).  pool: the normalised descriptors of all images back to back, img_off[n_img + 1] the row
 * offsets of the images (img_off[0] = 0, img_off[n_img] = f).  A query that the filter PROVABLY drops at `ratio` - an
 * int8 screening pass bounds its two smallest cross-image distances, :129-147 - may come back as k copies of itself
 * (distance 0): the filter removes them as self matches, fewer than two candidates remain, the query is skipped exactly
 * as the reference skips it.  Every other row gets its exact k nearest (k <= 4), bit-identical to aps_knn_global.  The
 * output of aps_global_filter on this table therefore equals its output on aps_knn_global's.  Pools whose (row, image)
 * table would exceed 2^31 slots (f x n_img; BASELINE configs[4]: 5.4 M rows x 500 images) are searched in several passes
 * over ranges of query images, every pass against all images: same result, bounded workspace. */
int aps_knn_global_screened(const float* pool, int64_t f, int64_t ld, int dim, int layout, const int64_t* img_off, int n_img,
                            float ratio, int k, uint32_t* idx, float* dist, int64_t ldo);
/* Diagnostics of the calling thread's most recent aps_knn_global_screened call: rows of the pool, rows that were searched. */
int aps_knn_global_screen_stats(int64_t* rows, int64_t* survivors);

/* a8 kNN, binary descriptors: [idx, dist] = flann_knn_win(train_u8, query_u8, k, 'bf' | 'flann', ...) - the uint8 branches
 * of flann_knn.cpp: cv::BFMatcher(NORM_HAMMING).knnMatch (:199-223) and the LSH index (:235-240), both replaced by ONE exact
 * brute-force Hamming k-NN (ascending distance, ties -> lower index; the LSH index is approximate and seed dependent).
 * nbytes <= 64 (ORB 32, BRISK 64).  idx: 1-based uint32 Fq x k, dist: f32 Fq x k in `layout` with leading dimension ldo;
 * a query with fewer than k train rows gets index 0 / Inf in the missing slots (:217-218). */
int aps_knn_hamming(const uint8_t* train, int64_t ft, int64_t ldt, const uint8_t* query, int64_t fq, int64_t ldq,
                    int nbytes, int layout, int k, uint32_t* idx, float* dist, int64_t ldo);

/* a8 pooling step: allDesc = allDesc ./ sqrt(sum(allDesc.^2, 2) + eps('single')) (featureMatchingGlobal.m:80-86; eps inside
 * the root, unlike matchFeaturesScratch's normalizeRowsL2).  out: n x dim f32 row-major (ld = dim), host or device. */
int aps_global_normalize(const float* X, int64_t n, int64_t ld, int dim, int layout, float* out);

/* a8 filter: the per-query loop of featureMatchingGlobal.m:123-161 (drop self, drop same-image,
 * need >= 2 left, reject iff d1/max(d2,eps('single')) > ratio, append [li lj] to pair (min,max) in
 * query order).  img_idx (1-based image id per row) and local_idx (1-based) are uint32[f].
 * Output CSR over pairs in the same pair order as aps_match_pairwise. */
int aps_global_filter(const uint32_t* nn_idx, const float* nn_dist, int64_t f, int k,
                      int64_t ldn, int layout, const uint32_t* img_idx, const uint32_t* local_idx,
                      int n_img, float ratio, int64_t* pair_ptr, uint32_t* idx_i, uint32_t* idx_j,
                      int64_t cap, int64_t* count);

/* a9: [idx2,d1,d2] = nearest2HammingExhaustiveMEX(Abytes,Bbytes)
 * (nearest2HammingExhaustiveMEX.cpp:16-80, ...OMPMEX.cpp:18-83): brute-force Hamming 2-NN on packed
 * bytes with the reference's tie rule (strict < for best, <= for second, :63-68), N2==0 -> idx 0 and
 * NaN (:42-45), single candidate -> second = nb*8 (:71-74). */
int aps_hamming_2nn(const uint8_t* A, int64_t n1, int64_t lda, const uint8_t* B, int64_t n2,
                    int64_t ldb, int nbytes, int layout, uint32_t* idx2, float* d1, float* d2);

/* ============================================================================================
 * (2) Geometric verification — PP/imageMatching/estimateTransformationRANSAC.m
 * ============================================================================================ */

/* input.transformationType (inputs.m:74) = transformType of estimateTransformationRANSAC.m:612-660; minimal samples
 * 4 / 3 / 2 / 2 / 1.  All five run through APS_ROBUST_RANSAC and through APS_ROBUST_MLESAC (whose estimators differ:
 * estimateTransformationMLESAC.m:345-510 - null vectors of the 2n x 9 / 7 / 5 systems, Kabsch for 'rigid', the mean
 * displacement for 'translation'). */
enum {
    APS_TFORM_PROJECTIVE = 0,
    APS_TFORM_AFFINE = 1,      /* estimateAffine :227-288: pseudo-inverse of the normalised design matrix        */
    APS_TFORM_SIMILARITY = 2,  /* estimateSimilarity :290-356: rotation from the 2x2 cross-covariance, median scale */
    APS_TFORM_RIGID = 3,       /* estimateRigid :358-421 (identity rotation when the cross-covariance is ill-conditioned) */
    APS_TFORM_TRANSLATION = 4  /* estimateTranslation :423-452: per-axis median displacement                      */
};

/* a12 findInliers (estimateTransformationRANSAC.m:444-516) for T hypotheses at once.
 *   Hs    : f64 3x3xT, each 3x3 column-major (MATLAB page layout)
 *   p1,p2 : f64 Mx2 column-major with leading dimension ldp (x column then y column)
 *   n_inl : int32[T]; mean_err: f64[T] (mean error over inliers, NaN if none);
 *   mask  : uint8 MxT column-major (may be NULL).
 * Projective: e = sqrt(|x2-Hx1|^2 + |x1-H^-1 x2|^2) < thr (:474-481); non-finite or |w|<eps -> inf
 * (:499-503); >=4 inliers whose centred x1 have s2/s1 < 1e-3 -> all false (:506-513,:567).
 * Affine / similarity / rigid: e = |x2 - Hx1| (:483-484); translation: the same error and the threshold both divided by
 * max(|coordinates|, 1) of the whole point set (:486-494); the collinearity test applies to affine (>= 3 inliers) only. */
int aps_ransac_score(const double* Hs, int n_hyp, const double* p1, const double* p2, int64_t m,
                     int64_t ldp, double thr, int tform_type, int32_t* n_inl, double* mean_err,
                     uint8_t* mask);

enum { APS_ROBUST_RANSAC = 0, APS_ROBUST_MLESAC = 1 };

typedef struct aps_ransac_opts {
    double max_distance; /* input.maxDistance        (inputs.m:69: 5.5)  */
    double confidence;   /* input.inliersConfidence  (inputs.m:72: 99.9) */
    int max_iter;        /* input.maxIter            (inputs.m:68: 500)  */
    int tform_type;      /* APS_TFORM_* (input.transformationType, inputs.m:74: 'projective')  */
    int method;          /* APS_ROBUST_RANSAC: estimateTransformationRANSAC.m (input.imageMatchingMethod 'ransac');
                            APS_ROBUST_MLESAC: estimateTransformationMLESAC.m:94-254 ('mlesac'): one-way distance,
                            truncated-loss score, refit on the inliers is the answer                */
} aps_ransac_opts;

/* a12 whole loop: [model, inliers, isFound] = estimateTransformationRANSAC(p1, p2, transformType, input)
 * (estimateTransformationRANSAC.m:54-183; the name dates from the projective-only rounds, opts->tform_type selects the
 * model).  The random subsets are an INPUT for every type: the first minPoints entries of a 4-column draw are the
 * sample.  sample_idx is
 * uint32 4 x n_samples column-major, 1-based, one column per loop iteration (each iteration of
 * :94-143 consumes one randperm draw, skipped ones included).  All hypotheses are fitted and scored
 * on the device in one batch; the data-dependent best/early-exit logic (:115-130) is replayed on the
 * host in order, so the result equals the sequential loop on the same draws.
 * model: f64 3x3 column-major; inlier_mask: uint8[m]; is_found: 0/1; trials_used: loop iterations
 * the sequential algorithm would have executed (may be NULL). */
int aps_ransac_homography(const double* p1, const double* p2, int64_t m, int64_t ldp,
                          const uint32_t* sample_idx, int n_samples, const aps_ransac_opts* opts,
                          double* model, uint8_t* inlier_mask, int* is_found, int* trials_used);

/* Number of pairs in the calling thread's most recent aps_ransac_homography / aps_ransac_homography_batch call whose
 * n_samples pre-drawn subsets ran out BEFORE the sequential loop's own stopping rule (trial > maxTrials, or 10*maxIter
 * skipped draws, estimateTransformationRANSAC.m:94): the reference would have kept drawing, so for those pairs the result
 * is that of a truncated loop.  Callers that want the reference's behaviour re-run with more draws (the counter-based
 * draw stream of aps_ransac_draw_samples only gets longer: earlier draws do not change). */
int aps_ransac_draws_exhausted(void);

/* The seeded stand-in for randperm(numPoints, 4) (estimateTransformationRANSAC.m:96): fills sample_idx
 * (uint32 4 x n_samples x n_pairs, 1-based) with distinct 4-subsets of 1..counts[p] from the counter-based
 * stream u = mix64(seed, keys[p] (or p if keys is NULL), 4*iteration + k) — identical to
 * imageMatching.draw_samples on the host.  A pair with counts[p] < 4 gets counts[p] distinct entries followed by ones
 * (enough for the transformTypes whose minimal sample it can still serve). */
int aps_ransac_draw_samples(const int64_t* counts, const uint64_t* keys, int n_pairs, int n_samples,
                            uint64_t seed, uint32_t* sample_idx);

/* The input of that batch, gathered on the device (imageMatching.m:121-135: keypoints{i}(matches(:,1),:) and
 * keypoints{j}(matches(:,2),:) per candidate pair): work pair q is images (img_a[q], img_b[q]); its matches are the
 * work_ptr[q+1]-work_ptr[q] entries of the resident 1-based lists idx_a / idx_b starting at list_start[q].
 *   kp       : host array of n_img DEVICE pointers, image i's keypoints as kp_count[i] x 2 f64 row-major [x y]
 *   idx_a/b  : device int32 lists (what aps_match_pairs leaves on the device)
 *   list_start, work_ptr (n_work+1), img_a, img_b: host arrays
 *   pts_a/b  : device f64, (total x 2) column-major with leading dimension ldp >= total = work_ptr[n_work];
 *              an index outside its image's table yields NaN coordinates (such a pair then fails RANSAC). */
int aps_gather_match_points(const double* const* kp, const int64_t* kp_count, int n_img, const int32_t* idx_a,
                            const int32_t* idx_b, const int64_t* list_start, const int64_t* work_ptr,
                            const int32_t* img_a, const int32_t* img_b, int n_work, double* pts_a, double* pts_b,
                            int64_t ldp);

/* Batched a10/a11/a12: every candidate pair of imageMatching.m:121-156 in one device batch.
 *   pts1/pts2 : f64, pair p's matched points are rows pair_ptr[p]..pair_ptr[p+1]-1 of two
 *               (total x 2) column-major arrays with leading dimension ldp
 *   sample_idx: uint32 4 x n_samples x n_pairs (1-based, local to the pair)
 * Outputs per pair: models f64 3x3xP, mask uint8[total], found int32[P], n_inl int32[P]. */
int aps_ransac_homography_batch(const double* pts1, const double* pts2, int64_t ldp,
                                const int64_t* pair_ptr, int n_pairs, const uint32_t* sample_idx,
                                int n_samples, const aps_ransac_opts* opts, double* models,
                                uint8_t* mask, int32_t* found, int32_t* n_inl);

/* ============================================================================================
 * (3) Rendering — PP/renderPanorama/renderPanorama.m, PP/blending/ (both .m files), PP/imageProcessing/imageWarp.m
 * ============================================================================================ */

enum {
    APS_PROJ_CYLINDRICAL = 0,
    APS_PROJ_SPHERICAL = 1, /* 'equirectangular' is an alias (renderPanorama.m:180-189,357) */
    APS_PROJ_PLANAR = 2,
    APS_PROJ_STEREOGRAPHIC = 3
};
enum { APS_BLEND_NONE = 0, APS_BLEND_LINEAR = 1, APS_BLEND_MULTIBAND = 2 };
enum { APS_NONE_LAST = 0, APS_NONE_FIRST = 1, APS_NONE_MAXANGLE = 2 };
enum {
    APS_IMG_U8_HWC = 0,   /* row-major interleaved H x W x C (numpy / torch)            */
    APS_IMG_U8_MATLAB = 1 /* column-major planar H x W x C (MATLAB uint8 array)        */
};

/* One source image + its camera (cameras struct: initializeCameraMatrices.m:114-122). */
typedef struct aps_image {
    const uint8_t* data; /* uint8 pixels, host or device                                        */
    int height, width, channels; /* channels: 1 or 3 (gray is replicated, loadImages.m:62)    */
    int layout;          /* APS_IMG_U8_*                                                        */
    double K[9];         /* 3x3 intrinsics, column-major                                        */
    double R[9];         /* 3x3 world->camera rotation, column-major                            */
    float gain[3];       /* per-channel gain (gainCompensationRKf output row; ones if off)      */
} aps_image;

/* Geometry of the panorama canvas: the values renderPanorama.m:125-232 derives on the host. */
typedef struct aps_canvas {
    int mode;            /* APS_PROJ_*                                                          */
    int height, width;   /* H, W (:137-146,180-189)                                             */
    double f_pan;        /* opts.fPan * opts.resScale enters only through these three:          */
    double origin0;      /* th0 (cyl/sph) or u0 (planar/stereo)                                 */
    double origin1;      /* h0 (cyl), ph0 (sph) or v0 (planar/stereo)                           */
    double R_ref[9];     /* cameras(refIdx).R, column-major (planar/stereographic only)         */
} aps_canvas;

typedef struct aps_render_opts {
    int tile_h, tile_w;    /* opts.tile — explicit, never derived from free memory */
    float angle_power;     /* opts.anglePower (displayPanorama.m:101: 2)                        */
    int blending;          /* APS_BLEND_*                                                       */
    int pyr_levels;        /* opts.pyrLevels = input.bands                                      */
    float pyr_sigma;       /* opts.pyrSigma  = input.MBBsigma                                   */
    int none_policy;       /* opts.composeNonePolicy                                            */
    int canvas_white;      /* opts.canvasColor == 'white'                                       */
} aps_render_opts;

/* a14-a17: the tile loop of renderPanorama.m:342-425 (ray generation, fuseTile, sampleOneTile,
 * sampleBlock, warpWeights, void paint, uint8 conversion), all tiles, on the device.
 *   pano    : uint8 H x W x 3 in `out_layout` (APS_IMG_U8_HWC or APS_IMG_U8_MATLAB)
 *   covered : uint8 H x W (same 2-D layout), may be NULL. */
int aps_render(const aps_image* images, int n_img, const aps_canvas* canvas,
               const aps_render_opts* opts, int out_layout, uint8_t* pano, uint8_t* covered);

/* The same, restricted to the tiles t (row-major tile index over the canvas) with t % tile_step ==
 * tile_first: tiles are independent in the reference (renderPanorama.m:342-406; pyramids are tile
 * local), so this is the unit of multi-GPU sharding.  Pixels of other tiles are left untouched. */
int aps_render_tiles(const aps_image* images, int n_img, const aps_canvas* canvas,
                     const aps_render_opts* opts, int out_layout, int tile_first, int tile_step,
                     uint8_t* pano, uint8_t* covered);

/* The same for the CONTIGUOUS tiles tile_begin <= t < tile_end of the row-major tile list: a rank that owns a run of
 * neighbouring tiles meets (and converts) only the ~20 of 64 views under its band of the canvas, where tiles dealt
 * t % p meet 47 (round 5; section 6 of DESIGN.md). */
int aps_render_tile_range(const aps_image* images, int n_img, const aps_canvas* canvas, const aps_render_opts* opts, int out_layout,
                          int tile_begin, int tile_end, uint8_t* pano, uint8_t* covered);

/* SURVEY 8(f) rank 2 -- imresize(I, s | [oh ow], 'bicubic' | 'bilinear') on a uint8 image, the preprocessing step in
 * front of SIFT (PP/imageProcessing/resizeImagesToLimits.m:57-61,103; loadImages.m:66-68).  Antialiased on shrink,
 * half-pixel centres, replicate borders, the smaller-scale dimension first, uint8 rounding after each pass.
 * scale_r/scale_c: the per-dimension scales imresize uses (s, s for the scalar form; oh/h, ow/w for the size form). */
enum { APS_RESIZE_BILINEAR = 0, APS_RESIZE_BICUBIC = 1 };
int aps_imresize_u8(const uint8_t* img, int h, int w, int c, int layout, int oh, int ow, double scale_r, double scale_c,
                    int method, uint8_t* out);

/* SURVEY 8(f) rank 3 -- the per-pair blocks of the bundle adjustment's normal equations: the parfor body of
 * accumulateNormalEqnsBlock (PP/bundleAdjustment/bundleAdjustmentRKf.m:717-741) with jacobianPair (:793-899),
 * computeSingleResidual (:1641-1686), computeJacobianWrtCamera (:1688-1783) and huberWeight (:1806-1829).
 * Ui, Uj: total x 2 f64, column-major with leading dimension ldu (x at [k], y at [ldu + k]): the matched points on image
 * i and j of all pairs back to back; pair p owns rows pair_ptr[p] .. pair_ptr[p+1]-1.  cams: per pair four cameras
 * (base i, base j, incremented i, incremented j), 12 f64 each: f, cx, cy, R (3 x 3 column-major, world -> camera).
 * out: per pair 59 f64 = Hii, Hjj, Hij (4 x 4 column-major over [dthx dthy dthz df]), gi, gj (4), E, r2sum, rcnt.
 * A camera with fewer parameters uses the leading rows/columns, as the reference's J(:, 1:numel(cols)) does.
 * both_directions = !opts.OneDirection.  The LM loop, prior, sparse assembly and solve stay with the caller. */
int aps_ba_pair_blocks(const double* Ui, const double* Uj, int64_t ldu, const int64_t* pair_ptr, int n_pairs,
                       const double* cams, double sigma_huber, int both_directions, double* out);

/* SURVEY 8(f) rank 4 -- the crop rectangle of PP/imageProcessing/panoramaCropper.m:73-165: rgb2gray + imbinarize against
 * `range` (input.blackRange, or input.whiteRange with canvas_white = 1 and the mask complemented), imfill(.,'holes'), and
 * the line-by-line largest-rectangle scan (first maximum in (line, column) order, the last column never part of a
 * rectangle, as in the reference).  img: h x w x 3 uint8 (layout as for the renderers).  rect[0..3] = offsetx, offsety,
 * cropW, cropH, 1-based as in :153-157; the reference then takes rows offsety..offsety+cropH and columns
 * offsetx..offsetx+cropW.  *valid = 0 when that range leaves the image (the reference warns and returns the input). */
int aps_crop_rect(const uint8_t* img, int64_t h, int64_t w, int layout, int canvas_white, double range, int32_t* rect,
                  int32_t* valid);

/* a20: [panoCropped, rect, didCrop] = cropNonzeroBbox(panorama, canvasColor) (renderPanorama.m:1459-1504), the step
 * renderPanorama runs on its output when opts.cropBorder is set (:430-432; displayPanorama.m:101 sets it): bounding box of
 * rgb2gray(panorama) > 0 (black canvas) or < 255 (white canvas), padded by 6 pixels and clipped to the image.
 * img: h x w x 3 uint8 (layout as for the renderers, host or device).  rect = {r1, r2, c1, c2}, 1-based inclusive;
 * *did_crop = 0 and rect = the whole image when there is no foreground.  The caller slices (no pixels are moved here). */
int aps_crop_nonzero_bbox(const uint8_t* img, int64_t h, int64_t w, int layout, int canvas_white, int64_t* rect,
                          int* did_crop);

/* SURVEY 8(f) rank 1 -- the overlap statistics of gainCompensationRKf (PP/gainCompensation/gainCompensationRKf.m:96-149,
 * 239-367): every `stride`-th canvas point (1-based coordinates, :106-107) that two images i < j both cover
 * (front, inside, tent weight > 0) adds 1 to n_ij(i,j) and the two bilinear RAW (0..255) colour samples to
 * sum_ci(i,j,:) / sum_cj(i,j,:).  Outputs: f64 N x N and N x N x 3, column-major, upper triangle, zero elsewhere.
 * The N x N solve for the gains (:151-238) stays on the host.  Sums are added in an unspecified order: compare
 * them with a relative tolerance (the reference itself sums in single per tile); counts are exact. */
int aps_gain_overlap_stats(const aps_image* images, int n_img, const aps_canvas* canvas, int stride,
                           double* n_ij, double* sum_ci, double* sum_cj);

/* SURVEY 8(f) rank 1, planar scans -- the overlap statistics of gainCompensationH
 * (PP/gainCompensation/gainCompensationH.m:45-52,78-149; caller renderPanorama.m:584-588).  iw[k]: the k-th image warped to
 * the common canvas, f32 height x width x channels; ww[k]: its weight map, f32 height x width (host or device memory, like
 * every other buffer of this library; layout APS_ROWMAJOR = interleaved rows as C / numpy hold them, APS_COLMAJOR = MATLAB's
 * planar column-major arrays).  Every `downsample`-th row and column of the canvas is sampled (1:ds:end, :45-52); a sample
 * is valid for image k when ww[k] > 0 and all channels are finite (:117); each pair i < j valid there adds 1 to n_ij(i,j)
 * and its two colours to sum_ci(i,j,:) / sum_cj(i,j,:), accumulated in double (:126-146).  Outputs as aps_gain_overlap_stats:
 * f64 N x N and N x N x 3, column-major, upper triangle.  Sums are added in an unspecified order (compare with a relative
 * tolerance); counts are exact.  The edge selection and the N x N solve (:152-223) stay on the host. */
int aps_gain_overlap_stats_warped(const float* const* iw, const float* const* ww, int n_img, int64_t height, int64_t width,
                                  int channels, int layout, int downsample, double* n_ij, double* sum_ci, double* sum_cj);

/* a15/a16 for ONE tile, layers out (for tests): rows r0..r0+ht-1, cols c0..c0+wt-1 (0-based) of
 * the canvas sampled from ONE image.  S: f32 ht x wt x 3 row-major interleaved, Wang/Wf: f32 ht x wt,
 * M: uint8 ht x wt (sampleOneTile, renderPanorama.m:1063-1146). */
int aps_warp_tile(const aps_image* image, const aps_canvas* canvas, int r0, int c0, int ht, int wt,
                  float angle_power, float* S, uint8_t* M, float* Wang, float* Wf);

/* a21: F = multiBandBlending(Ci, Wi, levels, onGPU, sigma) (multiBandBlending.m:45-171).
 *   C: f32 K x h x w x 3 (row-major interleaved per layer), Wt: f32 K x h x w, F: f32 h x w x 3. */
int aps_multiband_blend(const float* C, const float* Wt, int k, int h, int w, int levels,
                        float sigma, float* F);

/* a22: linearBlending (linearBlending.m:46-115) on f32 layers: sum(I.*W)/max(sum(W),eps('single')). */
int aps_linear_blend(const float* C, const float* Wt, int k, int h, int w, float* F);

/* a19: warped = imageWarp(image, tform, outputView, 'bilinear') (imageWarp.m:39-168) for uint8 or
 * f32 images: inverse homography warp onto an imref2d-like grid, valid only when all four taps are
 * inside (:133), fill value elsewhere.
 *   H: f64 3x3 column-major; x0,y0,sx,sy: outputView.XWorldLimits(1), YWorldLimits(1),
 *   PixelExtentInWorldX/Y (:36-41).  in/out: row-major interleaved h x w x c. */
int aps_image_warp_h_u8(const uint8_t* in, int in_h, int in_w, int c, const double* H, int out_h,
                        int out_w, double x0, double y0, double sx, double sy, uint8_t fill,
                        uint8_t* out);
int aps_image_warp_h_f32(const float* in, int in_h, int in_w, int c, const double* H, int out_h,
                         int out_w, double x0, double y0, double sx, double sy, float fill,
                         float* out);
/* The same with options.method: 'nearest' (imageWarp.m:109-123: round(src), valid inside [1,w] x [1,h]), 'bilinear'
 * (:125-168) or 'bicubic' (:170-264: Keys kernel bicubicKernel :275-301 on the 4 x 4 taps around floor(src), valid for
 * 2 <= floor(src) <= size - 2, x direction first, result clamped to [0, 255] and rounded for uint8, to [0, 1] for f32). */
enum { APS_WARP_NEAREST = 0, APS_WARP_BILINEAR = 1, APS_WARP_BICUBIC = 2 };
int aps_image_warp_u8(const uint8_t* in, int in_h, int in_w, int c, const double* H, int out_h, int out_w, double x0,
                      double y0, double sx, double sy, uint8_t fill, int method, uint8_t* out);
int aps_image_warp_f32(const float* in, int in_h, int in_w, int c, const double* H, int out_h, int out_w, double x0,
                       double y0, double sx, double sy, float fill, int method, float* out);

/* ============================================================================================
 * (4) SIFT — PP/featureMatching/getFeaturePoints.m:36-40,71-74
 * ============================================================================================ */

typedef struct aps_sift_params {
    double sigma;              /* input.Sigma              (inputs.m:34: 1.6)     */
    int n_layers;              /* input.NumLayersInOctave  (inputs.m:35: 4)       */
    double contrast_threshold; /* input.ContrastThreshold  (inputs.m:36: 0.00133) */
    double edge_threshold;     /* input.EdgeThreshold      (inputs.m:40: 6)       */
    int max_features;          /* capacity guard; 0 = library default (262144)    */
} aps_sift_params;

/* a1: [features, validPts] = getFeaturePoints(input, img) for detector 'SIFT': rgb2gray, Lowe/OpenCV
 * SIFT (detectSIFTFeatures + extractFeatures are closed toolbox code; the algorithm restated is
 * OpenCV's cv::SIFT, which the toolbox is documented to wrap — see DESIGN.md "SIFT contract").
 *   desc : f32 count x 128, `desc_layout` with leading dimension ldd, unit L2 norm
 *   loc  : f64 count x 2 [x y], 1-based sub-pixel, column-major with leading dimension ldl
 *   aux  : f32 count x 4 row-major [scale(size), angle_deg, response, octave_layer] or NULL
 * cap = rows available in desc/loc/aux; *count = features found (APS_E_CAP if cap is too small).
 * Feature order is canonical: ascending (octave, layer, row, col, orientation bin). */
int aps_sift_extract(const uint8_t* img, int height, int width, int channels, int img_layout,
                     const aps_sift_params* params, float* desc, int desc_layout, int64_t ldd,
                     double* loc, int64_t ldl, float* aux, int64_t cap, int64_t* count);

/* ============================================================================================
 * Bench / test support — NOT part of the reference boundary
 * ============================================================================================ */
/* One uint8 H x W x 3 (row-major interleaved) view of the seeded procedural world used by bench.py and the
 * pipeline tests, through the pinhole camera (K, R), both 3x3 ROW-major here.  See synth.py. */
int aps_synth_view(const double* K_rowmajor, const double* R_rowmajor, int height, int width,
                   unsigned seed, float finest_px, float gain, uint8_t* out);

#ifdef __cplusplus
}
#endif
#endif /* APS_H_ */
