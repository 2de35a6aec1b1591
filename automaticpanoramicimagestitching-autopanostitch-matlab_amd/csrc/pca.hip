// pca.hip - a6: nearest2ApproxFloatFast / doBlock (PP/featureMatching/matchFeaturesScratch.m:442-573) on the device.
//
//   muB = mean(B, 1, 'omitnan')                                   (:480)
//   coeff = pca(B - muB, 'NumComponents', k)                      (:481-482, toolbox: principal axes of the covariance by
//                                                                  descending variance, each column signed so that its
//                                                                  largest-magnitude entry is positive)
//   B = (B - muB) * coeff;  A = (A - muB) * coeff                 (:483-484)
//   rows / (sqrt(sum(rows.^2, 2)) + eps('single'))                (:488-489)
//   G = Ablk * Bfull.';  [sim1, id1] = max(G, [], 2);  G(id1) = -inf;  sim2 = max(G, [], 2);  d = 2 - 2 * sim   (:552-570)
//
// Everything but the 128 x 128 eigen-decomposition runs in kernels; the arithmetic order is fixed (the contract stated
// in include/aps.h at aps_match_pca2nn, restated by oracle/pca_oracle.c):
//   * sums over rows (mean, covariance) are taken per chunk of 256 rows in ascending row order - the covariance chunk as
//     the f32 fma chain v_mfma_f32_32x32x2_f32 computes - and the chunk partials are added in f64 in ascending chunk order;
//   * the projection is the k-ascending f32 fma chain per (row, component), the squared norm s = s + y*y over ascending
//     components, the cosine product the component-ascending f32 fma chain (again one MFMA chain);
//   * the symmetric eigen-problem is solved on the host in f64 by cyclic Jacobi rotations in a fixed order.
#include "aps_internal.h"

#include <algorithm>
#include <cmath>
#include <numeric>

namespace aps {

namespace {
constexpr int kD = 128;       // descriptor length (SIFT)
constexpr int kChunk = 256;   // rows per partial sum: the canonical reduction tree
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float elem(const float* __restrict__ X, int64_t ld, int layout, int64_t r, int c) {
    return layout == APS_ROWMAJOR ? X[r * ld + c] : X[(int64_t)c * ld + r];
}

// ---- (1) column sums of a chunk: thread = column, rows in ascending order, f64, NaN skipped ('omitnan') -----------------
__global__ __launch_bounds__(kD) void pca_colsum_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout,
                                                        double* __restrict__ part_sum, int* __restrict__ part_cnt) {
    const int c = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kChunk, r1 = min(r0 + (int64_t)kChunk, n);
    double s = 0.0;
    int cnt = 0;
    for (int64_t r = r0; r < r1; ++r) {
        const float v = elem(X, ld, layout, r, c);
        if (v == v) {
            s += (double)v;
            ++cnt;
        }
    }
    part_sum[(size_t)blockIdx.x * kD + c] = s;
    part_cnt[(size_t)blockIdx.x * kD + c] = cnt;
}

__global__ __launch_bounds__(kD) void pca_mean_kernel(const double* __restrict__ part_sum, const int* __restrict__ part_cnt, int n_chunks,
                                                      float* __restrict__ mu) {
    const int c = threadIdx.x;
    double s = 0.0;
    long long cnt = 0;
    for (int b = 0; b < n_chunks; ++b) {
        s += part_sum[(size_t)b * kD + c];
        cnt += part_cnt[(size_t)b * kD + c];
    }
    mu[c] = (float)(s / (double)cnt);  // (a column of NaNs only: 0 / 0 = NaN, as mean(..., 'omitnan') gives)
}

// ---- (2) covariance of a chunk: C(i, j) = sum_r (x(r,i) - mu(i)) * (x(r,j) - mu(j)), rows ascending, one f32 fma chain ----
// 256 threads: wave w owns the 32 x 128 strip of rows i = 32 w .. 32 w + 31 as four 32 x 32 MFMA tiles.  MFMA operand maps
// (v_mfma_f32_32x32x2_f32): A: lane l holds A[i = l & 31][k = l >> 5], B: lane l holds B[k = l >> 5][j = l & 31]; the two
// k of one instruction are two consecutive rows of the chunk, accumulated in that order.
__global__ __launch_bounds__(256) void pca_cov_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout,
                                                      const float* __restrict__ mu, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kChunk;
    float m[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) m[t] = mu[32 * t + c];
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int kk = 0; kk < kChunk / 2; ++kk) {
        const int64_t r = r0 + 2 * kk + h;
        float b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = r < n ? __fsub_rn(elem(X, ld, layout, r, 32 * t + c), m[t]) : 0.f;  // rows past the end add +0
        float a = b[0];
        a = wave == 1 ? b[1] : a;
        a = wave == 2 ? b[2] : a;
        a = wave == 3 ? b[3] : a;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
    }
    float* out = part + (size_t)blockIdx.x * kD * kD;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = 32 * wave + (e & 3) + 8 * (e >> 2) + 4 * h;  // C/D map: row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
            out[(size_t)i * kD + 32 * t + c] = acc[t][e];
        }
}

__global__ __launch_bounds__(256) void pca_cov_fold_kernel(const float* __restrict__ part, int n_chunks, double inv_nm1, double* __restrict__ cov) {
    const int e = blockIdx.x * 256 + threadIdx.x;  // 16384 entries
    double s = 0.0;
    for (int b = 0; b < n_chunks; ++b) s += (double)part[(size_t)b * kD * kD + e];
    cov[e] = s * inv_nm1;
}

// ---- (3) projection + L2 normalisation: one thread per row ------------------------------------------------------------------
// 64 rows per workgroup: the centred rows are staged in LDS (pitch 129: conflict-free column walks), the coefficients too
// (read as broadcasts).  y(c) = fma chain over k = 0 .. 127 of (x(k) - mu(k)) * coeff(k, c); s = s + y*y over ascending c;
// y / (sqrt(s) + eps).  Output row: KP floats in the MFMA order of cos2nn_kernel: position h * KH + s holds component
// 2 s + h (h = 0, 1; KH = KP / 2), components >= ncomp are zero.  ncomp == 0: no projection (UsePCA = false or D <= k):
// the row itself, D = 128 components.
__global__ __launch_bounds__(64) void pca_project_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout,
                                                         const float* __restrict__ mu, const float* __restrict__ coeff, int ncomp,
                                                         int KP, float* __restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_x = smem;                       // [64][129]
    float* s_c = smem + 64 * (kD + 1);       // [128][ncomp]
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const bool project = ncomp > 0;
    for (int e = tid; e < 64 * kD; e += 64) {
        const int rr = layout == APS_ROWMAJOR ? e / kD : e % 64, cc = layout == APS_ROWMAJOR ? e % kD : e / 64;
        const int64_t r = min(r0 + rr, n - 1);
        const float v = elem(X, ld, layout, r, cc);
        s_x[rr * (kD + 1) + cc] = project ? __fsub_rn(v, mu[cc]) : v;
    }
    if (project)
        for (int e = tid; e < kD * ncomp; e += 64) s_c[e] = coeff[e];  // coeff is [k][c] (row-major 128 x ncomp)
    __syncthreads();
    const int64_t r = r0 + tid;
    const float* xr = &s_x[tid * (kD + 1)];
    const int nc = project ? ncomp : kD, KH = KP / 2;
    float* yrow = Y + (size_t)min(r, n - 1) * KP;
    const bool live = r < n;
    float s = 0.f;
    // pass 1: the components, 16 at a time (registers), parked in the output row in canonical order first
    for (int c0 = 0; c0 < nc; c0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        if (project) {
            for (int k = 0; k < kD; ++k) {
                const float xk = xr[k];
                const float* ck = &s_c[k * ncomp + c0];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (c0 + j < nc) acc[j] = fmaf(xk, ck[j], acc[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = xr[c0 + j];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (c0 + j < nc) {
                s = __fadd_rn(s, __fmul_rn(acc[j], acc[j]));
                const int comp = c0 + j;
                if (live) yrow[(comp & 1) * KH + (comp >> 1)] = acc[j];
            }
    }
    if (!live) return;
    const float nrm = __fadd_rn(sqrtf(s), 1.1920929e-07f);  // sqrt(sum(x.^2, 2)) + eps('single')
    for (int comp = 0; comp < 2 * KH; ++comp) {
        const int pos = (comp & 1) * KH + (comp >> 1);
        yrow[pos] = comp < nc ? __fdiv_rn(yrow[pos], nrm) : 0.f;
    }
}

// ---- (4) the two largest cosine similarities per A row -------------------------------------------------------------------------
// Wave = 32 A rows held as the MFMA's B operand (lane l: A row j = l & 31, components 2 s + (l >> 5)); the B set streams by in
// tiles of 32 rows as the A operand, so a result register holds sim(B row i = 32 t + (reg & 3) + 8 (reg >> 2) + 4 h, A row j):
// each lane reduces ITS A row over 16 B rows per tile, in ascending B order, with max's first-index rule (a later equal
// value does not replace the best, :558 `max` returns the first maximum; it does become the second, :559-560).  grid.y splits
// the B tiles; cos2nn_merge_kernel combines the splits and the two lane halves.
struct Top2 {
    float best, second;
    unsigned int idx;
};
__device__ __forceinline__ void top2_merge(Top2& a, const Top2& b) {  // a <- the top two of the union (ties: lower index first)
    const bool b_wins = b.best > a.best || (b.best == a.best && b.idx < a.idx);
    const float sec = b_wins ? fmaxf(b.second, a.best) : fmaxf(a.second, b.best);
    a.best = b_wins ? b.best : a.best;
    a.idx = b_wins ? b.idx : a.idx;
    a.second = sec;
}

template <int KH>  // KP / 2: 24 for ApproxNumComponents = 48
__global__ __launch_bounds__(256) void cos2nn_kernel(const float* __restrict__ YA, int n1, const float* __restrict__ YB, int n2,
                                                     int tiles_per_split, Top2* __restrict__ part) {
    constexpr int KP = 2 * KH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
    const int arow = blockIdx.x * 128 + wave * 32 + c;
    f32x4 av[KH / 4];
    {
        const f32x4* ap = reinterpret_cast<const f32x4*>(YA + (size_t)min(arow, n1 - 1) * KP + h * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) av[q] = ap[q];
    }
    Top2 run{-INFINITY, -INFINITY, 0xffffffffu};
    const int n_tiles = (n2 + 31) / 32;
    const int t0 = blockIdx.y * tiles_per_split, t1 = min(t0 + tiles_per_split, n_tiles);
    f32x4 cur[KH / 4], nxt[KH / 4];
    auto load = [&](int t, f32x4* dst) {
        const f32x4* bp = reinterpret_cast<const f32x4*>(YB + (size_t)min(32 * t + c, n2 - 1) * KP + h * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) dst[q] = bp[q];
    };
    if (t0 < t1) load(t0, cur);
    for (int t = t0; t < t1; ++t) {
        if (t + 1 < t1) load(t + 1, nxt);
        f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].x, av[q].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].y, av[q].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].z, av[q].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[q].w, av[q].w, acc, 0, 0, 0);
        }
        const int ib = 32 * t + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = ib + (e & 3) + 8 * (e >> 2);
            const float v = i < n2 ? acc[e] : -INFINITY;
            const bool gt = v > run.best;
            run.second = gt ? run.best : fmaxf(run.second, v);  // (NaN similarities are ignored, as max does)
            run.idx = gt ? (unsigned)i : run.idx;
            run.best = gt ? v : run.best;
        }
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) cur[q] = nxt[q];
    }
    if (arow < n1) part[((size_t)blockIdx.y * 2 + h) * n1 + arow] = run;
}

__global__ void cos2nn_merge_kernel(const Top2* __restrict__ part, int n1, int n_parts, uint32_t* __restrict__ idx2, float* __restrict__ d1,
                                    float* __restrict__ d2) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n1) return;
    Top2 a = part[r];
    for (int p = 1; p < n_parts; ++p) top2_merge(a, part[(size_t)p * n1 + r]);
    idx2[r] = a.idx == 0xffffffffu ? 1u : a.idx + 1u;  // (a row of NaN similarities: max's index 1)
    d1[r] = __fsub_rn(2.0f, __fmul_rn(2.0f, a.best));
    d2[r] = __fsub_rn(2.0f, __fmul_rn(2.0f, a.second));
}

// ---- host: symmetric eigen-decomposition by cyclic Jacobi rotations (f64, fixed order) ---------------------------------------
// The rotation order (p ascending, q ascending), the rotation formulas and the stopping rule are part of the contract: the
// oracle performs the same operations, so the coefficient bits agree.  A: n x n symmetric, destroyed (its diagonal ends as
// the eigenvalues); V: eigenvectors in columns.
void jacobi_eigh(std::vector<double>& A, int n, std::vector<double>& V) {
    V.assign((size_t)n * n, 0.0);
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
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; ++k) {  // columns p, q
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = cs * akp - sn * akq;
                    A[(size_t)k * n + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < n; ++k) {  // rows p, q
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

// principal axes of cov (128 x 128, f64) by descending eigenvalue (ties: lower Jacobi column first), each signed so that
// its largest-magnitude entry (first on ties) is positive; coeff[k * ncomp + c], f32.  Only the first `keep` axes are filled:
// pca() of n observations returns min(n - 1, NumComponents) columns (the centred data has rank <= n - 1), and an axis of zeros
// projects every row of either set to exactly 0 - the same sums, norms and cosines as a basis without that column.
void pca_axes(std::vector<double>& cov, int ncomp, int keep, std::vector<float>& coeff) {
    std::vector<double> V;
    jacobi_eigh(cov, kD, V);
    std::vector<int> order(kD);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cov[(size_t)a * kD + a] > cov[(size_t)b * kD + b]; });
    coeff.assign((size_t)kD * ncomp, 0.f);
    for (int c = 0; c < std::min(ncomp, keep); ++c) {
        const int col = order[c];
        int big = 0;
        for (int k = 1; k < kD; ++k)
            if (std::fabs(V[(size_t)k * kD + col]) > std::fabs(V[(size_t)big * kD + col])) big = k;
        const double sgn = V[(size_t)big * kD + col] < 0.0 ? -1.0 : 1.0;
        for (int k = 0; k < kD; ++k) coeff[(size_t)k * ncomp + c] = (float)(sgn * V[(size_t)k * kD + col]);
    }
}
}  // namespace

}  // namespace aps

extern "C" int aps_match_pca2nn(const float* A, int64_t n1, int64_t lda, const float* B, int64_t n2, int64_t ldb, int dim, int layout,
                                int n_components, int use_pca, uint32_t* idx2, float* d1, float* d2, float* mu_out, float* coeff_out) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(dim == kD, APS_E_DIM, "descriptor length %d not supported (built for %d-D SIFT)", dim, kD);
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(n1 >= 1 && n2 >= 1 && n1 < (1ll << 31) && n2 < (1ll << 31) && A && B && idx2 && d1 && d2, APS_E_ARG,
                    "Expected input to be nonempty.");  // (the reference's `arguments` block message for empty A / B)
        APS_REQUIRE(n_components >= 1, APS_E_ARG, "ApproxNumComponents must be positive");
        APS_REQUIRE(layout == APS_ROWMAJOR ? (lda >= dim && ldb >= dim) : (lda >= n1 && ldb >= n2), APS_E_DIM, "leading dimension too small");
        ctx();
        const bool project = use_pca != 0 && dim > n_components;  // :478
        const int ncomp = project ? n_components : 0;
        const int nc = project ? n_components : kD;
        const int KH = nc <= 48 ? 24 : 64, KP = 2 * KH;
        const size_t ea = layout == APS_ROWMAJOR ? (size_t)(n1 - 1) * lda + dim : (size_t)(dim - 1) * lda + n1;
        const size_t eb = layout == APS_ROWMAJOR ? (size_t)(n2 - 1) * ldb + dim : (size_t)(dim - 1) * ldb + n2;
        In<float> dA(A, ea), dB(B, eb);
        Out<uint32_t> oi(idx2, (size_t)n1);
        Out<float> o1(d1, (size_t)n1), o2(d2, (size_t)n1);
        Ws<float> mu(kD), coeff((size_t)kD * std::max(ncomp, 1));
        if (project) {
            Prof prof("pca_basis");
            const int n_chunks = (int)((n2 + kChunk - 1) / kChunk);
            Ws<double> ps((size_t)n_chunks * kD), cov((size_t)kD * kD);
            Ws<int> pc((size_t)n_chunks * kD);
            Ws<float> part((size_t)n_chunks * kD * kD);
            pca_colsum_kernel<<<n_chunks, kD, 0, stream()>>>(dB, n2, ldb, layout, ps, pc);
            pca_mean_kernel<<<1, kD, 0, stream()>>>(ps, pc, n_chunks, mu);
            pca_cov_kernel<<<n_chunks, 256, 0, stream()>>>(dB, n2, ldb, layout, mu, part);
            pca_cov_fold_kernel<<<kD * kD / 256, 256, 0, stream()>>>(part, n_chunks, 1.0 / (double)std::max<int64_t>(n2 - 1, 1), cov);
            check_launch("pca_cov_kernel");
            std::vector<double> h_cov((size_t)kD * kD);
            APS_HIP(hipMemcpyAsync(h_cov.data(), cov, h_cov.size() * sizeof(double), hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
            std::vector<float> h_coeff;
            pca_axes(h_cov, ncomp, (int)std::min<int64_t>(n2 - 1, ncomp), h_coeff);  // :481-482 with fewer than ncomp + 1 rows in B
            APS_HIP(hipMemcpyAsync(coeff, h_coeff.data(), h_coeff.size() * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipStreamSynchronize(stream()));  // (h_coeff is pageable and goes out of scope)
            if (coeff_out) {
                if (is_device_ptr(coeff_out))
                    APS_HIP(hipMemcpyAsync(coeff_out, coeff, h_coeff.size() * sizeof(float), hipMemcpyDeviceToDevice, stream()));
                else
                    std::memcpy(coeff_out, h_coeff.data(), h_coeff.size() * sizeof(float));
            }
            if (mu_out) {
                APS_HIP(hipMemcpyAsync(mu_out, mu, kD * sizeof(float), is_device_ptr(mu_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream()));
                APS_HIP(hipStreamSynchronize(stream()));
            }
        }
        Ws<float> YA((size_t)n1 * KP), YB((size_t)n2 * KP);
        {
            Prof prof("pca_project");
            const size_t smem = ((size_t)64 * (kD + 1) + (size_t)kD * std::max(ncomp, 1)) * sizeof(float);
            APS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&pca_project_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            pca_project_kernel<<<cdiv(n2, 64), 64, smem, stream()>>>(dB, n2, ldb, layout, mu, coeff, ncomp, KP, YB);
            pca_project_kernel<<<cdiv(n1, 64), 64, smem, stream()>>>(dA, n1, lda, layout, mu, coeff, ncomp, KP, YA);
            check_launch("pca_project_kernel");
        }
        {
            Prof prof("pca_cos2nn");
            const int gx = (int)cdiv(n1, 128), n_tiles = (int)((n2 + 31) / 32);
            int splits = std::max(1, std::min(n_tiles, (2048 + gx - 1) / gx));  // about two thousand workgroups
            const int tps = (n_tiles + splits - 1) / splits;
            splits = (n_tiles + tps - 1) / tps;
            Ws<Top2> part((size_t)splits * 2 * n1);
            if (KH == 24)
                cos2nn_kernel<24><<<dim3(gx, splits), 256, 0, stream()>>>(YA, (int)n1, YB, (int)n2, tps, part);
            else
                cos2nn_kernel<64><<<dim3(gx, splits), 256, 0, stream()>>>(YA, (int)n1, YB, (int)n2, tps, part);
            cos2nn_merge_kernel<<<cdiv(n1, 256), 256, 0, stream()>>>(part, (int)n1, splits * 2, oi, o1, o2);
            check_launch("cos2nn_kernel");
        }
        oi.commit();
        o1.commit();
        o2.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}
