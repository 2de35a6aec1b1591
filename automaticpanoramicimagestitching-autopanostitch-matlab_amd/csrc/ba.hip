// ba.hip — per-pair normal-equation blocks of the bundle adjustment on gfx950.
//
// Restates the parfor body of accumulateNormalEqnsBlock (PP/bundleAdjustment/bundleAdjustmentRKf.m:717-741) with
// jacobianPair (:793-899), computeSingleResidual (:1641-1686), computeJacobianWrtCamera (:1688-1783) and huberWeight
// (:1806-1829): for every matched pair of images the blocks Hii = Ji'Ji, Hjj = Jj'Jj, Hij = Ji'Jj, gi = Ji'r, gj = Jj'r
// and the energy / residual statistics.  The LM loop, the prior, the sparse assembly and the solve stay on the host.
//
// All arithmetic is f64 in the order fixed by oracle/ba_oracle.c (matrix chains left to right as written in the
// reference, inner index ascending, no fma; per-pair sums as 64 lane-strided partials + xor butterfly), so the blocks
// are bit-identical to the oracle's.  One wavefront per pair: the ten 3x3 matrices of a direction depend on the pair
// only and are computed once per lane (uniformly), a lane then walks its matches with a handful of mat-vecs each.
// Neither bound is in sight (<= 10^9 flop for the 64-view scene); the point is to take 384 pairs x 4 k matches x
// ~50 LM evaluations of interpreted per-match loops off the host.
#include <cmath>
#include <cstdint>

#include "aps_internal.h"

namespace aps {

struct BaCam {
    double f, cx, cy;
    double R[9];  // column-major
};

#define M3(A, r, c) (A)[(r) + 3 * (c)]

__device__ __forceinline__ void mul33(const double* A, const double* B, double* C) {
    double T[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double s = M3(A, r, 0) * M3(B, 0, c);
            s = s + M3(A, r, 1) * M3(B, 1, c);
            s = s + M3(A, r, 2) * M3(B, 2, c);
            M3(T, r, c) = s;
        }
#pragma unroll
    for (int e = 0; e < 9; ++e) C[e] = T[e];
}

__device__ __forceinline__ void mulv(const double* A, const double* x, double* y) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double s = M3(A, r, 0) * x[0];
        s = s + M3(A, r, 1) * x[1];
        s = s + M3(A, r, 2) * x[2];
        y[r] = s;
    }
}

__device__ __forceinline__ void kmat(const BaCam& c, double* K) {
#pragma unroll
    for (int e = 0; e < 9; ++e) K[e] = 0.0;
    M3(K, 0, 0) = c.f;
    M3(K, 1, 1) = c.f;
    M3(K, 0, 2) = c.cx;
    M3(K, 1, 2) = c.cy;
    M3(K, 2, 2) = 1.0;
}

__device__ __forceinline__ void skew_unit(int m, double* S) {
#pragma unroll
    for (int e = 0; e < 9; ++e) S[e] = 0.0;
    const double v0 = m == 0 ? 1.0 : 0.0, v1 = m == 1 ? 1.0 : 0.0, v2 = m == 2 ? 1.0 : 0.0;
    M3(S, 0, 1) = -v2;
    M3(S, 0, 2) = v1;
    M3(S, 1, 0) = v2;
    M3(S, 1, 2) = -v0;
    M3(S, 2, 0) = -v1;
    M3(S, 2, 1) = v0;
}

__device__ __forceinline__ void ksolve(const BaCam& c, double x, double y, double* out) {
    const double z = 1.0;
    out[2] = z;
    out[1] = (y - c.cy * z) / c.f;
    out[0] = (x - c.cx * z) / c.f;
}

struct DirMats {
    double M[9], G[3][9], N[3][9], D[9], Q[9], ML[9];
};

__device__ void make_dir(const BaCam& ob, const BaCam& sb, const BaCam& ol, const BaCam& sl, DirMats& d) {
    double K[9], A[9], RsT[9], S[9], T[9], nR[9];
    kmat(ob, K);
    mul33(K, ob.R, A);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M3(RsT, c, r) = M3(sb.R, r, c);
    mul33(A, RsT, d.M);
#pragma unroll
    for (int e = 0; e < 9; ++e) nR[e] = -RsT[e];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        skew_unit(m, S);
        mul33(A, S, T);
        mul33(T, RsT, d.G[m]);
        mul33(nR, S, T);
        mul33(A, T, d.N[m]);
    }
    double dK[9] = {1, 0, 0, 0, 1, 0, 0, 0, 0};
    mul33(dK, ob.R, T);
    mul33(T, RsT, d.D);
    const double f = sb.f;
    double dKi[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) dKi[e] = 0.0;
    M3(dKi, 0, 0) = -1.0 / (f * f);
    M3(dKi, 1, 1) = -1.0 / (f * f);
    M3(dKi, 0, 2) = sb.cx / (f * f);
    M3(dKi, 1, 2) = sb.cy / (f * f);
    mul33(d.M, dKi, d.Q);
    kmat(ol, K);
    mul33(K, ol.R, A);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) M3(RsT, c, r) = M3(sl.R, r, c);
    mul33(A, RsT, d.ML);
}

__device__ __forceinline__ double one_direction(const DirMats& d, const BaCam& sb, const BaCam& sl, double uox, double uoy,
                                                double usx, double usy, double sigma, double* r, double Jobs[2][4],
                                                double Jsrc[2][4]) {
    double xb[3], pH[3], v[3];
    ksolve(sb, usx, usy, xb);
    mulv(d.M, xb, pH);
    const double x = pH[0], y = pH[1];
    double z = pH[2];
    if (fabs(z) < 1e-10) z = 1e-10;
    const double iz = 1.0 / z, zz = z * z;
    const double a = -iz, cx_ = x / zz, cy_ = y / zz;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        mulv(d.G[m], xb, v);
        Jobs[0][m] = a * v[0] + cx_ * v[2];
        Jobs[1][m] = a * v[1] + cy_ * v[2];
        mulv(d.N[m], xb, v);
        Jsrc[0][m] = a * v[0] + cx_ * v[2];
        Jsrc[1][m] = a * v[1] + cy_ * v[2];
    }
    mulv(d.D, xb, v);
    Jobs[0][3] = a * v[0] + cx_ * v[2];
    Jobs[1][3] = a * v[1] + cy_ * v[2];
    const double uh[3] = {usx, usy, 1.0};
    mulv(d.Q, uh, v);
    Jsrc[0][3] = a * v[0] + cx_ * v[2];
    Jsrc[1][3] = a * v[1] + cy_ * v[2];
    double xl[3], pL[3];
    ksolve(sl, usx, usy, xl);
    mulv(d.ML, xl, pL);
    double zl = pL[2];
    if (fabs(zl) < 1e-10) zl = 1e-10;
    const double r0 = uox - pL[0] / zl, r1 = uoy - pL[1] / zl;
    const double rr = r0 * r0 + r1 * r1;
    const double nr = sqrt(rr);
    const double w = nr < sigma ? 1.0 : sigma / nr;
    const double sw = sqrt(w);
    r[0] = sw * r0;
    r[1] = sw * r1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            Jobs[q][e] = sw * Jobs[q][e];
            Jsrc[q][e] = sw * Jsrc[q][e];
        }
    return (sw * sw) * rr;
}

constexpr int kBaAcc = 59;  // Hii 16, Hjj 16, Hij 16 (column-major 4x4), gi 4, gj 4, E, r2sum, rcnt

__device__ __forceinline__ void add_rows(double* acc, const double* r, double Ji[2][4], double Jj[2][4]) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a + 4 * b] = acc[a + 4 * b] + Ji[q][a] * Ji[q][b];
                acc[16 + a + 4 * b] = acc[16 + a + 4 * b] + Jj[q][a] * Jj[q][b];
                acc[32 + a + 4 * b] = acc[32 + a + 4 * b] + Ji[q][a] * Jj[q][b];
            }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[48 + a] = acc[48 + a] + Ji[q][a] * r[q];
            acc[52 + a] = acc[52 + a] + Jj[q][a] * r[q];
        }
    }
}

__global__ __launch_bounds__(64) void ba_pair_blocks_kernel(const double* __restrict__ Ui, const double* __restrict__ Uj,
                                                           int64_t ldu, const int64_t* __restrict__ pair_ptr,
                                                           const double* __restrict__ cams, double sigma, int both,
                                                           double* __restrict__ out) {
    const int p = blockIdx.x;
    const int lane = threadIdx.x;
    BaCam c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double* s = cams + ((int64_t)p * 4 + q) * 12;
        c[q].f = s[0];
        c[q].cx = s[1];
        c[q].cy = s[2];
#pragma unroll
        for (int e = 0; e < 9; ++e) c[q].R[e] = s[3 + e];
    }
    // the direction matrices live in LDS (two x 90 doubles): every lane computes the same values, lane 0's copy is kept
    __shared__ DirMats s_dir[2];
    if (lane == 0) {
        make_dir(c[0], c[1], c[2], c[3], s_dir[0]);  // j -> i: observed in i, source j
        make_dir(c[1], c[0], c[3], c[2], s_dir[1]);  // i -> j
    }
    __syncthreads();
    double acc[kBaAcc];
#pragma unroll
    for (int e = 0; e < kBaAcc; ++e) acc[e] = 0.0;
    const int64_t r0 = pair_ptr[p], m = pair_ptr[p + 1] - r0;
    for (int64_t k = lane; k < m; k += 64) {
        const double uix = Ui[r0 + k], uiy = Ui[ldu + r0 + k], ujx = Uj[r0 + k], ujy = Uj[ldu + r0 + k];
        double r[2], Jo[2][4], Js[2][4];
        double wr = one_direction(s_dir[0], c[1], c[3], uix, uiy, ujx, ujy, sigma, r, Jo, Js);
        add_rows(acc, r, Jo, Js);
        acc[56] = acc[56] + 0.5 * wr;
        acc[57] = acc[57] + wr;
        acc[58] = acc[58] + 2.0;
        if (both) {
            wr = one_direction(s_dir[1], c[0], c[2], ujx, ujy, uix, uiy, sigma, r, Jo, Js);
            add_rows(acc, r, Js, Jo);
            acc[56] = acc[56] + 0.5 * wr;
            acc[57] = acc[57] + wr;
            acc[58] = acc[58] + 2.0;
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
#pragma unroll
        for (int e = 0; e < kBaAcc; ++e) acc[e] = acc[e] + __shfl_xor(acc[e], s);
    }
    if (lane == 0)
        for (int e = 0; e < kBaAcc; ++e) out[(int64_t)p * kBaAcc + e] = acc[e];
}

}  // namespace aps

using namespace aps;

extern "C" int aps_ba_pair_blocks(const double* Ui, const double* Uj, int64_t ldu, const int64_t* pair_ptr, int n_pairs,
                                  const double* cams, double sigma_huber, int both_directions, double* out) {
    return guarded([&] {
        APS_REQUIRE(n_pairs >= 0, APS_E_ARG, "negative pair count");
        if (n_pairs == 0) return;
        APS_REQUIRE(Ui && Uj && pair_ptr && cams && out, APS_E_ARG, "NULL argument");
        APS_REQUIRE(sigma_huber > 0.0 && std::isfinite(sigma_huber), APS_E_ARG, "sigmaHuber must be positive and finite");
        ctx();
        std::vector<int64_t> hp(n_pairs + 1);
        const bool dev_ptr = is_device_ptr(pair_ptr);
        if (dev_ptr)
            APS_HIP(hipMemcpy(hp.data(), pair_ptr, (n_pairs + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        else
            std::copy(pair_ptr, pair_ptr + n_pairs + 1, hp.begin());
        APS_REQUIRE(hp[0] >= 0, APS_E_ARG, "pair_ptr[0] < 0");
        for (int p = 0; p < n_pairs; ++p) APS_REQUIRE(hp[p + 1] >= hp[p], APS_E_ARG, "pair_ptr is not ascending");
        APS_REQUIRE(hp[n_pairs] <= ldu, APS_E_DIM, "pair_ptr[end] = %lld exceeds the leading dimension %lld",
                    (long long)hp[n_pairs], (long long)ldu);
        In<double> dUi(Ui, (size_t)2 * ldu), dUj(Uj, (size_t)2 * ldu), dc(cams, (size_t)n_pairs * 48);
        In<int64_t> dp(pair_ptr, (size_t)n_pairs + 1);
        Out<double> dout(out, (size_t)n_pairs * kBaAcc);
        {
            Prof prof("ba_pair_blocks");
            ba_pair_blocks_kernel<<<(unsigned)n_pairs, 64, 0, stream()>>>(dUi, dUj, ldu, dp, dc, sigma_huber,
                                                                          both_directions ? 1 : 0, dout.get());
        }
        check_launch("ba_pair_blocks_kernel");
        dout.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}
