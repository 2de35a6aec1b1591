// match.hip — float-descriptor matching on gfx950.
//
// Restates PP/featureMatching/matchFeaturesScratch.m (normalizeRowsL2 :217-234, nearest2SSDExhaustive
// :322-366, ratio/threshold :170-178, greedy uniqueness :186-207) and the pair scheduling of
// PP/featureMatching/featureMatchingPairwise.m:48-63.
//
// Kernels
//   prep_desc_kernel     : optional row L2-normalisation, canonical ||x||^2, and a k-permuted copy
//                          P[i][h*64+s] = X[i][2s+h] so that one lane's MFMA operands are contiguous.
//   match2nn_kernel      : the N1 x 128 . 128 x N2 distance GEMM on v_mfma_f32_32x32x2_f32 (exact f32,
//                          k-ascending fma chain) with the top-2 reduction fused into the epilogue.
//                          A (32 rows x 128 k per wave) lives in 64 VGPRs for the whole workgroup
//                          lifetime; B streams through a padded, double-buffered LDS tile.
//   filter/unique kernels: ratio + threshold test in f64 (as MATLAB evaluates it), one-to-one
//                          resolution as a per-column atomicMin on an order-preserving 64-bit key,
//                          segmented radix sort (rocPRIM) to the reference's stable ascending order.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <atomic>
#include <type_traits>

#include "aps_internal.h"

#include <rocprim/rocprim.hpp>

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDim = 128;   // SIFT descriptor length; the MFMA path is specialised for it
constexpr int kTM = 128;    // A rows per workgroup (4 waves x 32)
constexpr int kTN = 64;     // B rows (= distance-matrix columns) per LDS tile
constexpr int kLdsRow = 132;  // floats per LDS row: 128 + one 16-B pad => ds_read_b128 conflict-free

// ------------------------------------------------------------------------------------------------
// |x| maximum per descriptor set (the reference's "looks unnormalised" probe, :105)
// ------------------------------------------------------------------------------------------------
// contiguous rows (the common case): a flat float4 sweep, no index arithmetic; one atomic per workgroup
// (thousands of same-address atomics cost more than the sweep itself)
__global__ __launch_bounds__(256) void absmax_flat_kernel(const float4* __restrict__ X, int64_t n4, float* __restrict__ out) {
    __shared__ float s_m[4];
    float m = 0.f;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = X[e];
        m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}

// The same sweep for all sets of a batch in one launch (blockIdx.y = set): 64 probes of ~9 us each, one after the other
// on one stream, were 0.6 ms of every matching call.
struct AbsmaxJob {
    const float4* x;
    int64_t n4;
};
__global__ __launch_bounds__(256) void absmax_batch_kernel(const AbsmaxJob* __restrict__ jobs, float* __restrict__ out) {
    __shared__ float s_m[4];
    const AbsmaxJob jb = jobs[blockIdx.y];
    // (global address space stated explicitly: a pointer read from a table otherwise compiles to flat loads)
    const __attribute__((address_space(1))) f32x4* X = (const __attribute__((address_space(1))) f32x4*)jb.x;
    float m = 0.f;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < jb.n4; e += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = X[e];
        m = fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0 && jb.n4 > 0)
        atomicMax(reinterpret_cast<unsigned*>(out) + blockIdx.y, __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));
}

__global__ void absmax_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int dim,
                              int layout, float* __restrict__ out) {
    float m = 0.f;
    const int64_t total = n * dim;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        int64_t i, k;
        if (layout == APS_ROWMAJOR) {
            i = e / dim;
            k = e % dim;
        } else {
            k = e / n;
            i = e % n;
        }
        const float v = fabsf(layout == APS_ROWMAJOR ? X[i * ld + k] : X[i + k * ld]);
        m = fmaxf(m, v);  // NaN is ignored like MATLAB's max
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
}

// ------------------------------------------------------------------------------------------------
// prep: one thread per descriptor row (the canonical sums are serial k-ascending chains)
// ------------------------------------------------------------------------------------------------
// f32 -> f16, round to nearest even, saturated to the finite range and with subnormal results flushed to zero
// (the matrix pipe is never asked to honour f16 subnormals); `back` = the value the f16 stands for
__device__ __forceinline__ unsigned short f32_to_f16_flush(float f, float& back) {
    const float c = fminf(fmaxf(f, -65504.f), 65504.f);
    _Float16 hv = (_Float16)c;
    back = (float)hv;
    if (fabsf(back) < 6.103515625e-05f) {
        hv = (_Float16)0.f;
        back = 0.f;
    }
    return __builtin_bit_cast(unsigned short, hv);
}

// smallest f16 >= f for f >= 0 (saturating at 65504; a nonzero f below the normal range becomes 2^-14)
__device__ __forceinline__ unsigned short f16_round_up(float f, float& back) {
    if (!(f > 0.f)) {
        back = 0.f;
        return 0;
    }
    const float c = fminf(fmaxf(f, 6.103515625e-05f), 65504.f);
    unsigned short bits = __builtin_bit_cast(unsigned short, (_Float16)c);
    if ((float)__builtin_bit_cast(_Float16, bits) < c) ++bits;  // positive finite: the next pattern is the next value
    back = (float)__builtin_bit_cast(_Float16, bits);
    return bits;
}

// order-preserving f32 <-> u32 (larger float <=> larger unsigned)
__device__ __forceinline__ unsigned ord_f32(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ unsigned wave_umax(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
    return v;
}

// One 256-thread workgroup = 64 rows (round 4; it was ONE wave with a row per lane and x[128] in registers - 256 VGPRs, one
// wave per SIMD, ~100 us of dependent work per workgroup: 0.9 ms for 64 sets whose 1.6 GB move in 0.4).  The tile lives in
// LDS, and only what has to be serial is: phase A, all threads load the 64 x 128 tile as whole rows (a row-major set; other
// layouts element by element); phase B, one lane per row walks the canonical k-ascending chain ||x||^2 (and, when the set
// is normalised, the chain of the raw row first, all threads divide the tile, then the chain again); phase C, sixteen
// threads per row turn eight consecutive elements each into the permuted f32 copy (evens | odds), the f16 copy and their
// share of the row's rounding loss and of the set's extremes.  Values and codes are those of the one-wave kernel; the
// rounding-loss norm dn is summed in a different order (eight elements per thread, then a butterfly over the sixteen), which
// its 2^-10 margin covers just the same.
constexpr int kPrepRows = 64, kPrepThreads = 256;
constexpr int kPrepPitch = kDim + 4;  // floats; 132 = 4 (mod 64): the ds_read_b128 of 16 lanes on 16 rows hit 64 banks
struct PrepLds {
    float t[kPrepRows * kPrepPitch];
    float row[kPrepRows];    // ||x||^2 of the (normalised) row
    float nrm[kPrepRows];    // normalisation divisor
    unsigned red[4][5];      // per wave: max ||x||^2, max dn, max x, ~min x, ~min ||x||^2 (as order-preserving integers)
};
__device__ __forceinline__ float prep_row_chain(const float* __restrict__ row) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < kDim / 4; ++j) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * j);
        s = __fadd_rn(s, __fmul_rn(v.x, v.x));
        s = __fadd_rn(s, __fmul_rn(v.y, v.y));
        s = __fadd_rn(s, __fmul_rn(v.z, v.z));
        s = __fadd_rn(s, __fmul_rn(v.w, v.w));
    }
    return s;
}
typedef __attribute__((address_space(1))) float GF32;
typedef __attribute__((address_space(1))) unsigned short GU16;
typedef __attribute__((address_space(1))) f32x4 GF32x4;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4_t GU32x4;
__device__ __forceinline__ void prep_desc_rows(int64_t blk, PrepLds& L, const float* __restrict__ X_, int64_t n, int64_t ld,
                                 int layout, int normalize, float* __restrict__ P_, float* __restrict__ sq_,
                                 unsigned short* __restrict__ Hf_, float* __restrict__ dn_,
                                 unsigned* __restrict__ part) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r0 = blk * (int64_t)kPrepRows;
    // (global address space stated explicitly: pointers read from a job table otherwise compile to FLAT loads and stores,
    // which count on the LDS counter too - every wait for an LDS read then waited for the stores in flight as well)
    const bool tiled = layout == APS_ROWMAJOR && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(X_) & 15) == 0;
    const GF32* X = (const GF32*)X_;
    GF32* P = (GF32*)P_;
    GF32* sq = (GF32*)sq_;
    GF32* dn = (GF32*)dn_;
    GU16* Hf = (GU16*)Hf_;
    // phase A (rows past the end repeat the last row: they change no extreme and store nothing)
    if (tiled) {
#pragma unroll
        for (int it = 0; it < kPrepRows * (kDim / 4) / kPrepThreads; ++it) {  // eight 16-byte pieces per thread, two rows per wave instruction
            const int idx = it * kPrepThreads + tid, row = idx >> 5, c4 = idx & 31;
            const int64_t gr = r0 + row < n ? r0 + row : n - 1;
            *reinterpret_cast<f32x4*>(&L.t[row * kPrepPitch + 4 * c4]) = *(const GF32x4*)(X + gr * ld + 4 * c4);
        }
    } else if (layout == APS_ROWMAJOR) {
        for (int e = tid; e < kPrepRows * kDim; e += kPrepThreads) {
            const int row = e >> 7, k = e & (kDim - 1);
            const int64_t gr = r0 + row < n ? r0 + row : n - 1;
            L.t[row * kPrepPitch + k] = X[gr * ld + k];
        }
    } else {
        for (int e = tid; e < kPrepRows * kDim; e += kPrepThreads) {
            const int row = e & (kPrepRows - 1), k = e >> 6;
            const int64_t gr = r0 + row < n ? r0 + row : n - 1;
            L.t[row * kPrepPitch + k] = X[gr + k * ld];
        }
    }
    __syncthreads();
    // phase B
    if (normalize) {  // (uniform)
        if (tid < kPrepRows) {
            // n = sqrt(sum(X.^2,2)) + eps('single'); Xn = X ./ n   (matchFeaturesScratch.m:232-233)
            // NB: sqrtf is correctly rounded on gfx950/ROCm 7.2; __fsqrt_rn is NOT (scripts/probe/fpcheck.hip)
            L.nrm[tid] = __fadd_rn(sqrtf(prep_row_chain(&L.t[tid * kPrepPitch])), 1.1920928955078125e-07f);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kPrepRows * (kDim / 4) / kPrepThreads; ++it) {
            const int idx = it * kPrepThreads + tid, row = idx >> 5, c4 = idx & 31;
            f32x4* pv = reinterpret_cast<f32x4*>(&L.t[row * kPrepPitch + 4 * c4]);
            const float nr = L.nrm[row];
            f32x4 v = *pv;
            v.x = __fdiv_rn(v.x, nr);
            v.y = __fdiv_rn(v.y, nr);
            v.z = __fdiv_rn(v.z, nr);
            v.w = __fdiv_rn(v.w, nr);
            *pv = v;
        }
        __syncthreads();
    }
    if (tid < kPrepRows) {
        const float s = prep_row_chain(&L.t[tid * kPrepPitch]);
        L.row[tid] = s;
        if (r0 + tid < n) sq[r0 + tid] = s;
    }
    // phase C: item = (row, eight consecutive elements); a wave instruction covers four rows
    float mx = -INFINITY, mn = INFINITY;
    unsigned md = 0u;  // max dn as a bit pattern (dn >= 0: unsigned order = float order, and a NaN dn stays on top instead of being dropped)
#pragma unroll
    for (int it = 0; it < kPrepRows * (kDim / 8) / kPrepThreads; ++it) {
        const int idx = it * kPrepThreads + tid, row = idx >> 4, c8 = idx & 15;
        const bool live = r0 + row < n;
        const f32x4 va = *reinterpret_cast<const f32x4*>(&L.t[row * kPrepPitch + 8 * c8]);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(&L.t[row * kPrepPitch + 8 * c8 + 4]);
        const float x[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        if (live) {  // the permuted f32 copy: even k, then odd k
            *(GF32x4*)(P + (r0 + row) * kDim + 4 * c8) = f32x4{x[0], x[2], x[4], x[6]};
            *(GF32x4*)(P + (r0 + row) * kDim + kDim / 2 + 4 * c8) = f32x4{x[1], x[3], x[5], x[7]};
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // fmaxf / fminf drop NaNs; they surface in ||x||^2 / maxsq
            mx = fmaxf(mx, x[j]);
            mn = fminf(mn, x[j]);
        }
        if (Hf) {  // (uniform) screening copy in natural k order: xf = f16(x), and the norm of what the rounding dropped,
                   // ||x - xf|| (rounded up): the candidate kernel's error bound is built from these norms, not from a worst case
            unsigned short hv[8];
            float ds = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float back;
                hv[j] = f32_to_f16_flush(x[j], back);
                const float d = x[j] - back;  // exact when not saturated
                ds = fmaf(d, d, ds);
            }
            if (live)
                *(GU32x4*)(Hf + (r0 + row) * kDim + 8 * c8) =
                    u32x4_t{hv[0] | ((uint32_t)hv[1] << 16), hv[2] | ((uint32_t)hv[3] << 16), hv[4] | ((uint32_t)hv[5] << 16),
                            hv[6] | ((uint32_t)hv[7] << 16)};
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) ds += __shfl_xor(ds, off);
            const float dnv = sqrtf(ds) * 1.0009765625f;  // 128 + 4 roundings of 2^-24 in ds, one in the root: 2^-10 covers
            if (c8 == 0 && live) dn[r0 + row] = dnv;
            md = max(md, __float_as_uint(dnv));
        }
    }
    // the workgroup's maxima (the bit patterns are compared as unsigned: same result as on the floats)
    __syncthreads();  // (L.row is complete)
    const float srow = tid < kPrepRows ? L.row[tid] : 0.f;  // (rows past the end repeat the last row)
    const unsigned ms = wave_umax(tid < kPrepRows ? __float_as_uint(srow) : 0u);  // s >= 0
    const unsigned sb = wave_umax(tid < kPrepRows ? ~__float_as_uint(fabsf(srow)) : 0u);
    const unsigned mdw = wave_umax(md);
    const unsigned xb = wave_umax(ord_f32(mx)), nb = wave_umax(~ord_f32(mn));
    if (lane == 0) {
        L.red[wv][0] = ms;
        L.red[wv][1] = mdw;
        L.red[wv][2] = xb;
        L.red[wv][3] = nb;
        L.red[wv][4] = sb;
    }
    __syncthreads();
    if (tid < 5)  // this workgroup's five maxima; prep_stats_reduce folds them into the set's statistics words
        part[tid] = max(max(L.red[0][tid], L.red[1][tid]), max(L.red[2][tid], L.red[3][tid]));
}

// The set statistics are maxima over all rows of a set.  They used to be atomicMax updates of the set's words, one per
// workgroup and word behind a volatile look - and that WAS the preparation's time: 20 000 workgroups queueing on the two
// cache lines that hold 64 sets' words kept every workgroup resident for ~50 us (the kernel without its last six lines:
// 283 us instead of 1151; q8_desc_kernel likewise, one look per row).  Now a workgroup stores its maxima, and one small
// workgroup per set folds them: [0] max ||x||^2 and [1] max dn (f16 path), [4] max x, [7] ~min x, [6] ~min ||x||^2 (int8
// codes), all as order-preserving unsigned words over a zero fill, as before.
__device__ __forceinline__ void prep_stats_reduce(const unsigned* __restrict__ part, int n_blk, float* __restrict__ stat, bool hf,
                                                  unsigned (*red)[5]) {
    unsigned r[5] = {0u, 0u, 0u, 0u, 0u};
    for (int b = threadIdx.x; b < n_blk; b += blockDim.x)
#pragma unroll
        for (int k = 0; k < 5; ++k) r[k] = max(r[k], part[(size_t)b * 5 + k]);
#pragma unroll
    for (int k = 0; k < 5; ++k) r[k] = wave_umax(r[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = r[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* su = reinterpret_cast<unsigned*>(stat);
#pragma unroll
        for (int k = 0; k < 5; ++k) r[k] = max(max(red[0][k], red[1][k]), max(red[2][k], red[3][k]));
        if (hf) {
            su[0] = max(su[0], r[0]);
            su[1] = max(su[1], r[1]);
        }
        su[4] = max(su[4], r[2]);
        su[7] = max(su[7], r[3]);
        su[6] = max(su[6], r[4]);
    }
}
// ... and from q8_desc_kernel's workgroups (kQ8Part words each): [5] the largest column-side residual norm of the int8 codes,
// and the exact codes' words (round 6): [8] the largest row divisor, [9] ~ the smallest, [10] != 0: some row has no exact code
constexpr int kQ8Part = 4;
constexpr int kStatWords = 12;  // statistics words per set
__device__ __forceinline__ void q8_stats_reduce(const unsigned* __restrict__ part, int n_blk, float* __restrict__ stat, unsigned (*red)[5]) {
    unsigned r[kQ8Part] = {0u, 0u, 0u, 0u};
    for (int b = threadIdx.x; b < n_blk; b += blockDim.x)
#pragma unroll
        for (int k = 0; k < kQ8Part; ++k) r[k] = max(r[k], part[(size_t)b * kQ8Part + k]);
#pragma unroll
    for (int k = 0; k < kQ8Part; ++k) r[k] = wave_umax(r[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < kQ8Part; ++k) red[threadIdx.x >> 6][k] = r[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* su = reinterpret_cast<unsigned*>(stat);
#pragma unroll
        for (int k = 0; k < kQ8Part; ++k) r[k] = max(max(red[0][k], red[1][k]), max(red[2][k], red[3][k]));
        su[5] = max(su[5], r[0]);
        su[8] = max(su[8], r[1]);
        su[9] = max(su[9], r[2]);
        su[10] = max(su[10], r[3]);
    }
}
__global__ __launch_bounds__(256) void prep_stats_kernel(const unsigned* __restrict__ part, int n_blk1, int n_blk2, float* __restrict__ stat,
                                                         int hf, int which) {
    __shared__ unsigned red[4][5];
    if (which == 1)
        prep_stats_reduce(part, n_blk1, stat, hf != 0, red);
    else
        q8_stats_reduce(part + (size_t)5 * n_blk1, n_blk2, stat, red);
}

__global__ __launch_bounds__(kPrepThreads) void prep_desc_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout,
                                 int normalize, float* __restrict__ P, float* __restrict__ sq,
                                 unsigned short* __restrict__ Hf, float* __restrict__ dn, unsigned* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) PrepLds L;
    prep_desc_rows(blockIdx.x, L, X, n, ld, layout, normalize, P, sq, Hf, dn, part + (size_t)5 * blockIdx.x);
}

// Every set of a pair batch in ONE launch (round 4): block b belongs to job j with blk_ptr[j] <= b < blk_ptr[j + 1].
// (History: round 2 measured a batched launch at 7.3 ms against 2.1 for eight streams of per-set launches, early round 4 both
// at 2.0 - all of them the statistics' atomics, see prep_stats_reduce: 0.5 ms for 64 sets now.)
struct PrepJob {
    const float* X;
    int64_t n, ld, n_pad;
    int layout, normalize;
    float *P, *sq, *dn, *stat;  // stat: the set's eight statistics words ([0] max ||x||^2, [1] max dn, [2..3] aug residuals, [4..7] int8)
    unsigned short* Hf;
    signed char *QA, *QB;
    float *dnq, *invs;
    int* sumq;
    uint4* aug;
    unsigned* part;  // per-workgroup maxima: 5 words per prep_desc workgroup, then one per q8_desc workgroup
    int nb1, nb2;    // ... and how many of each
    // round 6, the exact integer codes (q8_desc_rows): code bytes, per row the divisor, per row 128 x the code sum (n_pad entries)
    signed char* QX;
    float* tt;
    int* cin;
};
__device__ __forceinline__ int prep_find_job(const int* __restrict__ blk_ptr, int n_jobs, int b) {
    int lo = 0, hi = n_jobs - 1;  // largest j with blk_ptr[j] <= b
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (blk_ptr[mid] <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(kPrepThreads) void prep_desc_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ blk_ptr, int n_jobs) {
    __shared__ __attribute__((aligned(16))) PrepLds L;
    const int j = prep_find_job(blk_ptr, n_jobs, (int)blockIdx.x);
    const PrepJob J = jobs[j];
    const int64_t blk = (int64_t)blockIdx.x - blk_ptr[j];
    prep_desc_rows(blk, L, J.X, J.n, J.ld, J.layout, J.normalize, J.P, J.sq, J.Hf, J.dn, J.part + (size_t)5 * blk);
}
__global__ __launch_bounds__(256) void prep_stats_batch_kernel(const PrepJob* __restrict__ jobs, int which) {
    __shared__ unsigned red[4][5];
    const PrepJob J = jobs[blockIdx.x];
    if (which == 1)
        prep_stats_reduce(J.part, J.nb1, J.stat, J.Hf != nullptr, red);
    else
        q8_stats_reduce(J.part + (size_t)5 * J.nb1, J.nb2, J.stat, red);
}

// Scales of the three b2/2 pieces of a descriptor set: b2/2 <= 2^e for the set's largest norm; piece i carries bits
// [e-11i-11, e-11i) and is stored as p_i = r_i / c_i with c_i = 2^clamp(e-11i, -14, 15), so that c_i is a normal f16
// and p_i is one whenever the set's norms are not wildly apart (a piece that would be subnormal is dropped: it shows
// up in the residual).  The candidate kernel derives the same c_i for the A-side constants.
__device__ __forceinline__ void aug_scales(float maxsq, float* ca, float* cinv) {
    int e;
    (void)frexpf(0.5f * maxsq, &e);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int ea = min(max(e - 11 * i, -14), 15);
        ca[i] = ldexpf(1.0f, ea);
        cinv[i] = ldexpf(1.0f, -ea);
    }
}

// The extra k-step's B-side operand of every column, ready to be DMA'd into LDS: 16 bytes per descriptor =
// [p0 p1 p2 dn^ 0 0 0 0] (f16): b2/2 in three pieces and the rounding-loss norm rounded up.  Rows n .. n_pad-1 (the
// ragged end of the last 128-column tile) hold 65504 in p0: -65504 c0 loses against every real column.  Runs after
// prep_desc_kernel of the same set on the same stream (it needs the set's max ||x||^2).  res[0] / res[1] = the largest
// |b2/2 - pieces| and the largest saturation loss dn - dn^ over the set (both 0 for ordinary data).
__device__ __forceinline__ void aug_desc_row(int64_t j, const float* __restrict__ sq, const float* __restrict__ dn, int64_t n,
                                             int64_t n_pad, const float* __restrict__ maxsq, uint4* __restrict__ aug,
                                             float* __restrict__ res) {
    if (j >= n_pad) return;
    if (j >= n) {
        aug[j] = make_uint4(0x7bffu, 0u, 0u, 0u);
        return;
    }
    float ca[3], cinv[3];
    aug_scales(*maxsq, ca, cinv);
    float r = 0.5f * sq[j];
    unsigned short pc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float back;
        pc[i] = f32_to_f16_flush(r * cinv[i], back);
        r -= back * ca[i];  // exact (Sterbenz) unless the piece saturated
    }
    float dn_back;
    const unsigned short pd = f16_round_up(dn[j], dn_back);
    const float rd = fmaxf(dn[j] - dn_back, 0.f);
    aug[j] = make_uint4(pc[0] | ((uint32_t)pc[1] << 16), pc[2] | ((uint32_t)pd << 16), 0u, 0u);
    if (r != 0.f) atomicMax(reinterpret_cast<unsigned*>(res), __float_as_uint(fabsf(r)));
    if (rd > 0.f) atomicMax(reinterpret_cast<unsigned*>(res) + 1, __float_as_uint(rd));
}

// int8 copies of a prepared set for the screening pre-pass (match_screen_i8_kernel), in P's (permuted) k order - a dot
// product does not care as long as both sides agree.  A set is the row side (A) of some pairs and the column side (B) of
// others, and the two sides can afford different codes, because anything that is constant along a row of the distance
// matrix leaves the row's ranking alone and is put back in the kernel's tail:
//   row side   : qa = round(x SA_i), SA_i = 127 / max_k |x_ik| - a scale per ROW (finer steps for rows with small entries)
//   column side: qb = round(x SB) - cB, SB = 255 / (max x - min x) over the SET, cB = round(SB min x) + 128 - the full
//                eight bits whatever the data's sign; a.b = invSA_i invSB (qa.qb + cB sum_k qa_k) + error terms.
// The value a code stands for is defined through the F32 numbers invSA_i = max|x_i| / 127 and invSB = (max - min) / 255:
// x = q inv + e holds exactly with e the real residual, and the stored norms ||e|| are rounded up (each e_k comes out of
// an fma correctly rounded).  Eight lanes per row, 16 elements each.
struct Q8Set {
    float inv_sb, sb;  // column-side step and its reciprocal (0 / 0 for a degenerate range)
    float cb;          // column-side offset (an integer)
};

__device__ int g_q8_symmetric = 0;  // experiment switch (APS_Q8_SYMMETRIC=1): column code without the offset
__device__ int g_scr_variant = 0;   // experiment switch (APS_SCR_VARIANT, bits: see match_screen_i8x16_kernel)
__device__ int g_q8_noexact = 0;    // A/B switch (APS_MATCH_NO_EXACT=1): no set gets exact integer codes (rounds 2-5's screen)
#ifdef APS_MATCH_TIMING
__device__ int g_q8_center = 0;     // timing builds only (APS_Q8_CENTER=c, -1 for 0): the exact codes as clamp(u - c): WRONG results, for
                                    // measuring how the screening kernel's clock depends on the operands' magnitudes
#endif

__device__ __forceinline__ Q8Set q8_set(const float* __restrict__ qstat) {
    float mx = unord_f32(__float_as_uint(qstat[0])), mn = unord_f32(~__float_as_uint(qstat[3]));
    if (g_q8_symmetric) {
        mx = fmaxf(fabsf(mx), fabsf(mn));
        mn = -mx;
    }
    Q8Set q;
    const float range = mx - mn;
    const bool ok = range > 0.f && range < 1e30f;
    q.inv_sb = ok ? __fdiv_rn(range, 255.f) : 0.f;
    q.sb = ok ? __fdiv_rn(255.f, range) : 0.f;
    q.cb = ok ? rintf(mn * q.sb) + 128.f : 0.f;
    if (!(fabsf(q.cb) < 1e9f)) {  // a range far from zero relative to its width: the offset no longer fits the tail's arithmetic
        q.inv_sb = q.sb = q.cb = 0.f;
    }
    return q;
}

// (The launch also carries aug_desc_row - same dependency on the finished prep_desc_kernel, one launch fewer per set: the
// preparation of a pair batch is bound by its launch count.  Lane part 0 of row i, i < n_pad, writes that row's operand.)
__device__ __forceinline__ void q8_desc_rows(int64_t g, const float* __restrict__ P, int64_t n, signed char* __restrict__ QA,
                                                      signed char* __restrict__ QB, float* __restrict__ dnqa,
                                                      float* __restrict__ invsa, int* __restrict__ sumqa,
                                                      float* __restrict__ qstat, const float* __restrict__ row_sq,
                                                      const float* __restrict__ row_dn, int64_t n_pad,
                                                      const float* __restrict__ maxsq, uint4* __restrict__ aug,
                                                      float* __restrict__ aug_res, unsigned* __restrict__ part_out, unsigned (*s_red)[kQ8Part],
                                                      signed char* __restrict__ QX, float* __restrict__ ttv, int* __restrict__ cin) {
    unsigned db_bits = 0u;
    const int64_t i = g >> 3;
    const int part = (int)(g & 7);
    if (part == 0) aug_desc_row(i, row_sq, row_dn, n, n_pad, maxsq, aug, aug_res);
    const Q8Set qs = q8_set(qstat);
    float xs[16];
    float rmax = 0.f;
    if (i < n) {
        const GF32x4* src = (const GF32x4*)((const GF32*)P + i * kDim + 16 * part);  // (global, not flat: see prep_desc_rows)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const f32x4 v = src[q4];
            xs[4 * q4] = v.x;
            xs[4 * q4 + 1] = v.y;
            xs[4 * q4 + 2] = v.z;
            xs[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) rmax = fmaxf(rmax, fabsf(xs[e]));
    }
    rmax = fmaxf(rmax, __shfl_xor(rmax, 1));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 2));
    rmax = fmaxf(rmax, __shfl_xor(rmax, 4));
    const bool rok = rmax > 0.f && rmax < 1e30f;
    const float inv_sa = rok ? __fdiv_rn(rmax, 127.f) : 0.f;
    const float sa = rok ? __fdiv_rn(127.f, rmax) : 0.f;
    float dsa = 0.f, dsb = 0.f;
    int sq = 0;
    if (i < n) {
        uint32_t wa[4], wb[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            uint32_t pa = 0, pb = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = xs[4 * q4 + e];
                float qa = rintf(x * sa);
                qa = fminf(fmaxf(qa, -127.f), 127.f);  // (a NaN becomes -127; such sets are never screened)
                const float ra = fmaf(-qa, inv_sa, x);
                dsa = fmaf(ra, ra, dsa);
                sq += (int)qa;
                pa |= ((uint32_t)(int)qa & 0xffu) << (8 * e);
                float qb = rintf(x * qs.sb) - qs.cb;
                qb = fminf(fmaxf(qb, -128.f), 127.f);
                const float rb = fmaf(-(qb + qs.cb), qs.inv_sb, x);
                dsb = fmaf(rb, rb, dsb);
                pb |= ((uint32_t)(int)qb & 0xffu) << (8 * e);
            }
            wa[q4] = pa;
            wb[q4] = pb;
        }
        typedef __attribute__((address_space(1))) signed char GI8;
        *(GU32x4*)((GI8*)QA + i * kDim + 16 * part) = u32x4_t{wa[0], wa[1], wa[2], wa[3]};
        *(GU32x4*)((GI8*)QB + i * kDim + 16 * part) = u32x4_t{wb[0], wb[1], wb[2], wb[3]};
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        dsa += __shfl_xor(dsa, off);
        dsb += __shfl_xor(dsb, off);
        sq += __shfl_xor(sq, off);
    }
    if (i < n && part == 0) {
        dnqa[i] = sqrtf(dsa) * 1.001f;  // 130 roundings of 2^-24 in the sum, one in the root
        invsa[i] = inv_sa;
        sumqa[i] = sq;
        db_bits = __float_as_uint(sqrtf(dsb) * 1.001f);  // qstat[1]: the largest column-side residual norm of the set
    }
    // ---- the exact code (round 6) ----
    // SIFT descriptors are integers u of 0 .. 255 (OpenCV's quantisation) divided by their norm: x_k = fl(u_k / t), t =
    // fl(sqrt(sum u^2)).  Where a row has that form the integers are recovered here - the smallest positive entry is tried as
    // u = 1 .. kXTry, the candidate u = rint(x m / x_min) is accepted when every entry is an integer to 1e-3, none exceeds 255
    // and, with t = fl(sqrt(sum u^2)), fl(u_k / t) reproduces x_k BIT FOR BIT for all k.  Then p = u - 128 is an exact int8
    // code for rows and columns alike:
    //     u_i . u_j = p_i . p_j + 128 sum p_i + 128 sum p_j + 128^3            (integers)
    //     a . b_j   = u_i . u_j / (t_i t_j) within a factor 1 +- 2^-23          (every term >= 0, two roundings each)
    // The column term 128 sum p_j is what the screening kernel's accumulators start from (cin: its MFMA's C operand), the
    // row terms are put back in its tail, which bounds the unknown column divisor by the set's extremes of t (words [8], [9];
    // an all-zero row has every dot product 0 whatever divisor it is given: t = 1, left out of the extremes).  One row without
    // such a form (word [10]) and the set's jobs use the general codes above - both kinds are always written.
    // A set that has already lost (some earlier workgroup met a row without such a form: word [11], a plain store that later
    // workgroups may or may not see yet) skips the search: sets of ordinary floats pay for a few workgroups' tries only.
    constexpr int kXTry = 24;
    unsigned tmax_bits = 0u, tmin_bits = 0u, fail_bits = 0u;
    volatile unsigned* const xlost = reinterpret_cast<volatile unsigned*>(qstat) + 7;
    const bool lost = __builtin_amdgcn_readfirstlane((int)*xlost) != 0 || g_q8_noexact != 0;
    {
        float mn = INFINITY;
        bool nonneg = true;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (xs[e] > 0.f) mn = fminf(mn, xs[e]);
            nonneg = nonneg && xs[e] >= 0.f;  // (a NaN is not)
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            mn = fminf(mn, __shfl_xor(mn, off));
            nonneg = (__shfl_xor(nonneg ? 1 : 0, off) != 0) && nonneg;
        }
        const bool live = i < n;
        const bool zero_row = live && nonneg && mn == INFINITY;
        bool done = !live || !nonneg || zero_row || lost;  // nothing (more) to try
        bool found = zero_row;
        float t_row = 1.f;
        int usum = zero_row ? -128 * 16 : 0;  // this lane's share of sum p
        uint32_t wx[4] = {0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u};  // u = 0 everywhere
        for (int m = 1; m <= kXTry && __any(!done); ++m) {
            const float t = __fdiv_rn((float)m, mn);
            bool ok = !done;
            int qq = 0;
            float uf[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = __fmul_rn(xs[e], t);
                uf[e] = rintf(v);
                ok = ok && fabsf(v - uf[e]) <= 1e-3f * fmaxf(uf[e], 1.f) && uf[e] <= 255.f;
                qq += (int)fminf(uf[e], 255.f) * (int)fminf(uf[e], 255.f);
            }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                ok = (__shfl_xor(ok ? 1 : 0, off) != 0) && ok;
                qq += __shfl_xor(qq, off);
            }
            if (!__any(ok)) continue;  // (no row of the wave has integers at this try: the sixteen divisions below are the expensive part)
            float tq = sqrtf((float)qq);  // (qq <= 128 * 255^2 < 2^24: exact; sqrtf is correctly rounded here)
            bool same = ok;
#pragma unroll
            for (int e = 0; e < 16; ++e) same = same && __float_as_uint(__fdiv_rn(uf[e], tq)) == __float_as_uint(xs[e]);
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) same = (__shfl_xor(same ? 1 : 0, off) != 0) && same;
            // The representation (u, t) is unique only up to a common factor: a row whose integers are small has several, and
            // the one with the LARGEST integers has the divisor closest to the ~512 of ordinary rows - which is what the sorted
            // column side wants (a one-hot row would come out as u = 1, t = 1 and loosen its whole segment's bounds 500-fold).
            // Try k u with k = floor(255 / max u) under the same bit-for-bit verification, else the largest power of two (exact).
            if (__any(same && !done)) {
                float umax = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) umax = fmaxf(umax, uf[e]);
#pragma unroll
                for (int off = 1; off < 8; off <<= 1) umax = fmaxf(umax, __shfl_xor(umax, off));
                // (only rows that are clearly under-scaled, max u < 64: an ordinary SIFT row has max u >= 512 / sqrt(128) = 45, ~100
                // typically, and must keep its t ~ 512 - doubling the 38 % of rows whose max u is below 128 would split every
                // set's divisors into two clusters)
                const float kf = (umax >= 1.f && umax < 64.f) ? floorf(255.f / umax) : 1.f;
                if (__any(same && !done && kf >= 2.f)) {
                    const float tk = sqrtf((float)qq * kf * kf);  // (k^2 qq <= 128 * 255^2: exact)
                    bool same_k = same && kf >= 2.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) same_k = same_k && __float_as_uint(__fdiv_rn(uf[e] * kf, tk)) == __float_as_uint(xs[e]);
#pragma unroll
                    for (int off = 1; off < 8; off <<= 1) same_k = (__shfl_xor(same_k ? 1 : 0, off) != 0) && same_k;
                    float k2 = 1.f;  // the fallback: the largest power of two <= kf scales u and t exactly
                    while (k2 * 2.f <= kf) k2 *= 2.f;
                    const float ks = same_k ? kf : (kf >= 2.f ? k2 : 1.f);
                    if (same && !done && ks > 1.f) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) uf[e] *= ks;
                        tq = same_k ? tk : tq * ks;
                    }
                }
            }
            if (same && !done) {
                found = true;
                done = true;
                t_row = tq;
                usum = 0;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    uint32_t pw = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#ifdef APS_MATCH_TIMING
                        const int cx = g_q8_center == 0 ? 128 : max(g_q8_center, 0);
                        const int pq = min(max((int)uf[4 * q4 + e] - cx, -128), 127);
#else
                        const int pq = (int)uf[4 * q4 + e] - 128;
#endif
                        usum += pq;
                        pw |= ((uint32_t)pq & 0xffu) << (8 * e);
                    }
                    wx[q4] = pw;
                }
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) usum += __shfl_xor(usum, off);
        if (live) {
            typedef __attribute__((address_space(1))) signed char GI8;
            *(GU32x4*)((GI8*)QX + i * kDim + 16 * part) = u32x4_t{wx[0], wx[1], wx[2], wx[3]};
            if (part == 0) {
                ((GF32*)ttv)[i] = t_row;
                if (!found || lost) {
                    fail_bits = 1u;
                    if (!lost) *xlost = 1u;
                }
#ifdef APS_MATCH_TIMING
                if (!found && !lost && g_q8_center == 999) printf("[q8] row %lld of %lld has no exact code: min positive entry %g, non-negative %d, first entries %g %g %g %g\n", (long long)i, (long long)n, mn, (int)nonneg, xs[0], xs[1], xs[2], xs[3]);
#endif
                if (found && !zero_row) {
                    tmax_bits = __float_as_uint(t_row);   // t > 0: the bit patterns order like the values
                    tmin_bits = ~__float_as_uint(t_row);
                }
            }
        }
        if (i < n_pad && part == 0) ((__attribute__((address_space(1))) int*)cin)[i] = (live && found) ? 128 * usum : 0;
    }
    // (kQ8Part words per workgroup, folded by q8_stats_reduce: a look at the running maximum and an atomic per ROW kept this
    // kernel at a quarter of the rate its bytes move at)
    db_bits = wave_umax(db_bits);
    tmax_bits = wave_umax(tmax_bits);
    tmin_bits = wave_umax(tmin_bits);
    fail_bits = wave_umax(fail_bits);
    if ((threadIdx.x & 63) == 0) {
        s_red[threadIdx.x >> 6][0] = db_bits;
        s_red[threadIdx.x >> 6][1] = tmax_bits;
        s_red[threadIdx.x >> 6][2] = tmin_bits;
        s_red[threadIdx.x >> 6][3] = fail_bits;
    }
    __syncthreads();
    if (threadIdx.x < kQ8Part)
        part_out[threadIdx.x] = max(max(s_red[0][threadIdx.x], s_red[1][threadIdx.x]), max(s_red[2][threadIdx.x], s_red[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void q8_desc_kernel(const float* __restrict__ P, int64_t n, signed char* __restrict__ QA,
                                                      signed char* __restrict__ QB, float* __restrict__ dnqa,
                                                      float* __restrict__ invsa, int* __restrict__ sumqa,
                                                      float* __restrict__ qstat, const float* __restrict__ row_sq,
                                                      const float* __restrict__ row_dn, int64_t n_pad,
                                                      const float* __restrict__ maxsq, uint4* __restrict__ aug,
                                                      float* __restrict__ aug_res, unsigned* __restrict__ part,
                                                      signed char* __restrict__ QX, float* __restrict__ ttv, int* __restrict__ cin) {
    __shared__ unsigned s_red[4][kQ8Part];
    q8_desc_rows(blockIdx.x * (int64_t)blockDim.x + threadIdx.x, P, n, QA, QB, dnqa, invsa, sumqa, qstat, row_sq, row_dn, n_pad, maxsq, aug,
                 aug_res, part + (size_t)kQ8Part * blockIdx.x, s_red, QX, ttv, cin);
}
__global__ __launch_bounds__(256) void q8_desc_batch_kernel(const PrepJob* __restrict__ jobs, const int* __restrict__ blk_ptr, int n_jobs) {
    __shared__ unsigned s_red[4][kQ8Part];
    const int j = prep_find_job(blk_ptr, n_jobs, (int)blockIdx.x);
    const PrepJob J = jobs[j];
    const int64_t blk = (int64_t)blockIdx.x - blk_ptr[j];
    q8_desc_rows(blk * (int64_t)blockDim.x + threadIdx.x, J.P, J.n, J.QA, J.QB, J.dnq, J.invs, J.sumq, J.stat + 4,
                 J.sq, J.dn, J.n_pad, J.stat, J.aug, J.stat + 2, J.part + (size_t)5 * J.nb1 + (size_t)kQ8Part * blk, s_red, J.QX, J.tt, J.cin);
}

// ------------------------------------------------------------------------------------------------
// the distance GEMM + fused top-2
// ------------------------------------------------------------------------------------------------
struct MatchJob {
    const float* PA;
    const float* sqA;
    const float* PB;
    const float* sqB;
    int nA;
    int nB;
    int64_t out_off;  // first output slot of this job's rows
    // screening operands of the candidate kernel: f16 copies (natural k order), the per-row rounding-loss norm of
    // the A rows, and max ||b||^2 / max ||b - f16(b)|| of the B set
    const unsigned short* AF;
    const unsigned short* BF;
    const float* dnA;
    const uint4* augB;    // per B row the extra k-step's operand (aug_desc_kernel), padded to a multiple of 128 rows
    const float* augresB;  // [0] largest b2/2 residual, [1] largest dn saturation loss of the B set
    const float* maxsqB;
    const float* maxdnB;
    // int8 screening operands (q8_desc_kernel): the row-side code of A with its per-row step, residual norm and code sum,
    // the column-side code of B with the set's statistics ([0] max x, [1] max residual norm, [2] ~min ||x||^2, [3] ~min x)
    const signed char* AQ;
    const signed char* BQ;
    const float* dnqA;
    const float* invsA;
    const int* sumqA;
    const float* qstatB;
    // round 6, the exact integer codes (q8_desc_rows): code bytes of both sets, the A rows' divisors, 128 x the code sums of the
    // A rows and of the B columns (cinB: ncinB entries, a multiple of 128), and the sets' words [8..10]: largest divisor, ~smallest,
    // != 0 when some row has no exact code (then the job takes the general codes)
    const signed char* AX;
    const signed char* BX;
    const float* ttA;
    const int* cinA;
    const int* cinB;
    const unsigned* xstatA;
    const unsigned* xstatB;
    int ncinB;
    // ... and the B set's columns in ascending order of their divisor t (sort_columns): code bytes, 128 x code sums (ncinB
    // entries), per 256-column tile {1 / tlo (rounded up), 1 / thi (rounded down), tlo, thi} over the tile's non-zero columns
    // (zeros for a tile of all-zero columns), and the sorted position -> column map
    const signed char* BXs;
    const int* cinBs;
    const float4* tscB;
    const uint32_t* permB;
};

struct WgJob {
    int job;
    int row0;      // first A row of the tile (dense mode) / first entry of the row list (list mode)
    int list_cnt;  // list mode: number of listed rows in this tile (<= kTM)
    // match2nn_kernel's list mode with n_pool > 0 only: the B tiles [t0, t1) of a column-split tile and the index of
    // this part; parts write (idx, d1, d2) to part * n_pool + list position (merged by match2nn_merge_parts_kernel)
    int t0, t1, part;
};

__device__ __forceinline__ void top2_merge(float& b, int& i, float& s, float ob, int oi, float os) {
    const bool take = (ob < b) || (ob == b && oi < i);
    const float nb = take ? ob : b;
    const int ni = take ? oi : i;
    const float ns = take ? fminf(b, os) : fminf(s, ob);
    b = nb;
    i = ni;
    s = ns;
}

// LIST = false: the tile is the 128 consecutive A rows from w.row0.
// LIST = true : the tile is w.list_cnt rows named by row_list[w.row0 ...] (global output slots of ONE job):
//               the exact-f32 fallback for rows the split-precision path could not certify.
template <bool LIST>
__global__ __launch_bounds__(256, 2) void match2nn_kernel(const MatchJob* __restrict__ jobs,
                                                           const WgJob* __restrict__ wgs,
                                                           const uint32_t* __restrict__ row_list, int n_jobs,
                                                           uint32_t* __restrict__ out_idx,
                                                           float* __restrict__ out_d1,
                                                           float* __restrict__ out_d2, int64_t n_pool = 0) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kTN * kLdsRow];
    // list mode: a tile's rows may come from different jobs that share the B set (w.job names one of them); per row
    // the output slot, its A row (pointer into that job's permuted copy) and ||a||^2
    __shared__ int s_rows[kTM];
    __shared__ const float* s_pa[kTM];
    __shared__ float s_a2[kTM];

    const WgJob w = wgs[blockIdx.x];
    const MatchJob jb = jobs[w.job];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c = lane & 31;
    const int h = lane >> 5;
    const int nA = jb.nA, nB = jb.nB;
    if (LIST) {
        if (tid < kTM) {
            int slot = -1;
            const float* pa = jb.PA;
            float a2v = 0.f;
            if (tid < w.list_cnt) {
                slot = (int)row_list[w.row0 + tid];
                int lo = 0, hi = n_jobs - 1;  // the job whose output range holds the slot (ranges ascend with the job index)
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (jobs[mid].out_off <= (int64_t)slot) lo = mid;
                    else hi = mid - 1;
                }
                const int ar = (int)((int64_t)slot - jobs[lo].out_off);
                pa = jobs[lo].PA + (size_t)ar * kDim;
                a2v = jobs[lo].sqA[ar];
            }
            s_rows[tid] = slot;
            s_pa[tid] = pa;
            s_a2[tid] = a2v;
        }
        __syncthreads();
    }
    const int rowbase = LIST ? wave * 32 : w.row0 + wave * 32;
    auto a_row = [&](int local) -> int {  // A row index (dense mode) / output slot (list mode) of a tile row, or -1
        if (LIST) return s_rows[local];
        return local < nA ? local : -1;
    };

    // A fragments: lane (c,h) holds A[rowbase+c][2s+h], s = 0..63  == 64 contiguous floats of P
    f32x4 av[16];
    {
        const int ar = a_row(rowbase + c);
        const float* prow = LIST ? s_pa[ar >= 0 ? rowbase + c : 0] : jb.PA + (size_t)(ar >= 0 ? ar : nA - 1) * kDim;
        const f32x4* ap = reinterpret_cast<const f32x4*>(prow + h * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) av[q] = ap[q];
    }
    // C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float a2[16], best[16], second[16];
    int bidx[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = rowbase + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int ar = a_row(lr);
        a2[r] = LIST ? s_a2[lr] : jb.sqA[ar >= 0 ? ar : 0];
        best[r] = INFINITY;
        second[r] = INFINITY;
        bidx[r] = c;
    }

    // B tile staging: 64 rows x 512 B = 2048 float4; 256 threads x 8
    const int ntiles_all = (nB + kTN - 1) / kTN;
    const bool part = LIST && n_pool > 0;  // a column range [t0, t1) of a split tile (possibly empty)
    const int tfirst = part ? w.t0 : 0, ntiles = part ? min(w.t1, ntiles_all) : ntiles_all;  // (empty when t0 >= t1)
    f32x4 stage[8];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;  // float4 index in tile
            const int brow = min(t * kTN + (f >> 5), nB - 1);
            stage[u] = *reinterpret_cast<const f32x4*>(jb.PB + (size_t)brow * kDim + (f & 31) * 4);
        }
    };
    auto store_tile = [&](int buf) {
        float* base = lds + buf * (kTN * kLdsRow);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;
            *reinterpret_cast<f32x4*>(base + (f >> 5) * kLdsRow + (f & 31) * 4) = stage[u];
        }
    };

    load_tile(min(tfirst, max(ntiles_all - 1, 0)));  // (an empty part still stages a valid tile and skips the loop)
    store_tile(tfirst & 1);
    __syncthreads();

    for (int t = tfirst; t < ntiles; ++t) {
        if (t + 1 < ntiles) load_tile(t + 1);
        const float* tile = lds + (t & 1) * (kTN * kLdsRow);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int j = t * kTN + cb * 32 + c;
            const float b2 = (j < nB) ? jb.sqB[j] : INFINITY;
            const f32x4* bp =
                reinterpret_cast<const f32x4*>(tile + (cb * 32 + c) * kLdsRow + h * 64);
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                          0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 b = bp[q];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q].w, b.w, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // D2 = a2 + b2.' - 2*G   (matchFeaturesScratch.m:353), left to right, no contraction
                const float d = __fsub_rn(__fadd_rn(a2[r], b2), __fmul_rn(2.0f, acc[r]));
                const bool lt = d < best[r];
                const float s2 = (d < second[r]) ? d : second[r];
                second[r] = lt ? best[r] : s2;
                bidx[r] = lt ? j : bidx[r];
                best[r] = lt ? d : best[r];
            }
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        __syncthreads();
    }

    // merge the 32 column-lanes of each half-wave (first index wins ties, :356)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float ob = __shfl_xor(best[r], off);
            const int oi = __shfl_xor(bidx[r], off);
            const float os = __shfl_xor(second[r], off);
            top2_merge(best[r], bidx[r], second[r], ob, oi, os);
        }
    }
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = rowbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int row = a_row(lr);
            if (row >= 0) {
                const int64_t o = part ? (int64_t)w.part * n_pool + w.row0 + lr : LIST ? (int64_t)row : jb.out_off + row;
                out_idx[o] = nB > 0 ? (uint32_t)bidx[r] + 1u : 0u;
                out_d1[o] = best[r];
                out_d2[o] = second[r];
            }
        }
    }
}

// The parts of column-split list tiles, merged per row in part order (top2_merge: the smaller index wins equal
// distances, as in the kernel's own lane merge) and written to the row's output slot.
__global__ void match2nn_merge_parts_kernel(const uint32_t* __restrict__ row_list, int64_t n_pool, int n_parts,
                                            const uint32_t* __restrict__ p_idx, const float* __restrict__ p_d1,
                                            const float* __restrict__ p_d2, uint32_t* __restrict__ out_idx,
                                            float* __restrict__ out_d1, float* __restrict__ out_d2) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n_pool) return;
    float b = p_d1[e], s2 = p_d2[e];
    int i = (int)p_idx[e];  // 1-based (0: an empty column set - then every part says 0)
    for (int p = 1; p < n_parts; ++p) top2_merge(b, i, s2, p_d1[p * n_pool + e], (int)p_idx[p * n_pool + e], p_d2[p * n_pool + e]);
    const uint32_t slot = row_list[e];
    out_idx[slot] = (uint32_t)i;
    out_d1[slot] = b;
    out_d2[slot] = s2;
}

// ------------------------------------------------------------------------------------------------
// split-precision candidate search: bf16 hi/lo, three MFMA products, error-bounded
// ------------------------------------------------------------------------------------------------
// G~ = Ah.Bh + Ah.Bl + Al.Bh on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate, 3 products => 5.3x).
// With x = hi + lo + r, |r| <= 2^-16 |x|, the dropped terms are bounded by 3*2^-16 |a||b| per product and
// the f32 accumulation of 384 exact bf16 products by ~384*2^-24; kSplitEps below covers both with margin:
//   |d~ - d| <= kSplitEps * sqrt(a2 * max b2) + 2^-20   for every pair (i, j).
// The kernel keeps, per A row, the three smallest d~ with their indices and the FOURTH smallest value (a
// lower bound for every column that is not a candidate).  rescore_kernel then evaluates the canonical f32
// distance of the three candidates exactly; the row is certified iff its exact second-best is below
// (fourth - eps), otherwise it goes to the exact f32 kernel (row-list mode).  The final (idx, d1, d2) are
// therefore bit-identical to the all-f32 path.
//
// Operands are swapped w.r.t. match2nn_kernel: the streamed B-descriptor tile is the MFMA "A" operand and
// the resident A rows are the MFMA "B" operand, so that in the 32x32 accumulator a lane owns ONE A row
// (col = lane&31) and sees 16 B columns per block: the running top-4 of a row is 7 registers.
//
// The candidate kernel (see DESIGN.md section 4 for the measurements behind each point):
//  * ONE f16 product per column (v_mfma_f32_32x32x16_f16) screens the distance matrix; the bound that certifies a
//    row is built from the measured rounding loss ||x - f16(x)|| of its operands, the exact f32 distance of the three
//    best candidates is evaluated in the kernel's tail, uncertified rows (0.3 %) go to the exact f32 kernel;
//  * the B tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write
//    pass.  The LDS image is lane-linear (256-B rows, no pad); bank conflicts are avoided by an XOR swizzle of the
//    16-B chunk position with (row & 15), applied to the per-lane SOURCE address of the DMA and to the
//    ds_read_b128 address (the same involution on both sides); three buffers, hand-over one block early;
//  * -||b||^2/2 (three f16 pieces, per-set power-of-two scales) and ||a|| ||b - f16(b)|| enter the accumulator
//    through one extra 16-wide k-step (B-side operands precomputed per set, aug_desc_kernel, and DMA'd with the tile), so the row's best columns are the LARGEST acc and the selection needs no
//    arithmetic per element;
//  * the search of block g-1 is cut into eight three-instruction slices placed between the MFMA pairs of block g
//    (two accumulator sets); hits are parked and inserted in bulk from a cold path;
//  * workgroups are renumbered so that the ones sharing a B set run on one XCD (one L2).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kTMB = 512;  // A rows per workgroup of the split-precision kernel: 8 waves x 64 rows
constexpr int kTNB = 128;   // B rows per LDS tile of the candidate kernel (4 column blocks per hand-over)

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef const __attribute__((address_space(1))) float gbl_f32;

__device__ __forceinline__ void top4_insert_max(float t, int j, float& u0, float& u1, float& u2, float& u3,
                                                int& i0, int& i1, int& i2) {
    // u0 >= u1 >= u2 >= u3; ties keep the earlier entry ahead (strict compares for the indices)
    const bool g2 = t > u2, g1 = t > u1, g0 = t > u0;
    u3 = __builtin_amdgcn_fmed3f(u2, t, u3);
    i2 = g1 ? i1 : (g2 ? j : i2);
    u2 = __builtin_amdgcn_fmed3f(u1, t, u2);
    i1 = g0 ? i0 : (g1 ? j : i1);
    u1 = __builtin_amdgcn_fmed3f(u0, t, u1);
    i0 = g0 ? j : i0;
    u0 = fmaxf(u0, t);
}

// the same insertion written as plain selects (no nested conditionals for the compiler to turn into branches)
__device__ __forceinline__ void top4_insert_sel(float t, int j, float& u0, float& u1, float& u2, float& u3,
                                                int& i0, int& i1, int& i2) {
    const bool g2 = t > u2, g1 = t > u1, g0 = t > u0;
    u3 = __builtin_amdgcn_fmed3f(u2, t, u3);
    u2 = __builtin_amdgcn_fmed3f(u1, t, u2);
    u1 = __builtin_amdgcn_fmed3f(u0, t, u1);
    u0 = __builtin_amdgcn_fmed3f(u0, t, INFINITY);  // max without a canonicalising pre-pass
    int n2 = g2 ? j : i2;
    n2 = g1 ? i1 : n2;
    int n1 = g1 ? j : i1;
    n1 = g0 ? i0 : n1;
    i0 = g0 ? j : i0;
    i1 = n1;
    i2 = n2;
}

__device__ __forceinline__ float max4_raw(float a, float b, float c, float d) {
    float m;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(d));
    return m;
}

constexpr int kTileBytes = kTNB * 256;  // one f16 B tile in LDS

// exact canonical distance of A row `pa` and B row `pb` (both in the permuted f32 layout of prep_desc_kernel):
// G = k-ascending fma chain, d = (a2 + b2) - 2G — the same arithmetic as match2nn_kernel / the oracle.
__device__ __forceinline__ float exact_dist(const float* __restrict__ pa, const float* __restrict__ pb, float a2,
                                            float b2) {
    float g = 0.f;
#pragma unroll 8
    for (int s4 = 0; s4 < 16; ++s4) {
        const f32x4 ae = *reinterpret_cast<const f32x4*>(pa + 4 * s4);
        const f32x4 ao = *reinterpret_cast<const f32x4*>(pa + 64 + 4 * s4);
        const f32x4 be = *reinterpret_cast<const f32x4*>(pb + 4 * s4);
        const f32x4 bo = *reinterpret_cast<const f32x4*>(pb + 64 + 4 * s4);
        g = fmaf(ae.x, be.x, g);
        g = fmaf(ao.x, bo.x, g);
        g = fmaf(ae.y, be.y, g);
        g = fmaf(ao.y, bo.y, g);
        g = fmaf(ae.z, be.z, g);
        g = fmaf(ao.z, bo.z, g);
        g = fmaf(ae.w, be.w, g);
        g = fmaf(ao.w, bo.w, g);
    }
    return __fsub_rn(__fadd_rn(a2, b2), __fmul_rn(2.0f, g));
}

// Exact rescoring of one row's three candidates and the certification test (see the header comment): writes the
// final (idx, d1, d2) of a certified row, or appends the row to the fallback list.
// k-ascending fma chains of N candidate rows against the A row, side by side (each chain in the canonical order; the
// A row is read once and N gathers are in flight: this tail is pure memory latency - unrolled by 8 it keeps 48 16-byte
// loads per lane in flight, which measured 3 % of the kernel faster than 24; touching the lines first did not help)
template <int N>
__device__ __forceinline__ void exact_chains(const float* __restrict__ pa, const float* const* pb, float* g) {
#pragma unroll
    for (int e = 0; e < N; ++e) g[e] = 0.f;
#pragma unroll 8
    for (int s4 = 0; s4 < 16; ++s4) {
        const f32x4 ae = *reinterpret_cast<const f32x4*>(pa + 4 * s4);
        const f32x4 ao = *reinterpret_cast<const f32x4*>(pa + 64 + 4 * s4);
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const f32x4 be = *reinterpret_cast<const f32x4*>(pb[e] + 4 * s4);
            const f32x4 bo = *reinterpret_cast<const f32x4*>(pb[e] + 64 + 4 * s4);
            float t = g[e];
            t = fmaf(ae.x, be.x, t);
            t = fmaf(ao.x, bo.x, t);
            t = fmaf(ae.y, be.y, t);
            t = fmaf(ao.y, bo.y, t);
            t = fmaf(ae.z, be.z, t);
            t = fmaf(ao.z, bo.z, t);
            t = fmaf(ae.w, be.w, t);
            t = fmaf(ao.w, bo.w, t);
            g[e] = t;
        }
    }
}

// Exact rescoring of one row's candidates and the certification test (see the header comment): writes the final
// (idx, d1, d2) of a certified row, or appends the row to the fallback list.  bnd3 / bnd4 = a2 - 2 u for the third- /
// fourth-largest screened value of the row.
__device__ __forceinline__ void rescore_row(const MatchJob& jb, int job, int row, int c0, int c1, int c2, float bnd3,
                                            float bnd4, float aug_res, float dn_res, uint32_t* __restrict__ out_idx,
                                            float* __restrict__ out_d1, float* __restrict__ out_d2,
                                            uint32_t* __restrict__ fb_list, unsigned int* __restrict__ fb_count) {
    const int64_t slot = jb.out_off + row;
    const float a2 = jb.sqA[row];
    const float* pa = jb.PA + (size_t)row * kDim;
    float d[3];
    int id[3] = {c0, c1, c2};
    const float* pb[3];
    bool ok[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        ok[e] = id[e] >= 0 && id[e] < jb.nB;
        pb[e] = ok[e] ? jb.PB + (size_t)id[e] * kDim : pa;
    }
    // The screened value of column j is U_j = a^.b^_j - y~_j + na^ dn^_j, an UPPER bound (up to the row-wise terms
    // below) of s_j = a.b_j - b2_j/2:
    //   |a.b - a^.b^| <= ||a - a^|| ||b^|| + ||a|| ||b - b^||   (Cauchy-Schwarz on the two rounding losses);
    //   the second term, ||a|| dn_j, is column specific and is added by the matrix pipe itself (k-slot 3 of the
    //   extra step: na^ >= ||a|| on the A side, dn^_j >= ||b_j - b^_j|| on the B side, both rounded UP to f16);
    //   row-wise remainder: ||a - a^|| max||b^|| + the largest residual of the three-piece b2/2 the workgroup has
    //   staged + any saturation loss of dn^ + f32 accumulation in the matrix pipe, bounded by 2^-15 (sum|a_k b_k|
    //   + y) (measured <= 7 x 2^-24, scripts/probe/mfma_f16_err.hip).
    // Every column outside the best K has U_j <= u_(K+1), hence s_j <= u_(K+1) + eg; in distance units twice that,
    // plus the roundings of the canonical f32 evaluation itself.
    const float msb = *jb.maxsqB, mdb = *jb.maxdnB;
    const float nb = sqrtf(msb) * 1.000001f + mdb;
    const float na = sqrtf(a2) * 1.000001f;
    float eg = jb.dnA[row] * nb + aug_res + na * dn_res + 3.0517578125e-05f * (na * (nb + mdb) + 0.5f * msb);
    if (!(na < 65000.f)) eg = INFINITY;  // ||a|| does not fit the f16 slot: nothing is certified
    const float eps = 2.002f * eg + 1.52587890625e-05f * (a2 + msb + 2.0f * na * nb) + 1e-37f;

#define APS_CSWAP(a, b)                                                    \
    if (d[b] < d[a] || (d[b] == d[a] && id[b] < id[a])) {                  \
        const float td = d[a]; d[a] = d[b]; d[b] = td;                     \
        const int ti = id[a]; id[a] = id[b]; id[b] = ti;                   \
    }
    // Stage 1: the best two alone.  If their exact second-best beats the bound of everything else - the third
    // candidate included, through ITS screened value - the row is done after two 512-byte gathers instead of three
    // (the gathers of this tail are 60 % of the kernel's HBM traffic).
    float g[3];
    exact_chains<2>(pa, pb, g);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        d[e] = ok[e] ? __fsub_rn(__fadd_rn(a2, jb.sqB[ok[e] ? id[e] : 0]), __fmul_rn(2.0f, g[e])) : INFINITY;
        if (!ok[e]) id[e] = 0x7fffffff;
    }
    APS_CSWAP(0, 1)
    bool certified = jb.nB > 3 && d[1] < bnd3 - eps;
    if (!certified) {
        // Stage 2: the third candidate joins; the bound is the fourth screened value
        exact_chains<1>(pa, pb + 2, g + 2);
        d[2] = ok[2] ? __fsub_rn(__fadd_rn(a2, jb.sqB[ok[2] ? id[2] : 0]), __fmul_rn(2.0f, g[2])) : INFINITY;
        if (!ok[2]) id[2] = 0x7fffffff;
        APS_CSWAP(1, 2)
        APS_CSWAP(0, 1)
        certified = jb.nB <= 3 || (d[1] < bnd4 - eps);
    }
#undef APS_CSWAP
    if (certified) {
        out_idx[slot] = jb.nB > 0 ? (uint32_t)id[0] + 1u : 0u;
        out_d1[slot] = d[0];
        out_d2[slot] = d[1];
    } else {
        // the job's segment of the list starts at its first output slot (a job has at most nA uncertified rows)
        const unsigned int p = atomicAdd(fb_count + job, 1u);
        fb_list[jb.out_off + p] = (uint32_t)slot;
    }
}

// Top-3 form of rescore_row for the blocked global k-NN (aps_knn_global on a pool against itself): the exact canonical
// distances of the three candidates, ascending by (distance, index), cut to the prefix that is strictly below the bound of
// every non-candidate of this B set (a2 - 2 u4 - eps).  t3_b receives that bound: every row of the set that is not
// listed is at least that far away.
__device__ __forceinline__ void rescore_row3(const MatchJob& jb, int job, int row, int c0, int c1, int c2, float bnd4,
                                             float aug_res, float dn_res, uint32_t* __restrict__ t3_idx,
                                             float* __restrict__ t3_d, float* __restrict__ t3_b,
                                             uint32_t* __restrict__ fb_list, unsigned int* __restrict__ fb_count) {
    const int64_t slot = jb.out_off + row;
    const float a2 = jb.sqA[row];
    const float* pa = jb.PA + (size_t)row * kDim;
    float d[3];
    int id[3] = {c0, c1, c2};
    const float* pb[3];
    bool ok[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        ok[e] = id[e] >= 0 && id[e] < jb.nB;
        pb[e] = ok[e] ? jb.PB + (size_t)id[e] * kDim : pa;
    }
    const float msb = *jb.maxsqB, mdb = *jb.maxdnB;
    const float nb = sqrtf(msb) * 1.000001f + mdb;
    const float na = sqrtf(a2) * 1.000001f;
    float eg = jb.dnA[row] * nb + aug_res + na * dn_res + 3.0517578125e-05f * (na * (nb + mdb) + 0.5f * msb);
    if (!(na < 65000.f)) eg = INFINITY;
    const float eps = 2.002f * eg + 1.52587890625e-05f * (a2 + msb + 2.0f * na * nb) + 1e-37f;
    float g[3];
    exact_chains<3>(pa, pb, g);
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        d[e] = ok[e] ? __fsub_rn(__fadd_rn(a2, jb.sqB[ok[e] ? id[e] : 0]), __fmul_rn(2.0f, g[e])) : INFINITY;
        if (!ok[e]) id[e] = 0x7fffffff;
    }
#define APS_CSWAP(a, b)                                                    \
    if (d[b] < d[a] || (d[b] == d[a] && id[b] < id[a])) {                  \
        const float td = d[a]; d[a] = d[b]; d[b] = td;                     \
        const int ti = id[a]; id[a] = id[b]; id[b] = ti;                   \
    }
    APS_CSWAP(0, 1)
    APS_CSWAP(1, 2)
    APS_CSWAP(0, 1)
#undef APS_CSWAP
    // Every column that is not one of the three candidates is at least `bound` away.  A candidate whose exact distance is
    // strictly below the bound is therefore in its final place among the set's nearest; the others (if any) are no
    // nearer than the bound either, so they are simply left unlisted: the list is a CERTIFIED PREFIX of the set's
    // nearest rows, and "everything unlisted is >= bound" holds in every case.  The merge decides whether it needs more.
    const float bound = jb.nB <= 3 ? INFINITY : bnd4 - eps;
    (void)job; (void)fb_list; (void)fb_count;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const bool keep = id[e] < jb.nB && (jb.nB <= 3 || d[e] < bound);
        t3_idx[slot * 3 + e] = keep ? (uint32_t)id[e] + 1u : 0u;  // 1-based within the B set, 0 = none
        t3_d[slot * 3 + e] = keep ? d[e] : INFINITY;
    }
    t3_b[slot] = bound;
}

// Exact top-3 of one A row against its job's whole B set (the fallback of rescore_row3): one wave per listed slot, lane j
// takes columns j, j+64, ...; every distance is the canonical f32 chain (exact_dist); lists merged through LDS.
__global__ __launch_bounds__(64) void knn3_rows_kernel(const MatchJob* __restrict__ jobs, const int* __restrict__ job_of,
                                                       const uint32_t* __restrict__ rows, int n_rows,
                                                       uint32_t* __restrict__ t3_idx, float* __restrict__ t3_d,
                                                       float* __restrict__ t3_b) {
    __shared__ float s_d[64 * 4];
    __shared__ int s_i[64 * 4];
    if ((int)blockIdx.x >= n_rows) return;
    const MatchJob jb = jobs[job_of[blockIdx.x]];
    const int64_t slot = rows[blockIdx.x];
    const int row = (int)(slot - jb.out_off);
    const float a2 = jb.sqA[row];
    const float* pa = jb.PA + (size_t)row * kDim;
    // each lane keeps its FOUR nearest: a column among the overall four nearest has at most three columns ahead of it
    // anywhere, so it is among its own lane's four - the union of the lists holds the exact overall top four
    float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int id[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int j = threadIdx.x; j < jb.nB; j += 64) {
        const float dj = exact_dist(pa, jb.PB + (size_t)j * kDim, a2, jb.sqB[j]);
        if (dj < v[3]) {  // columns arrive in ascending j per lane: a tie stays behind the earlier entry
            bool lt[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) lt[e] = dj < v[e];
#pragma unroll
            for (int e = 3; e >= 1; --e) {
                v[e] = lt[e - 1] ? v[e - 1] : (lt[e] ? dj : v[e]);
                id[e] = lt[e - 1] ? id[e - 1] : (lt[e] ? j : id[e]);
            }
            v[0] = lt[0] ? dj : v[0];
            id[0] = lt[0] ? j : id[0];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s_d[threadIdx.x * 4 + e] = v[e];
        s_i[threadIdx.x * 4 + e] = id[e];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bd[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        for (int q = 0; q < 64 * 4; ++q) {
            const float dq = s_d[q];
            const int iq = s_i[q];
            int pos = 4;
            while (pos > 0 && (dq < bd[pos - 1] || (dq == bd[pos - 1] && iq < bi[pos - 1]))) --pos;
            if (pos >= 4) continue;
            for (int e = 3; e > pos; --e) {
                bd[e] = bd[e - 1];
                bi[e] = bi[e - 1];
            }
            bd[pos] = dq;
            bi[pos] = iq;
        }
        for (int e = 0; e < 3; ++e) {
            t3_idx[slot * 3 + e] = bi[e] < jb.nB ? (uint32_t)bi[e] + 1u : 0u;
            t3_d[slot * 3 + e] = bd[e];
        }
        t3_b[slot] = bd[3];  // the set's exact fourth distance: nothing unlisted is nearer (inf: nothing else exists)
    }
}

// Bounds on a row's EXACT (canonical f32) best and second-best distance from its two largest screened values u0 >= u1
// (columns c0, c1), with the same error terms as rescore_row:
//   for every column j the computed distance satisfies d_j >= a2 - 2 U_j - eps (that is what certifies a row), and
//   U_j <= u0 for all j, so   d1 = min_j d_j >= a2 - 2 u0 - eps =: L1;
//   conversely U_j <= s_j + (na^ dn^_j + ||a|| dn_j) + eg, i.e. s_j >= U_j - slack with slack = eg + 2.02 na (max dn + 1e-6)
//   (na^, dn^ are na, dn rounded UP to f16: < 0.1 % each), so the computed distances of c0 and c1 are both
//   <= a2 - 2 (u1 - slack) + delta, and d2, the second smallest of ALL columns, is at most the larger of those two: H2.
__device__ __forceinline__ void prune_bounds(const MatchJob& jb, int row, float u0r, float u1r, float aug_res, float dn_res,
                                             float& L1, float& H2) {
    const float a2 = jb.sqA[row];
    const float msb = *jb.maxsqB, mdb = *jb.maxdnB;
    const float nb = sqrtf(msb) * 1.000001f + mdb;
    const float na = sqrtf(a2) * 1.000001f;
    float eg = jb.dnA[row] * nb + aug_res + na * dn_res + 3.0517578125e-05f * (na * (nb + mdb) + 0.5f * msb);
    if (!(na < 65000.f)) eg = INFINITY;
    const float delta = 1.52587890625e-05f * (a2 + msb + 2.0f * na * nb) + 1e-37f;
    const float eps = 2.002f * eg + delta;
    const float slack = eg + 2.02f * na * (mdb + 1e-6f);
    L1 = a2 - 2.0f * u0r - eps;
    H2 = a2 - 2.0f * (u1r - slack) + delta;
    if (!(eg < INFINITY)) {  // data outside the f16 range: nothing is dismissed
        L1 = -INFINITY;
        H2 = INFINITY;
    }
}

template <bool LIST>
__global__ __launch_bounds__(512) void match_cand_f16_kernel(const MatchJob* __restrict__ jobs,
                                                                const WgJob* __restrict__ wgs, int n_wg,
                                                                uint32_t* __restrict__ out_idx,
                                                                float* __restrict__ out_d1, float* __restrict__ out_d2,
                                                                uint32_t* __restrict__ fb_list,
                                                                unsigned int* __restrict__ fb_count, int ablate,
                                                                float prune_r2, float prune_thr,
                                                                uint32_t* __restrict__ t3_idx, float* __restrict__ t3_d,
                                                                float* __restrict__ t3_b,
                                                                const uint32_t* __restrict__ row_list,
                                                                const int* __restrict__ row_job) {
    // row_job != nullptr (LIST only, round 4): a POOLED list - the tile's rows are row_list[w.row0 ...] of the jobs
    // row_job[w.row0 ...], all of which share w.job's B set (the survivor lists of the jobs that share a B set, packed into
    // common 512-row tiles: a pair of the 64 x 4K scene leaves ~600 survivors in a non-overlapping pair - one full tile and
    // one that is 83 % padding; pooled, 1806 such lists fill their tiles).  Everything on the A side is then per row.
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * kTileBytes];  // [buf][128][256 B], tile t in buf t % 3
    // b2/2 of each B row as three f16 pieces (p0 c0 + p1 c1 + p2 c2 == the f32 value, c_i powers of two chosen per
    // B set) + five zeros: one extra 16-wide k-step against the constant [-c0 -c1 -c2 0 ...] puts -b2/2 into the
    // accumulator (padded column: -65504 c0)
    __shared__ __attribute__((aligned(1024))) uint4 s_aug[3][kTNB];  // filled by DMA with the B tile (aug_desc_kernel's rows)

    // XCD-aware order: consecutive workgroup ids go to different XCDs (one L2 each); give each XCD a
    // contiguous run of the job-major list so that the workgroups sharing a B set share an L2
    int wg = blockIdx.x;
    {
        const int q = n_wg / 8, r = n_wg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
#ifdef APS_MATCH_TIMING
    const unsigned long long T_entry = __builtin_readcyclecounter();
#endif
    const WgJob w = wgs[wg];
    const MatchJob jb = jobs[w.job];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;
    const int h = lane >> 5;
    const int nA = jb.nA, nB = jb.nB;
    // this lane owns tile rows wave * 64 + c and + 32.  LIST = false: the tile is the 512 consecutive A rows from w.row0;
    // LIST = true: it is the w.list_cnt rows row_list[jb.out_off + w.row0 ...] of this job (the survivors of the int8 screen)
    int rowid[2], rjob[2];
    bool rvalid[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int local = wave * 64 + c + 32 * rb;
        rjob[rb] = w.job;
        if (LIST) {
            rvalid[rb] = local < w.list_cnt;
            if (row_job) {
                const int e = w.row0 + (rvalid[rb] ? local : 0);
                rowid[rb] = (int)row_list[e];
                rjob[rb] = row_job[e];
            } else {
                rowid[rb] = (int)row_list[jb.out_off + w.row0 + (rvalid[rb] ? local : 0)];
            }
        } else {
            rvalid[rb] = w.row0 + local < nA;
            rowid[rb] = min(w.row0 + local, nA - 1);
        }
    }

    f16x8 ah[2][8];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int arow = rowid[rb];
        const unsigned short* af = (LIST && row_job) ? jobs[rjob[rb]].AF : jb.AF;
#pragma unroll
        for (int s = 0; s < 8; ++s)
            ah[rb][s] = *reinterpret_cast<const f16x8*>(af + (size_t)arow * kDim + 16 * s + 8 * h);
    }
    float u0[2], u1[2], u2[2], u3[2];  // a.b - b2/2, descending, per owned row
    int i0[2], i1[2], i2[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        u0[rb] = u1[rb] = u2[rb] = u3[rb] = -INFINITY;
        i0[rb] = i1[rb] = i2[rb] = -1;
    }

    float ca[3], cinv[3];
    aug_scales(*jb.maxsqB, ca, cinv);
    (void)cinv;

    const int ntiles = (nB + kTNB - 1) / kTNB;
    // DMA pieces: 32 per tile, 1 KiB = 4 LDS rows each; wave w issues pieces 4w .. 4w+3, one per column block.
    // lane -> LDS row 4*piece + lane/16, chunk position lane&15, which must hold source chunk pos ^ (row&15)
    const int dma_sub = lane >> 4, dma_pos = lane & 15;
    // The DMA is issued from inline asm on purpose: for the builtin the compiler cannot tell the two LDS
    // buffers apart and drains vmcnt before the first ds_read that follows, which exposes the whole DMA
    // latency once per tile.  The DMA of tile t+1 is retired by the explicit vmcnt(0) before the barrier that
    // ends tile t; nothing reads that buffer earlier.
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const uint32_t aug_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&s_aug[0][0];
    // piece u of this wave's four; with its piece 0, wave 0 / wave 1 also fetch one half of the tile's 2 KiB of extra
    // k-step operands (64 columns x 16 B each, lane-linear)
    auto issue_piece = [&](int t, int buf, int u) {
        const int piece = wave * (kTNB / 32) + u;
        const int lrow = 4 * piece + dma_sub;
        const int brow = min(t * kTNB + lrow, nB - 1);
        const unsigned short* src = jb.BF + (size_t)brow * kDim + ((dma_pos ^ (lrow & 15)) << 3);
        const uint32_t dst = lds_base + buf * kTileBytes + piece * 1024;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
        if (u == 0 && wave < 2) {
            const uint4* asrc = jb.augB + (size_t)t * kTNB + wave * 64 + lane;  // the array is padded to whole tiles
            const uint32_t adst = aug_base + buf * (kTNB * 16) + wave * 1024;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(asrc), "s"(adst)
                : "memory");
        }
    };
    f32x16 acc[2][2];  // [block parity][owned row]: one set is being accumulated while the other is being searched
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = -INFINITY;
    // Selection.  thr = the larger of the two half-waves' fourth-best of a row (they see disjoint columns of the
    // same row): the union's fourth-best is >= either half's, so a value <= thr can neither be one of the
    // union's best three nor exceed the final bound, and each half's list still ends up holding every element
    // of the union's top four that fell into its columns, which is all the final merge needs.
    //
    // With a single product per column the selection, not the matrix pipe, sets the pace (VALU issued next to a
    // dense MFMA stream runs at a third of its rate, scripts/probe/mfma_valu_coexec.hip), and late in the stream
    // nearly every "some lane has a hit" event serves one lane of 64.  So a hit is not inserted on the spot: the
    // lane PARKS the whole group of four values (+ its first column) in five registers, and the sorted insertion of
    // all four runs for every lane at once only when a lane that is already holding a parked group hits again
    // (every ~10 events), with no per-value tests at all: inserting a value <= the list's fourth entry, or the
    // -inf of an empty slot, changes nothing.  A parked group only delays the tightening of that lane's threshold.
    float thr[2] = {-INFINITY, -INFINITY};
#ifdef APS_MATCH_TIMING
    if (ablate & 4) thr[0] = thr[1] = INFINITY;  // timing experiment: the screen never fires
#endif
    float pv[2][4];
    int pj[2] = {-1, -1};
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[rb][q] = -INFINITY;
    auto drain = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            top4_insert_sel(pv[rb][q], pj[rb] + q, u0[rb], u1[rb], u2[rb], u3[rb], i0[rb], i1[rb], i2[rb]);
            pv[rb][q] = -INFINITY;
        }
        pj[rb] = -1;
        thr[rb] = __builtin_amdgcn_fmed3f(thr[rb], u3[rb], INFINITY);
    };
    // One eighth of the search of a finished block: group g = sl / 2 (four columns) of owned row rb = sl % 2.
    // acc[par][rb][r] <-> B column (r&3) + 8*(r>>2) + 4*h of block (t, cb), A row = the lane's row (rb).
    float m_scr = 0.f;
    auto select_slice = [&](auto PAR, int t, auto CB, auto SL) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value, cb = decltype(CB)::value, sl = decltype(SL)::value;
#ifdef APS_MATCH_TIMING
        if (__builtin_expect(ablate & 1, 0)) {
            asm volatile("" ::"v"(acc[par][0]), "v"(acc[par][1]));
            return;
        }
#endif
        constexpr int g = sl >> 1, rb = sl & 1;
        const f32x16& a = acc[par][rb];
        // max of the four, compared against the row's threshold: three VALU instructions and a scalar branch are
        // all that sits between two MFMA pairs in the common case (every further instruction there costs matrix
        // time: a wave issues in order, and the next MFMA pair cannot start before the slice is through)
        // (the result register is one that stays live through the whole loop ("+v"): left to itself the allocator
        // puts the temporary into the operand register the two MFMAs just issued are still reading, and the
        // write-after-read interlock holds the VALU - and with it the in-order wave - until the MFMA is through)
        unsigned long long any_hit;
        asm("v_max3_f32 %1, %2, %3, %4\n\tv_max_f32 %1, %1, %5\n\tv_cmp_gt_f32 %0, %1, %6"
            : "=s"(any_hit), "+v"(m_scr)
            : "v"(a[4 * g]), "v"(a[4 * g + 1]), "v"(a[4 * g + 2]), "v"(a[4 * g + 3]), "v"(thr[rb]));
#ifdef APS_NO_EVENTS  // timing experiment: what the event code costs by merely being there
        asm volatile("" ::"s"(any_hit));
        if (false) {
#else
        if (__builtin_expect(any_hit != 0, 0)) {  // cold: the common case must be the fall-through (a taken branch
#endif
                                                  // per slice costs an instruction refetch the MFMAs cannot hide)
            const bool hit = m_scr > thr[rb];
            if (__any(hit && pj[rb] >= 0)) drain(rb);
            if (hit) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pv[rb][q] = a[4 * g + q];
                pj[rb] = t * kTNB + cb * 32 + 4 * h + 8 * g;
            }
        }
        if (sl == 7 && cb == kTNB / 32 - 1) {
            // once per tile: the fourth-best of the union of the two halves' sorted fours = max over i+j=3 of
            // min(a_i, b_j) (parked groups are not in the lists yet: the threshold is merely a little stale)
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2) {
                const float o0 = __shfl_xor(u0[r2], 32), o1 = __shfl_xor(u1[r2], 32), o2 = __shfl_xor(u2[r2], 32),
                            o3 = __shfl_xor(u3[r2], 32);
                const float mm = fmaxf(fmaxf(fminf(u0[r2], o2), fminf(u1[r2], o1)), fminf(u2[r2], o0));
                thr[r2] = fmaxf(thr[r2], fmaxf(fmaxf(u3[r2], o3), mm));
            }
        }
    };

    if (ntiles > 0) {
#pragma unroll
        for (int u = 0; u < kTNB / 32; ++u) issue_piece(0, 0, u);
    }
    // touch the resident operand here: otherwise the compiler's pending-load state for these registers
    // reaches the loop header and it drains vmcnt (DMA included) at their first use in EVERY iteration
#pragma unroll
    for (int s = 0; s < 8; ++s) asm volatile("" ::"v"(ah[0][s]), "v"(ah[1][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane read offsets: row c of a 32-row block, chunk (2s + h) ^ (c & 15)
    const int hx = (16 * h) ^ (16 * (c & 15));
    // constant operand of the extra k-step: k = 0,1,2 -> -c_i (held by the h == 0 half), everything else 0
    f16x8 aug_a;
#pragma unroll
    for (int e = 0; e < 8; ++e) aug_a[e] = (_Float16)0.f;
    if (h == 0) {
        aug_a[0] = (_Float16)(-ca[0]);
        aug_a[1] = (_Float16)(-ca[1]);
        aug_a[2] = (_Float16)(-ca[2]);
    }
    // k-slot 3: ||a|| of the lane's row (rounded up) against the column's rounding-loss norm
    f16x8 aug_a2 = aug_a;
    if (h == 0) {
        float unused;
        // (pooled list: the norms come from each row's own job - jb is only some job of the B group)
        const float* sq0 = (LIST && row_job) ? jobs[rjob[0]].sqA : jb.sqA;
        const float* sq1 = (LIST && row_job) ? jobs[rjob[1]].sqA : jb.sqA;
        const unsigned short n0 = f16_round_up(sqrtf(sq0[rowid[0]]) * 1.000001f, unused);
        const unsigned short n1 = f16_round_up(sqrtf(sq1[rowid[1]]) * 1.000001f, unused);
        aug_a[3] = __builtin_bit_cast(_Float16, n0);
        aug_a2[3] = __builtin_bit_cast(_Float16, n1);
    }
    // a second copy the compiler cannot see through: otherwise it folds the two initial MFMAs of a block into one
    // and chains the second accumulator's first MFMA behind it (a dependent issue, one MFMA latency per block)
    asm volatile("" : "+v"(aug_a2));
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Software pipeline.  (1) The search of block g-1 is cut into eight slices that sit BETWEEN the MFMA pairs of
    // block g in the same wave (two accumulator sets): VALU work issued next to one's own MFMAs is nearly free
    // up to ~3 operations per MFMA, whereas a wave that alternates a pure-MFMA phase with a pure-VALU phase finds
    // its VALU phase throttled to a third by the partner wave's MFMAs (scripts/probe/mfma_valu_coexec.hip: 1056
    // vs 1412 cycles per block pair).  (2) Operand reads run one k-step ahead, across block and tile boundaries:
    // the tile hand-over sits one block early - tile t+1 must have landed at the barrier between blocks 2 and 3 of
    // tile t, so that block 3 can already read ahead into it.  Tile t itself is still being read during block 3,
    // hence three LDS buffers: the DMA of tile t+1 is issued during blocks (t-1,3), (t,0), (t,1), (t,2) into the
    // buffer of tile t-2, which every wave left for good before it passed the hand-over of tile t-1.
    constexpr int kAhead = 3;  // an LDS round trip under load is longer than one k-step (two MFMAs + a slice)
    f16x8 bh[4], aug_q[2];
    if (ntiles > 0) {
        if (ntiles > 1 && !(ablate & 2)) issue_piece(1, 1, 0);
        aug_q[0] = *reinterpret_cast<const f16x8*>(&s_aug[0][c]);
#pragma unroll
        for (int s = 0; s < kAhead; ++s) bh[s] = *reinterpret_cast<const f16x8*>(lds + c * 256 + ((32 * s) ^ hx));
    }
#ifdef APS_MATCH_TIMING  // phase timing of one workgroup (make EXTRA=-DAPS_MATCH_TIMING, APS_MATCH_ABLATE=8)
    unsigned long long T_mf = 0, T_bar = 0;
    const unsigned long long T_c0 = __builtin_readcyclecounter(), T_w0 = wall_clock64();
    unsigned long long T_prev = T_c0, T_mark = T_c0;
#define APS_TICK(acc_)                                              \
    {                                                               \
        const unsigned long long n_ = __builtin_readcyclecounter(); \
        acc_ += n_ - T_prev;                                        \
        T_prev = n_;                                                \
    }
#else
#define APS_TICK(acc_)
#endif
    int b_cur = 0;  // t % 3
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles && !(ablate & 2);
        const bool more2 = t + 2 < ntiles && !(ablate & 2);
#ifdef APS_MATCH_TIMING
        if (t == 128) T_mark = __builtin_readcyclecounter();
#endif
        const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_nxt2 = b_nxt == 2 ? 0 : b_nxt + 1;
        const unsigned char* tile = lds + b_cur * kTileBytes + c * 256;
        const unsigned char* tile_n = lds + b_nxt * kTileBytes + c * 256;
        const int b_this = b_cur;
        b_cur = b_nxt;
        static_for<0, kTNB / 32>([&](auto CB) {
            constexpr int cb = decltype(CB)::value;
            constexpr int kLast = kTNB / 32 - 1;
            constexpr int par = cb & 1;
            APS_TICK(T_bar)
            // one DMA piece per block: pieces 1..3 of tile t+1 in blocks 0..2, piece 0 of tile t+2 in block 3
            if (cb < kLast ? more : more2) issue_piece(cb < kLast ? t + 1 : t + 2, cb < kLast ? b_nxt : b_nxt2,
                                                       cb < kLast ? cb + 1 : 0);
            const unsigned char* blk = tile + cb * 32 * 256;
            // first operands of the next block: same tile, or block 0 of the tile handed over one block ago
            const unsigned char* nblk = cb < kLast ? blk + 32 * 256 : tile_n;
            const uint4* naug = cb < kLast ? &s_aug[b_this][c + 32 * (cb + 1)] : &s_aug[b_nxt][c];
            const bool fetch = cb < kLast || t + 1 < ntiles;
            const int pt = cb ? t : t - 1;
            constexpr int pcb = cb ? cb - 1 : kLast;
            acc[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aug_q[par], aug_a, zero16, 0, 0, 0);
            acc[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aug_q[par], aug_a2, zero16, 0, 0, 0);
            static_for<0, 8>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const f16x8 xh = bh[s & 3];
                if (s + kAhead < 8) {
                    bh[(s + kAhead) & 3] = *reinterpret_cast<const f16x8*>(blk + ((32 * (s + kAhead)) ^ hx));
                } else if (fetch) {
                    bh[(s + kAhead) & 3] = *reinterpret_cast<const f16x8*>(nblk + ((32 * (s + kAhead - 8)) ^ hx));
                    if (s == 8 - kAhead) aug_q[par ^ 1] = *reinterpret_cast<const f16x8*>(naug);
                }
                acc[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, ah[0][s], acc[par][0], 0, 0, 0);
                acc[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, ah[1][s], acc[par][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // (before the first block the other set holds -inf, which never passes the screen)
                select_slice(std::integral_constant<int, par ^ 1>{}, pt, std::integral_constant<int, pcb>{}, S);
                __builtin_amdgcn_sched_barrier(0);
            });
            APS_TICK(T_mf)
            if (cb == kLast - 1) {
                // hand-over: this wave's DMA pieces of tile t+1 (and its b2) have landed; after the barrier that
                // holds for every wave, and every wave has left tile t-1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef APS_MATCH_TIMING
                if (!(ablate & 32))  // (bit 32 in timing builds: no hand-over barrier, no rescoring - racy, timing only)
#endif
                    __syncthreads();
                APS_TICK(T_bar)
            }
        });
    }
#ifdef APS_MATCH_TIMING
    const unsigned long long T_loop_end = __builtin_readcyclecounter();
    if ((ablate & 8) && blockIdx.x == 300 && lane == 0 && (wave == 0 || wave == 4)) {
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
        printf("wave %d: prologue %llu cycles (entry -> first block); steady state (tiles 128..end): %.0f cycles per block\n", wave,
               T_c0 - T_entry, ntiles > 128 ? (double)(c1 - T_mark) / ((ntiles - 128) * 4) : 0.0);
        printf("wave %d: %.3f GHz, %d blocks, cycles per block: total %.0f = mfma+selection %.0f + hand-over %.0f\n",
               wave, (double)(c1 - T_c0) / ((double)(w1 - T_w0) * 10.0), ntiles * (kTNB / 32),
               (double)(c1 - T_c0) / (ntiles * (kTNB / 32)), (double)T_mf / (ntiles * (kTNB / 32)),
               (double)T_bar / (ntiles * (kTNB / 32)));
    }
#endif
    if (ntiles > 0) {
        static_for<0, 8>([&](auto S) {
            select_slice(std::integral_constant<int, (kTNB / 32 - 1) & 1>{}, ntiles - 1,
                         std::integral_constant<int, kTNB / 32 - 1>{}, S);
        });
    }
    drain(0);
    drain(1);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const float p0 = __shfl_xor(u0[rb], 32), p1 = __shfl_xor(u1[rb], 32), p2 = __shfl_xor(u2[rb], 32),
                    p3 = __shfl_xor(u3[rb], 32);
        const int q0 = __shfl_xor(i0[rb], 32), q1 = __shfl_xor(i1[rb], 32), q2 = __shfl_xor(i2[rb], 32);
        top4_insert_max(p0, q0, u0[rb], u1[rb], u2[rb], u3[rb], i0[rb], i1[rb], i2[rb]);
        top4_insert_max(p1, q1, u0[rb], u1[rb], u2[rb], u3[rb], i0[rb], i1[rb], i2[rb]);
        top4_insert_max(p2, q2, u0[rb], u1[rb], u2[rb], u3[rb], i0[rb], i1[rb], i2[rb]);
        u3[rb] = fmaxf(u3[rb], p3);
    }
    // Exact rescoring in place (it used to be a separate, purely gather-bound launch): the h == 0 half holds the
    // merged lists of both owned rows; its lanes rescore row block 0 while the h == 1 lanes take over row block 1.
    {
        const int src = lane & 31;  // every lane takes part in the exchange (a masked-off source lane would read as 0)
        const int s0 = __shfl(i0[1], src), s1 = __shfl(i1[1], src), s2 = __shfl(i2[1], src);
        const float sb = __shfl(u3[1], src), sb2 = __shfl(u2[1], src), sb1 = __shfl(u1[1], src), sb0 = __shfl(u0[1], src);
        const int c0 = h ? s0 : i0[0], c1 = h ? s1 : i1[0], c2 = h ? s2 : i2[0];
        const float ub = h ? sb : u3[0], ub2 = h ? sb2 : u2[0], ub1 = h ? sb1 : u1[0], ub0 = h ? sb0 : u0[0];
        const int row = h ? rowid[1] : rowid[0];
        const int myjob = h ? rjob[1] : rjob[0];
        // (pooled list: the row's own job - same B set, its own A side and output slots)
        const MatchJob& jr = (LIST && row_job) ? jobs[myjob] : jb;
        if ((h ? rvalid[1] : rvalid[0]) && !(ablate & 32)) {  // (bit 32: timing experiment without the rescoring tail)
            // Rows that cannot pass the caller's ratio / threshold filter (matchFeaturesScratch.m:170-178) are dismissed
            // on the screened values alone: d1 >= L1 and d2 <= H2 hold for the exact f32 distances (see prune_bounds), so
            // L1 > r^2 H2 (or L1 > MatchThreshold) decides the filter's verdict without the exact evaluation - no gathers,
            // no fallback.  Only the filtered entry points enable this (the raw 2-NN API reports exact distances always).
            bool pruned = false;
            if (prune_r2 > 0.f && nB >= 2 && c0 >= 0 && c0 < nB && c1 >= 0 && c1 < nB) {
                float L1, H2;
                prune_bounds(jr, row, ub0, ub1, jb.augresB[0], jb.augresB[1], L1, H2);
                const float lo = L1 * (1.0f - 1e-5f) - 1e-30f;
                pruned = H2 >= 0.f && (lo > prune_r2 * H2 * (1.0f + 1e-5f) || lo > prune_thr * (1.0f + 1e-5f));
            }
            if (t3_idx) {  // top-3 mode (blocked global k-NN): three exact distances + the set's bound per row
                const float bnd = jr.sqA[row] - 2.0f * ub;
                rescore_row3(jr, myjob, row, c0, c1, c2, bnd, jb.augresB[0], jb.augresB[1], t3_idx, t3_d, t3_b, fb_list, fb_count);
            } else if (pruned) {
                const int64_t slot = jr.out_off + row;
                out_idx[slot] = 0u;  // "no match": the filter drops idx 0 (a row the reference's filter would drop too)
                out_d1[slot] = INFINITY;
                out_d2[slot] = INFINITY;
            } else {
                const float bnd = jr.sqA[row] - 2.0f * ub;    // approximate 4th-smallest distance (inf if < 4 columns)
                const float bnd3 = jr.sqA[row] - 2.0f * ub2;  // ... and the 3rd
                rescore_row(jr, myjob, row, c0, c1, c2, bnd3, bnd, jb.augresB[0], jb.augresB[1], out_idx, out_d1, out_d2, fb_list,
                            fb_count);
            }
        }
    }
#ifdef APS_MATCH_TIMING
    if ((ablate & 8) && blockIdx.x == 300 && lane == 0 && (wave == 0 || wave == 4))
        printf("wave %d: tail %llu cycles (last block -> exit)\n", wave, __builtin_readcyclecounter() - T_loop_end);
#endif
}

// ------------------------------------------------------------------------------------------------
// int8 screening pre-pass of the filtered entry points
// ------------------------------------------------------------------------------------------------
// The ratio / threshold filter (matchFeaturesScratch.m:170-178) keeps a few per cent of the rows of an overlapping pair and
// next to none of a non-overlapping one, and a row it drops needs no exact 2-NN at all - only a PROOF that it will be
// dropped.  v_mfma_i32_32x32x32_i8 runs at twice the f16 rate and accumulates exactly, so this pass streams every
// (A row, B column) product once on int8 copies (q8_desc_kernel) and keeps, per row, nothing but the two largest integer
// dot products D0 >= D1.  With a = qa invSA + ea and b_j = (qb_j + cB) invSB + eb_j (q8_desc_kernel),
//     a.b_j = invSA invSB (D_j + cB sum_k qa_k) + (invSA qa).eb_j + ea.b_j,   |error| <= E = dnA NB + (||a|| + dnA) DNB
// (NB >= every ||b_j||, DNB >= every ||eb_j||; sc D below stands for the bracketed main term), hence for the distances the
// filter will see
//     d1 >= a2 + min b2 - 2 (sc D0 + E) - delta =: L1        (every column lies below D0)
//     d2 <= a2 + max b2 - 2 (sc D1 - E) + delta =: H2        (two distinct columns lie at or above D1)
// and a row with L1 > r^2 H2 or L1 > MatchThreshold is dismissed here (idx 0, the filter's "no match").  The survivors
// are listed per job and go through match_cand_f16_kernel<LIST> - the exact machinery on a few per cent of the rows.
// The selection costs 20 VALU instructions per 32 x 64 block and has no branches: m = the largest of a lane's 16 columns
// of a block (v_max3), D1 = med3(D0, D1, m), D0 = max(D0, m).  Tracking block maxima instead of single values can only
// lower D1 (when a row's two best columns share a 16-column group), i.e. raise H2: still a bound.
// Layout: 128-B rows in LDS, DMA'd lane-linear; the 16-B chunk position is XORed with (row >> 1) & 7 on the DMA's
// source side and on the ds_read_b128 side, which puts each of ds_read_b128's four lane groups on 16 distinct bank
// quads.  Three buffers of 256 columns, the hand-over one block early, operand reads three k-steps ahead - the scheme of
// match_cand_f16_kernel.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int kQTN = 256;                 // B columns per LDS tile (8 column blocks per hand-over)
constexpr int kQBlk = kQTN / 32;
constexpr int kQTileBytes = kQTN * kDim;  // 32 KiB

struct ScreenSet {  // what the tail needs of a job's column set
    float inv_sb, cb, nb, dnb, b2min, b2max, msb;
    bool ok;
};

__device__ __forceinline__ ScreenSet screen_set(const MatchJob& jb) {
    ScreenSet q;
    const Q8Set qs = q8_set(jb.qstatB);
    q.inv_sb = qs.inv_sb;
    q.cb = qs.cb;
    q.msb = *jb.maxsqB;
    q.nb = sqrtf(q.msb) * 1.00001f;
    q.dnb = jb.qstatB[1];
    q.b2min = __uint_as_float(~__float_as_uint(jb.qstatB[2]));  // stored complemented (prep_desc_kernel)
    q.b2max = q.msb;
    // finite, non-degenerate data only (a NaN or an infinity anywhere in a set shows in its max ||x||^2)
    q.ok = qs.inv_sb > 0.f && q.msb < 1e37f && q.b2min >= 0.f;
    return q;
}

// The decision for one row from its two largest integer dots e0 >= e1 (kScreenNone = "no such column"): shared by the two
// MFMA shapes of the screening kernel.
constexpr int kScreenNone = -2147483647 - 1;

// Does this job take the exact integer codes (q8_desc_rows)?  Every row of both sets has one.
// ... and the column set's divisors lie within 2 % of each other: a column with an odd divisor (a descriptor concentrated in one or
// two bins saturates at 255: t = 255 .. 360 instead of ~512) shares its tile with 250 ordinary columns, whose bounds it would
// inflate beyond any real best similarity - no row of the job could be dismissed, where the rounded codes lose nothing.
__device__ __forceinline__ bool screen_exact(const MatchJob& jb) {
    if ((jb.xstatA[2] | jb.xstatB[2]) != 0u) return false;
    const float tmax = __uint_as_float(jb.xstatB[0]), tmin = __uint_as_float(~jb.xstatB[1]);
    return tmin > 0.f && tmax <= 1.02f * tmin;
}

template <bool BOUNDS>
__device__ __forceinline__ void screen_tail(const MatchJob& jb, int job, int row, int e0, int e1, int lane,
                                            uint32_t* __restrict__ out_idx, float* __restrict__ out_d1, float* __restrict__ out_d2,
                                            uint32_t* __restrict__ surv_list, unsigned int* __restrict__ surv_count, float prune_r2,
                                            float prune_thr, float* __restrict__ bounds_out, bool exact = false, float s_best = 0.f,
                                            float s_second = 0.f) {
    constexpr int kNone = kScreenNone;
    const int nA = jb.nA, nB = jb.nB;
    bool survive = false;
    if (BOUNDS) {
        if (row < nA) {
        const ScreenSet q = screen_set(jb);
        const float inv_sa = jb.invsA[row];
        float L1f = -INFINITY, H1f = INFINITY, H2f = INFINITY;
        if (q.ok && inv_sa > 0.f && nB >= 1 && e0 != kNone) {
            const double a2 = (double)jb.sqA[row];
            const double na = sqrt(a2) * 1.00001, dna = (double)jb.dnqA[row];
            const double E = (dna * (double)q.nb + (na + dna) * (double)q.dnb) * 1.0001 + 1e-7 * (na * (double)q.nb);
            const double sc = (double)inv_sa * (double)q.inv_sb;
            const double off = (double)q.cb * (double)jb.sumqA[row];
            const double delta = 1.52587890625e-05 * (a2 + (double)q.msb + 2.0 * na * (double)q.nb) + 1e-37;
            L1f = __double2float_rd(a2 + (double)q.b2min - 2.0 * (sc * ((double)e0 + off) + E) - delta);
            H1f = __double2float_ru(a2 + (double)q.b2max - 2.0 * (sc * ((double)e0 + off) - E) + delta);
            if (nB >= 2 && e1 != kNone) H2f = __double2float_ru(a2 + (double)q.b2max - 2.0 * (sc * ((double)e1 + off) - E) + delta);
        }
        float* bo = bounds_out + (size_t)(jb.out_off + row) * 3;
        bo[0] = L1f;
        bo[1] = H1f;
        bo[2] = H2f;
        }
    } else if (row < nA && exact) {
        // Exact integer codes (q8_desc_rows): u_i.u_j = acc + 128 sum p_i + 128^3 =: I is an exact integer >= 0 and the real dot
        // product of the two f32 rows is I / (t_i t_j) within a factor 1 +- 2^-23.  The columns are sorted by their divisor, and
        // every 256-column tile knows its range [tlo, thi]: the kernel tracked, over the tiles, s_best = max I / tlo (an upper
        // bound of the largest I_j / t_j) and s_second = the second largest I / thi over DISTINCT column groups (a lower
        // bound of the second largest I_j / t_j), in f32 with the reciprocals rounded outwards; L1, H2 follow as in the
        // general form below with E = 0.
        bool pruned = false;
        const float msb = *jb.maxsqB;
        const float b2min = __uint_as_float(~__float_as_uint(jb.qstatB[2]));
        const bool have2 = s_second > -1e9f;
        if (msb < 1e37f && b2min >= 0.f && nB >= 2 && have2 && s_best < 1e30f) {
            const double a2 = (double)jb.sqA[row];
            const double na = sqrt(a2) * 1.00001, nb = sqrt((double)msb) * 1.00001;
            const double ti = (double)jb.ttA[row];
            const double s_hi = fmax((double)s_best, 0.0) / ti * (1.0 + 6e-7);
            const double s_lo = fmax((double)s_second, 0.0) / ti * (1.0 - 6e-7);
            const double delta = 1.52587890625e-05 * (a2 + (double)msb + 2.0 * na * nb) + 1e-37;
            const double L1 = a2 + (double)b2min - 2.0 * s_hi - delta;
            const double H2 = a2 + (double)msb - 2.0 * s_lo + delta;
            const double lo = L1 * (1.0 - 1e-5) - 1e-30;
            pruned = ti > 0.0 && H2 >= 0.0 && (lo > (double)prune_r2 * H2 * (1.0 + 1e-5) || lo > (double)prune_thr * (1.0 + 1e-5));
        }
        if (pruned) {
            const int64_t slot = jb.out_off + row;
            out_idx[slot] = 0u;
            out_d1[slot] = INFINITY;
            out_d2[slot] = INFINITY;
        } else {
            survive = true;
            // The exact list pass (match_list_i8_kernel) will name the columns that can be the row's best or second best by the
            // canonical f32 distance: two distinct columns have I_j / t_j >= s_second, and a column of tile T can reach that
            // only with I_j >= s_second tlo_T.  The list pass forms that threshold per tile (with a margin of 2e-4 for the two
            // 2^-23 factors and the rounding of the canonical distance itself: delta above is 2^-16 of the magnitudes); what
            // travels in the row's d1 slot is s_second (-inf: every column is a candidate).
            out_d1[jb.out_off + row] = have2 && nB >= 2 ? s_second : -INFINITY;
        }
    } else if (row < nA) {
        const ScreenSet q = screen_set(jb);
        bool pruned = false;
        const float inv_sa = jb.invsA[row];
        if (q.ok && inv_sa > 0.f && nB >= 2 && e1 != kNone) {
            const double a2 = (double)jb.sqA[row];
            const double na = sqrt(a2) * 1.00001, dna = (double)jb.dnqA[row];
            const double E = (dna * (double)q.nb + (na + dna) * (double)q.dnb) * 1.0001 + 1e-7 * (na * (double)q.nb);
            const double sc = (double)inv_sa * (double)q.inv_sb;
            const double off = (double)q.cb * (double)jb.sumqA[row];  // the column code's offset, put back per row
            const double delta = 1.52587890625e-05 * (a2 + (double)q.msb + 2.0 * na * (double)q.nb) + 1e-37;
            const double L1 = a2 + (double)q.b2min - 2.0 * (sc * ((double)e0 + off) + E) - delta;
            const double H2 = a2 + (double)q.b2max - 2.0 * (sc * ((double)e1 + off) - E) + delta;
            const double lo = L1 * (1.0 - 1e-5) - 1e-30;
            pruned = H2 >= 0.0 && (lo > (double)prune_r2 * H2 * (1.0 + 1e-5) || lo > (double)prune_thr * (1.0 + 1e-5));
        }
        if (pruned) {
            const int64_t slot = jb.out_off + row;
            out_idx[slot] = 0u;  // "no match": the filter drops idx 0 (a row the reference's filter drops too)
            out_d1[slot] = INFINITY;
            out_d2[slot] = INFINITY;
        } else {
            survive = true;
        }
    }
    // survivors: one list segment per job (at the job's first output slot), one counter update per wave
    const unsigned long long sm = __ballot(survive);
    if (sm) {
        unsigned int base = 0;
        const int leader = __ffsll((long long)sm) - 1;
        if (lane == leader) base = atomicAdd(&surv_count[job], (unsigned int)__popcll(sm));
        base = __shfl(base, leader);
        if (survive) surv_list[jb.out_off + base + __popcll(sm & ((1ull << lane) - 1ull))] = (uint32_t)row;
    }
}

// BOUNDS: the pooled matcher's form (screened_global_top3) - a separate instantiation, so that profiles keep the two
// apart.
template <bool BOUNDS>
__global__ __launch_bounds__(512) void match_screen_i8_kernel(const MatchJob* __restrict__ jobs,
                                                                 const WgJob* __restrict__ wgs, int n_wg,
                                                                 uint32_t* __restrict__ out_idx, float* __restrict__ out_d1,
                                                                 float* __restrict__ out_d2, uint32_t* __restrict__ surv_list,
                                                                 unsigned int* __restrict__ surv_count, float prune_r2,
                                                                 float prune_thr, float* __restrict__ bounds_out) {
    // bounds_out != nullptr (the pooled matcher, screened_global_top3): nothing is decided here; per (row, job) slot the
    // three bounds L1 <= d1, H1 >= d1, H2 >= d2 on the row's two smallest distances in this job's column set are written
    // (rounded outwards; -inf / +inf where nothing can be said) and a later pass combines them over a row's jobs.
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * kQTileBytes];  // [buf][256][128 B], tile t in buf t % 3
    int wg = blockIdx.x;
    {  // XCD-aware order, as in match_cand_f16_kernel
        const int q = n_wg / 8, r = n_wg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    // The kernel claims the whole vector register file of its SIMDs (2 waves x 256 registers) although it needs 169:
    // waves of OTHER kernels that shared a SIMD with v_mfma_i32_32x32x32_i8 waves came back with different results
    // (SIFT's refine / orientation / descriptor kernels, from another stream, when feature extraction and matching
    // overlap; f16 MFMA, VALU, LDS or LDS-DMA neighbours leave them alone) - measured with scripts/probe/probe_overlap_race3.py,
    // DESIGN.md section 5.  With nothing co-resident the extraction is bit-identical again; this kernel's own results were
    // never affected.
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const WgJob w = wgs[wg];
    const MatchJob jb = jobs[w.job];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31;
    const int h = lane >> 5;
    const int nA = jb.nA, nB = jb.nB;
    const int row0 = w.row0 + wave * 64 + c;  // this lane owns rows row0 and row0 + 32 (both halves h see them)

    i32x4 aq[2][4];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int arow = min(row0 + 32 * rb, nA - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s)
            aq[rb][s] = *reinterpret_cast<const i32x4*>(jb.AQ + (size_t)arow * kDim + 32 * s + 16 * h);
    }
    constexpr int kNone = -2147483647 - 1;
    int d0[2] = {kNone, kNone}, d1[2] = {kNone, kNone};

    const int ntiles = (nB + kQTN - 1) / kQTN;
    // DMA pieces: 32 per tile, 1 KiB = 8 LDS rows each; wave w issues pieces 4w .. 4w+3.
    // lane -> LDS row 8*piece + lane/8, chunk position lane&7, which must hold source chunk pos ^ ((row >> 1) & 7)
    const int dma_sub = lane >> 3, dma_pos = lane & 7;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue_piece = [&](int t, int buf, int u) {
        const int piece = wave * 4 + u;
        const int lrow = 8 * piece + dma_sub;
        const int brow = min(t * kQTN + lrow, nB - 1);
        const signed char* src = jb.BQ + (size_t)brow * kDim + ((dma_pos ^ ((lrow >> 1) & 7)) << 4);
        const uint32_t dst = lds_base + buf * kQTileBytes + piece * 1024;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
    };
    i32x16 acc[2][2];  // [block parity][owned row]
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = acc[0][1][e] = acc[1][0][e] = acc[1][1][e] = kNone;

    // One quarter of the search of a finished block: owned row rb = q >> 1, columns 8 (q & 1) .. + 7 of the lane's 16.
    // acc[par][rb][r] <-> B column (r&3) + 8*(r>>2) + 4*h of the block.  `limit` (ragged last tile only) = number of
    // valid columns counted from the block's first.
    int m_run = kNone;
    auto fold_quarter = [&](auto PAR, auto Q, auto MASK, int limit) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value, q = decltype(Q)::value;
        constexpr bool mask = decltype(MASK)::value;
        constexpr int rb = q >> 1, half = q & 1;
        const i32x16& a = acc[par][rb];
        int v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = 8 * half + e;
            v[e] = a[r];
            if (mask) v[e] = ((r & 3) + 8 * (r >> 2) + 4 * h) < limit ? v[e] : kNone;
        }
        // (volatile asm: left to itself the compiler sinks the whole search of a tile's eight blocks behind the tile's last
        // MFMA - nothing needs the result earlier - and then has eight blocks' accumulators alive at once)
        int m = m_run;
        if (half)
            asm volatile("v_max3_i32 %0, %0, %1, %2\n\tv_max3_i32 %0, %0, %3, %4\n\tv_max3_i32 %0, %0, %5, %6\n\tv_max3_i32 %0, %0, %7, %8"
                         : "+v"(m)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        else
            asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %0, %0, %6, %7\n\tv_max_i32 %0, %0, %8"
                         : "=&v"(m)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        m_run = m;
        if (half) {
            // the runner-up of {d0, d1, m} given d0 >= d1, then the new best
            int t0 = d0[rb], t1 = d1[rb];
            asm volatile("v_med3_i32 %0, %1, %0, %2\n\tv_max_i32 %1, %1, %2" : "+v"(t1), "+v"(t0) : "v"(m));
            d0[rb] = t0;
            d1[rb] = t1;
        }
    };

    if (ntiles > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) issue_piece(0, 0, u);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" ::"v"(aq[0][s]), "v"(aq[1][s]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane read offsets: row c of a 32-row block, chunk (2s + h) ^ ((c >> 1) & 7)
    const int hx = (16 * h) ^ (16 * ((c >> 1) & 7));
    const i32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int kAhead = 3;
    i32x4 bq[4];
    if (ntiles > 0) {
        if (ntiles > 1) issue_piece(1, 1, 0);
#pragma unroll
        for (int s = 0; s < kAhead; ++s) bq[s] = *reinterpret_cast<const i32x4*>(lds + c * kDim + ((32 * s) ^ hx));
    }
    int b_cur = 0;  // t % 3
    auto run_tile = [&](int t, auto MASK) __attribute__((always_inline)) {
        constexpr bool mask = decltype(MASK)::value;
        const bool more = t + 1 < ntiles;
        const bool more2 = t + 2 < ntiles;
        const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_nxt2 = b_nxt == 2 ? 0 : b_nxt + 1;
        const unsigned char* tile = lds + b_cur * kQTileBytes + c * kDim;
        const unsigned char* tile_n = lds + b_nxt * kQTileBytes + c * kDim;
        b_cur = b_nxt;
        static_for<0, kQBlk>([&](auto CB) {
            constexpr int cb = decltype(CB)::value;
            constexpr int kLast = kQBlk - 1;
            constexpr int par = cb & 1;
            // DMA: pieces 1..3 of tile t+1 in blocks 0, 2, 4; piece 0 of tile t+2 in the last block (after the hand-over,
            // into the buffer tile t-1 has just left for good)
            if (cb == 0 || cb == 2 || cb == 4) {
                if (more) issue_piece(t + 1, b_nxt, cb / 2 + 1);
            } else if (cb == kLast) {
                if (more2) issue_piece(t + 2, b_nxt2, 0);
            }
            const unsigned char* blk = tile + cb * 32 * kDim;
            const unsigned char* nblk = cb < kLast ? blk + 32 * kDim : tile_n;
            const bool fetch = cb < kLast || more;
            // the block being searched is the previous one: (t, cb - 1), or the last block of tile t - 1 (never ragged)
            const int limit = mask ? nB - (t * kQTN + (cb - 1) * 32) : 0;
            static_for<0, 4>([&](auto S) {
                constexpr int s = decltype(S)::value;
                const i32x4 xq = bq[s & 3];
                if (s + kAhead < 4) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(blk + ((32 * (s + kAhead)) ^ hx));
                } else if (fetch) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(nblk + ((32 * (s + kAhead - 4)) ^ hx));
                }
                if (s == 0) {
                    acc[par][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xq, aq[0][0], zero16, 0, 0, 0);
                    acc[par][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xq, aq[1][0], zero16, 0, 0, 0);
                } else {
                    acc[par][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xq, aq[0][s], acc[par][0], 0, 0, 0);
                    acc[par][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xq, aq[1][s], acc[par][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (mask && cb > 0)
                    fold_quarter(std::integral_constant<int, par ^ 1>{}, S, std::true_type{}, limit);
                else
                    fold_quarter(std::integral_constant<int, par ^ 1>{}, S, std::false_type{}, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (cb == kLast - 1) {
                // hand-over: this wave's DMA pieces of tile t+1 have landed; after the barrier that holds for every
                // wave, and every wave has left tile t-1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        });
    };
    // full tiles run unmasked; a ragged last tile (columns >= nB: the DMA re-reads the last row for them) runs the masked copy
    const int nfull = nB / kQTN;
    for (int t = 0; t < nfull; ++t) run_tile(t, std::false_type{});
    if (nfull < ntiles) {
        run_tile(nfull, std::true_type{});
        const int limit = nB - (nfull * kQTN + (kQBlk - 1) * 32);
        static_for<0, 4>([&](auto S) { fold_quarter(std::integral_constant<int, (kQBlk - 1) & 1>{}, S, std::true_type{}, limit); });
    } else if (ntiles > 0) {
        static_for<0, 4>([&](auto S) { fold_quarter(std::integral_constant<int, (kQBlk - 1) & 1>{}, S, std::false_type{}, 0); });
    }
    // the two halves of a wave saw disjoint columns of the same rows
    int D0[2], D1[2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int p0 = __shfl_xor(d0[rb], 32), p1 = __shfl_xor(d1[rb], 32);
        D0[rb] = max(d0[rb], p0);
        D1[rb] = max(min(d0[rb], p0), max(d1[rb], p1));
    }
    // the h == 0 half decides row block 0, the h == 1 half row block 1
    const int row = row0 + 32 * h;
    const int e0 = h ? D0[1] : D0[0], e1 = h ? D1[1] : D1[0];
    screen_tail<BOUNDS>(jb, w.job, row, e0, e1, lane, out_idx, out_d1, out_d2, surv_list, surv_count, prune_r2, prune_thr, bounds_out);
}

// The same pass on v_mfma_i32_16x16x64_i8 (round 4, the default).  Both shapes do the same multiply-adds per cycle on
// paper, but the chip is power-limited under a dense int8 stream and the clock it holds depends on the shape:
// scripts/probe/mfma_i8_shapes.hip on random operands reads 3.44 POP/s at 1.71 GHz for 32x32x32 and 3.99 POP/s at
// 1.98 GHz for 16x16x64 (4.96 POP/s at 2.39 GHz for both on all-zero operands) - the 32x32x32 kernel above, at 3.1 POP/s
// on real codes, already sat at 0.9 of what its shape can be fed.
// Mapping: a wave owns 64 rows as four groups of 16; operand 2 of the MFMA is a row group (lane l: row l & 15 of the group,
// k bytes 16 (l >> 4) .. + 15 of the 64-byte k-step), operand 1 a 16-column sub-block of the B tile (lane l: column l & 15,
// the same k bytes), the result D[column 4 (l >> 4) + r][row l & 15] in four registers.  A 32-column block = two
// sub-blocks x two k-steps = four ds_read_b128 per lane (the LDS image and its XOR swizzle are unchanged and stay
// conflict-free for this read pattern: row c, chunk (4 ks + kq) ^ ((c >> 1) & 7)), each feeding four MFMAs of 16 cycles
// - the cadence of the 32x32x32 loop, so DMA, hand-over and read-ahead carry over.  Selection: see fold_group; the four
// lane quarters of a row are merged at the end.
template <bool BOUNDS>
__global__ __launch_bounds__(512) void match_screen_i8x16_kernel(const MatchJob* __restrict__ jobs,
                                                                    const WgJob* __restrict__ wgs, int n_wg,
                                                                    uint32_t* __restrict__ out_idx, float* __restrict__ out_d1,
                                                                    float* __restrict__ out_d2, uint32_t* __restrict__ surv_list,
                                                                    unsigned int* __restrict__ surv_count, float prune_r2,
                                                                    float prune_thr, float* __restrict__ bounds_out, unsigned int* __restrict__ exact_flag) {
    // [buf][256][128 B], tile t in buf t % 3; then [buf][256] accumulator start values (round 6: cinB, 1 KiB per tile)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * kQTileBytes + 3 * 1024];
    int wg = blockIdx.x;
    {  // XCD-aware order, as in match_cand_f16_kernel
        const int q = n_wg / 8, r = n_wg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    // (the whole register file of the SIMD is claimed: see match_screen_i8_kernel and DESIGN.md section 5)
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const WgJob w = wgs[wg];
    const MatchJob jb = jobs[w.job];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int kq = lane >> 4;
    const int nA = jb.nA, nB = jb.nB;
    const int rowb = w.row0 + wave * 64;  // this wave's rows: rowb + 16 g + c, every lane quarter kq sees them
    // Round 6: where every row of both sets has an exact integer code (q8_desc_rows) the pass runs on those - same loop, other
    // operands: the code bytes AX / BX and, as the C operand of each column block's first MFMA, the columns' 128 sum p_j
    // (cinB; zeros for the general codes).  The pooled matcher's bounds pass keeps the general codes.
    const bool exact = !BOUNDS && screen_exact(jb);
    if (!BOUNDS && exact && tid == 0) exact_flag[w.job] = 1u;  // (diagnostics: aps_match_screen_exact_jobs)
    const signed char* const opA = exact ? jb.AX : jb.AQ;
    const signed char* const opB = exact ? jb.BXs : jb.BQ;  // (exact: the columns in ascending order of their divisor)
    const int* const cin_src = jb.cinBs;
    // experiment switches (APS_SCR_VARIANT): 1 = static priority for waves 4-7, 2 / 4 = waves 4-7 sleep 64 / 128 cycles after
    // every hand-over barrier (a stagger between the two waves of a SIMD), 8 = the odd waves instead of waves 4-7
#ifdef APS_MATCH_TIMING  // (timing builds only, like the ablation bits)
    const int variant = __builtin_amdgcn_readfirstlane(g_scr_variant);
    const bool late_half = (variant & 8) ? (wave & 1) != 0 : wave >= 4;
    if ((variant & 1) && late_half) __builtin_amdgcn_s_setprio(1);
#endif

    i32x4 aq[4][2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int arow = min(rowb + 16 * g + c, nA - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aq[g][ks] = *reinterpret_cast<const i32x4*>(opA + (size_t)arow * kDim + 64 * ks + 16 * kq);
    }
    constexpr int kNone = kScreenNone;
    int d0[4] = {kNone, kNone, kNone, kNone}, d1[4] = {kNone, kNone, kNone, kNone};

    const int ntiles = (nB + kQTN - 1) / kQTN;
    // DMA pieces: as in match_screen_i8_kernel
    const int dma_sub = lane >> 3, dma_pos = lane & 7;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue_piece = [&](int t, int buf, int u) {
        const int piece = wave * 4 + u;
        const int lrow = 8 * piece + dma_sub;
        const int brow = min(t * kQTN + lrow, nB - 1);
        const signed char* src = opB + (size_t)brow * kDim + ((dma_pos ^ ((lrow >> 1) & 7)) << 4);
        const uint32_t dst = lds_base + buf * kQTileBytes + piece * 1024;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
    };
    // The accumulators' start values of tile t (cinB, one int per column: 1 KiB per tile) travel with piece 0 of the tile,
    // issued by wave 0; entries past the set's padded length are clamped to its last four (columns >= nB: masked in the fold).
    // General codes: the three 1-KiB areas are zero-filled once, below.
    const int ncin = jb.ncinB;
    auto issue_cin = [&](int t, int buf) {
        if (exact && wave == 0) {
            const int col = min(t * kQTN + 4 * lane, ncin - 4);
            const int* src = cin_src + col;
            const uint32_t dst = lds_base + 3 * kQTileBytes + buf * 1024;
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(dst)
                : "memory");
        }
    };
    if (!exact && tid < 192) *reinterpret_cast<i32x4*>(lds + 3 * kQTileBytes + 16 * tid) = i32x4{0, 0, 0, 0};
    i32x4 acc[2][2][4];  // [block parity][sub-block][row group]
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[p][u][g][e] = kNone;

    // The search of a finished block for row group g: the lane's 8 columns, 16 u + 4 kq + r of the block, are folded into the
    // running maximum of the lane's columns of the TILE (4 v_max3); the tile's last block (LAST) then updates the two best:
    // D1 = med3(D0, D1, m), D0 = max(D0, m).  A group is thus the 64 columns a lane holds of a 256-column tile: D1 is the
    // second largest GROUP maximum, which can only be lower than the second largest dot (when a row's two best columns fall
    // into one group of one lane: 64 of ~20 000 columns) - still a bound, and 17 instead of 24 VALU per block, which matters
    // here: a 16x16x64 MFMA holds the SIMD's vector issue for half of its 16 cycles, twice the share of the 32x32x32 form.
    // `limit` (ragged last tile only) = number of valid columns counted from the block's first.
    int m_run[4] = {kNone, kNone, kNone, kNone};
    // Exact codes: the columns come sorted by their divisor, and every kSeg tiles the integer best two (d0, d1) of the SEGMENT
    // are turned into bounds on I / t_j with the segment's divisor range [tlo of its first tile, thi of its last] and join the
    // row's best two in f32: s0 = max I / tlo, (f0, f1) = the two largest I / thi; then the integers start again.  (Per TILE
    // the same fold cost 40 vector instructions per tile and wave: the screen ran 64.2 against 60.9 ms; a segment of eight
    // tiles still spans only a tenth of the set's divisor range: 2.0x % of the rows survive either way.)
    constexpr int kSeg = 8;
    int seg_first = 0;  // first tile of the segment being collected (wave-uniform)
    float s0[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, f0[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY},
          f1[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int rc[4] = {0, 0, 0, 0};  // 128 sum p_i + 128^3 of the lane's four rows
    if (exact) {
#pragma unroll
        for (int g = 0; g < 4; ++g) rc[g] = jb.cinA[min(rowb + 16 * g + c, nA - 1)] + 2097152;
    }
    auto seg_fold = [&](int t_first, int t_last) __attribute__((always_inline)) {  // tiles t_first .. t_last are folded
        const float rlo = jb.tscB[t_first].x, rhi = jb.tscB[t_last].y;  // (wave-uniform)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int a0 = d0[g], a1 = d1[g];
            const float i0 = (float)(a0 + rc[g]), i1 = (float)(a1 + rc[g]);  // (I < 2^24: exact)
            const float hi = a0 != kNone ? i0 * rlo : -INFINITY;
            const float l0 = a0 != kNone ? i0 * rhi : -INFINITY, l1 = a1 != kNone ? i1 * rhi : -INFINITY;
            s0[g] = fmaxf(s0[g], hi);
            const float n0 = fmaxf(f0[g], l0);
            f1[g] = fmaxf(fminf(f0[g], l0), fmaxf(f1[g], l1));
            f0[g] = n0;
            d0[g] = d1[g] = kNone;
        }
    };
    auto fold_group = [&](auto PAR, auto G, auto MASK, int limit, auto FIRST, auto LAST) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value, g = decltype(G)::value;
        constexpr bool mask = decltype(MASK)::value, first = decltype(FIRST)::value, last = decltype(LAST)::value;
        int v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = acc[par][e >> 2][g][e & 3];
            if (mask) v[e] = (16 * (e >> 2) + 4 * kq + (e & 3)) < limit ? v[e] : kNone;
        }
        // (volatile asm: see match_screen_i8_kernel)
        int m = m_run[g];
        if (first)
            asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %0, %0, %6, %7\n\tv_max_i32 %0, %0, %8"
                         : "=&v"(m)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        else
            asm volatile("v_max3_i32 %0, %0, %1, %2\n\tv_max3_i32 %0, %0, %3, %4\n\tv_max3_i32 %0, %0, %5, %6\n\tv_max3_i32 %0, %0, %7, %8"
                         : "+v"(m)
                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        m_run[g] = m;
        if (last) {
            int t0 = d0[g], t1 = d1[g];
            asm volatile("v_med3_i32 %0, %1, %0, %2\n\tv_max_i32 %1, %1, %2" : "+v"(t1), "+v"(t0) : "v"(m));
            d0[g] = t0;
            d1[g] = t1;
        }
    };

    if (ntiles > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) issue_piece(0, 0, u);
        issue_cin(0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(aq[g][0]), "v"(aq[g][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane read offsets: slot s of a block = sub-block s >> 1, k-step s & 1: row 16 (s >> 1) + c, chunk (4 (s & 1) + kq)
    // ^ ((c >> 1) & 7)  (the sub-block's 16 rows leave (row >> 1) & 7 alone)
    const int hx = (16 * kq) ^ (16 * ((c >> 1) & 7));
    constexpr int kAhead = 3;
    auto slot_off = [&](int s) __attribute__((always_inline)) { return (s >> 1) * 16 * kDim + ((64 * (s & 1)) ^ hx); };
    i32x4 bq[4];
    // cin[u]: the start values of the lane's four columns 16 u + 4 kq + r of the NEXT block to start (read one block ahead)
    i32x4 cin[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned char* const cin_lds = lds + 3 * kQTileBytes + 16 * kq;
    if (ntiles > 0) {
        if (ntiles > 1) {
            issue_piece(1, 1, 0);
            issue_cin(1, 1);
        }
#pragma unroll
        for (int s = 0; s < kAhead; ++s) bq[s] = *reinterpret_cast<const i32x4*>(lds + c * kDim + slot_off(s));
        cin[0] = *reinterpret_cast<const i32x4*>(cin_lds);
        cin[1] = *reinterpret_cast<const i32x4*>(cin_lds + 64);
    }
    int b_cur = 0;  // t % 3
    auto run_tile = [&](int t, auto MASK) __attribute__((always_inline)) {
        constexpr bool mask = decltype(MASK)::value;
        const bool more = t + 1 < ntiles;
        const bool more2 = t + 2 < ntiles;
        const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_nxt2 = b_nxt == 2 ? 0 : b_nxt + 1;
        const unsigned char* tile = lds + b_cur * kQTileBytes + c * kDim;
        const unsigned char* tile_n = lds + b_nxt * kQTileBytes + c * kDim;
        const unsigned char* cin_t = cin_lds + b_cur * 1024;
        const unsigned char* cin_n = cin_lds + b_nxt * 1024;
        b_cur = b_nxt;

        static_for<0, kQBlk>([&](auto CB) {
            constexpr int cb = decltype(CB)::value;
            constexpr int kLast = kQBlk - 1;
            constexpr int par = cb & 1;
            if (cb == 0 || cb == 2 || cb == 4) {
                if (more) issue_piece(t + 1, b_nxt, cb / 2 + 1);
            } else if (cb == kLast) {
                if (more2) {
                    issue_piece(t + 2, b_nxt2, 0);
                    issue_cin(t + 2, b_nxt2);
                }
            }
            const unsigned char* blk = tile + cb * 32 * kDim;
            const unsigned char* nblk = cb < kLast ? blk + 32 * kDim : tile_n;
            const bool fetch = cb < kLast || more;
            const int limit = mask ? nB - (t * kQTN + (cb - 1) * 32) : 0;
            static_for<0, 4>([&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int u = s >> 1, ks = s & 1;
                const i32x4 xq = bq[s & 3];
                if (s + kAhead < 4) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(blk + slot_off(s + kAhead));
                } else if (fetch) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(nblk + slot_off(s + kAhead - 4));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[par][u][g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xq, aq[g][ks], ks == 0 ? cin[u] : acc[par][u][g], 0, 0, 0);
                // sub-block u's start values are consumed: fetch the next block's (the same tile's, or the next tile's first)
                if (ks == 1) {
                    if (cb < kLast)
                        cin[u] = *reinterpret_cast<const i32x4*>(cin_t + (32 * (cb + 1) + 16 * u) * 4);
                    else if (more)
                        cin[u] = *reinterpret_cast<const i32x4*>(cin_n + (16 * u) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                // the block being searched is the previous one: (t, cb - 1), or the last block of tile t - 1 (never ragged)
                constexpr bool first = cb == 1, last = cb == 0;
                if (mask && cb > 0)
                    fold_group(std::integral_constant<int, par ^ 1>{}, S, std::true_type{}, limit, std::integral_constant<bool, first>{},
                               std::integral_constant<bool, last>{});
                else
                    fold_group(std::integral_constant<int, par ^ 1>{}, S, std::false_type{}, 0, std::integral_constant<bool, first>{},
                               std::integral_constant<bool, last>{});
                __builtin_amdgcn_sched_barrier(0);
            });
            // (block 0 of this tile has folded the LAST block of tile t - 1: where that completes a segment, the segment joins)
            // Segments: the FIRST and the LAST tile alone (where columns with an odd divisor end up: they then loosen the bounds of
            // 256 columns, not of a segment's 2048), kSeg tiles each in between.
            if (cb == 0 && exact && t > 0 && (t == 1 || t - seg_first == kSeg || t == ntiles - 1)) {
                seg_fold(seg_first, t - 1);
                seg_first = t;
            }
            if (cb == kLast - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#ifdef APS_MATCH_TIMING
                if ((variant & 6) && late_half) {
                    if (variant & 2) __builtin_amdgcn_s_sleep(1);
                    if (variant & 4) __builtin_amdgcn_s_sleep(2);
                }
#endif
            }
        });
    };
    const int nfull = nB / kQTN;
    for (int t = 0; t < nfull; ++t) run_tile(t, std::false_type{});
    if (nfull < ntiles) {
        run_tile(nfull, std::true_type{});
        const int limit = nB - (nfull * kQTN + (kQBlk - 1) * 32);
        static_for<0, 4>([&](auto S) {
            fold_group(std::integral_constant<int, (kQBlk - 1) & 1>{}, S, std::true_type{}, limit, std::false_type{}, std::true_type{});
        });
    } else if (ntiles > 0) {
        static_for<0, 4>([&](auto S) {
            fold_group(std::integral_constant<int, (kQBlk - 1) & 1>{}, S, std::false_type{}, 0, std::false_type{}, std::true_type{});
        });
    }
    if (exact && ntiles > 0) seg_fold(seg_first, ntiles - 1);  // the last segment
    // the four lane quarters of a wave saw disjoint columns of the same rows; quarter kq then decides row group kq
    int e0 = kNone, e1 = kNone;
    float s_best = -INFINITY, s_second = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int a0 = d0[g], a1 = d1[g];
        float b0 = s0[g], c0 = f0[g], c1 = f1[g];
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const int p0 = __shfl_xor(a0, off), p1 = __shfl_xor(a1, off);
            const int n0 = max(a0, p0);
            a1 = max(min(a0, p0), max(a1, p1));
            a0 = n0;
            b0 = fmaxf(b0, __shfl_xor(b0, off));
            const float q0 = __shfl_xor(c0, off), q1 = __shfl_xor(c1, off);
            const float m0 = fmaxf(c0, q0);
            c1 = fmaxf(fminf(c0, q0), fmaxf(c1, q1));
            c0 = m0;
        }
        if (g == kq) {
            e0 = a0;
            e1 = a1;
            s_best = b0;
            s_second = c1;
        }
    }
    const int row = rowb + lane;  // = rowb + 16 kq + c
    screen_tail<BOUNDS>(jb, w.job, row, e0, e1, lane, out_idx, out_d1, out_d2, surv_list, surv_count, prune_r2, prune_thr, bounds_out, exact,
                        s_best, s_second);
}

// ------------------------------------------------------------------------------------------------
// the exact list pass (round 6): survivors of jobs on exact integer codes
// ------------------------------------------------------------------------------------------------
// The survivors of the screen need their exact two nearest columns.  Until round 5 that was one f16 product per (survivor,
// column) with a certified error bound (match_cand_f16_kernel<LIST>).  With exact integer codes the same sweep runs on the
// int8 pipe at twice the rate and needs no error analysis: the products are exact, and the screen has left every survivor a
// threshold (screen_tail) that only the columns which can be its best or second best by the canonical f32 distance reach - a
// handful per row.  This kernel streams a pooled 512-row survivor tile against its B set exactly like the screening kernel
// (same LDS image, DMA, hand-over, read-ahead, C operand), compares instead of folding - per (32-column block, row group)
// the maximum of the lane's eight products against the row's threshold, and only on a hit (rare) the eight values one by
// one - and appends (column; its position in the B set's divisor-sorted order) to the row's candidate list: cand[p * kCandCap ..], count in cand_cnt[p], p = the row's position
// in the pooled survivor list (counts beyond the capacity are kept: such a row goes to the exact-f32 fallback).  match_rescore_kernel then evaluates the canonical f32
// distance of every candidate and writes the row's (idx, d1, d2).  Whole register file claimed like the screening kernels.
constexpr int kCandCap = 32;

__global__ __launch_bounds__(512) void match_list_i8_kernel(const MatchJob* __restrict__ jobs, const WgJob* __restrict__ wgs, int n_wg,
                                                            const uint32_t* __restrict__ row_list, const int* __restrict__ list_job,
                                                            const float* __restrict__ thr_slot, uint32_t* __restrict__ cand,
                                                            unsigned int* __restrict__ cand_cnt) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * kQTileBytes + 3 * 1024];
    __shared__ unsigned int s_cnt[512];  // candidates found so far per row of the tile (the four lane quarters of a wave share a row)
    s_cnt[threadIdx.x] = 0u;
    int wg = blockIdx.x;
    {
        const int q = n_wg / 8, r = n_wg % 8, x = wg % 8;
        wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + wg / 8;
    }
    asm volatile("v_mov_b32 v255, 0" ::: "v255");  // (nothing shares a SIMD with int8-MFMA waves)
    const WgJob w = wgs[wg];
    const MatchJob jb = jobs[w.job];  // (any job of the tile's group: the B set is common)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15;
    const int kq = lane >> 4;
    const int nB = jb.nB;
    constexpr int kNone = kScreenNone;
    // this lane's four rows: list entries w.row0 + wave * 64 + 16 g + c (entries past the tile's end: no row, nothing can hit)
    i32x4 aq[4][2];
    // thr[g]: the row's threshold in accumulator units for the tile being streamed, thr_prev[g]: for the tile before (the last
    // block of a tile is tested in the first block of the next); s2[g] = the row's lower bound of its second largest I / t_j
    // (screen_tail), rcr[g] = 128 sum p_i + 128^3.  A column of a tile with smallest divisor tlo can reach s2 only with
    // I >= s2 tlo; margin 2e-4 (see screen_tail), -2 for the f32 product's rounding.  A tile of all-zero columns (tlo = 0: I = 0,
    // similarity 0) qualifies only for a row whose bound is not positive.
    int thr[4], thr_prev[4];
    float s2[4];
    int rcr[4];
    int pos_of[4];
    auto tile_thr = [&](int g, float tlo) __attribute__((always_inline)) {
        if (!(s2[g] > -1e9f)) return -2147483647 - 1;          // no bound (or a dead lane's +inf, below): everything / nothing
        if (!(tlo > 0.f)) return s2[g] > 0.f ? 2147483647 : -2147483647 - 1;
        const float v = floorf(s2[g] * tlo * (1.0f - 2e-4f)) - 2.0f;
        return v < 2.0e9f ? (int)v - rcr[g] : 2147483647;
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int e = wave * 64 + 16 * g + c;
        const bool live = e < w.list_cnt;
        pos_of[g] = w.row0 + (live ? e : 0);
        const int arow = (int)row_list[pos_of[g]];  // (the pooled list names rows within their job: list_pool_kernel)
        const MatchJob& jr = jobs[list_job[pos_of[g]]];
        s2[g] = live ? thr_slot[jr.out_off + arow] : INFINITY;  // (+inf: a lane without a row never hits)
        rcr[g] = jr.cinA[arow] + 2097152;
        thr[g] = thr_prev[g] = 2147483647;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aq[g][ks] = *reinterpret_cast<const i32x4*>(jr.AX + (size_t)arow * kDim + 64 * ks + 16 * kq);
    }
    const int ntiles = (nB + kQTN - 1) / kQTN;
    const int dma_sub = lane >> 3, dma_pos = lane & 7;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto issue_piece = [&](int t, int buf, int u) {
        const int piece = wave * 4 + u;
        const int lrow = 8 * piece + dma_sub;
        const int brow = min(t * kQTN + lrow, nB - 1);
        const signed char* src = jb.BXs + (size_t)brow * kDim + ((dma_pos ^ ((lrow >> 1) & 7)) << 4);
        const uint32_t dst = lds_base + buf * kQTileBytes + piece * 1024;
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(src), "s"(dst)
            : "memory");
    };
    const int ncin = jb.ncinB;
    auto issue_cin = [&](int t, int buf) {
        if (wave == 0) {
            const int col = min(t * kQTN + 4 * lane, ncin - 4);
            const int* src = jb.cinBs + col;
            const uint32_t dst = lds_base + 3 * kQTileBytes + buf * 1024;
            uint32_t keep;
            asm volatile(
                "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                : "=&s"(keep)
                : "v"(src), "s"(dst)
                : "memory");
        }
    };
    i32x4 acc[2][2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[p][u][g][e] = kNone;
    // the test of a finished block for row group g: col0 = first column of the block; columns >= nB never count
    auto test_group = [&](auto PAR, auto G, int col0, const int (&thr)[4]) __attribute__((always_inline)) {
        constexpr int par = decltype(PAR)::value, g = decltype(G)::value;
        int v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[par][e >> 2][g][e & 3];
        int m;
        asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_max3_i32 %0, %0, %4, %5\n\tv_max3_i32 %0, %0, %6, %7\n\tv_max_i32 %0, %0, %8"
                     : "=&v"(m)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        if (__builtin_expect(__any(m >= thr[g]), 0)) {
            // (one emission site behind a bit mask of the lane's hits: with the eight tests unrolled around eight atomics the
            // kernel's loop body outgrew the instruction cache - 32 call sites per tile - and ran at half the screen's rate)
            unsigned hits = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) hits |= (v[e] >= thr[g] ? 1u : 0u) << e;
#pragma unroll 1
            while (hits) {
                const int e = __builtin_ctz(hits);
                hits &= hits - 1u;
                const int col = col0 + 16 * (e >> 2) + 4 * kq + (e & 3);
                if (col < nB) {
                    // (the row's counter lives in LDS: a returning GLOBAL atomic is waited for on vmcnt, behind the tile's
                    // LDS-DMA pieces in flight - every hit then stalled its wave until the next tile had landed)
                    const unsigned int pos = atomicAdd(&s_cnt[wave * 64 + 16 * g + c], 1u);
                    if (pos < (unsigned)kCandCap) cand[(size_t)pos_of[g] * kCandCap + pos] = (uint32_t)col;
                }
            }
        }
    };
    if (ntiles > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) issue_piece(0, 0, u);
        issue_cin(0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" ::"v"(aq[g][0]), "v"(aq[g][1]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int hx = (16 * kq) ^ (16 * ((c >> 1) & 7));
    constexpr int kAhead = 3;
    auto slot_off = [&](int s) __attribute__((always_inline)) { return (s >> 1) * 16 * kDim + ((64 * (s & 1)) ^ hx); };
    i32x4 bq[4];
    i32x4 cin[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned char* const cin_lds = lds + 3 * kQTileBytes + 16 * kq;
    if (ntiles > 0) {
        if (ntiles > 1) {
            issue_piece(1, 1, 0);
            issue_cin(1, 1);
        }
#pragma unroll
        for (int s = 0; s < kAhead; ++s) bq[s] = *reinterpret_cast<const i32x4*>(lds + c * kDim + slot_off(s));
        cin[0] = *reinterpret_cast<const i32x4*>(cin_lds);
        cin[1] = *reinterpret_cast<const i32x4*>(cin_lds + 64);
    }
    int b_cur = 0;
    for (int t = 0; t < ntiles; ++t) {
        const bool more = t + 1 < ntiles;
        const bool more2 = t + 2 < ntiles;
        const int b_nxt = b_cur == 2 ? 0 : b_cur + 1, b_nxt2 = b_nxt == 2 ? 0 : b_nxt + 1;
        const unsigned char* tile = lds + b_cur * kQTileBytes + c * kDim;
        const unsigned char* tile_n = lds + b_nxt * kQTileBytes + c * kDim;
        const unsigned char* cin_t = cin_lds + b_cur * 1024;
        const unsigned char* cin_n = cin_lds + b_nxt * 1024;
        b_cur = b_nxt;
        {  // this tile's thresholds (its smallest divisor: wave-uniform); the previous tile's stay for the first block's test
            const float tlo = jb.tscB[t].z;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                thr_prev[g] = thr[g];
                thr[g] = tile_thr(g, tlo);
            }
        }
        static_for<0, kQBlk>([&](auto CB) {
            constexpr int cb = decltype(CB)::value;
            constexpr int kLast = kQBlk - 1;
            constexpr int par = cb & 1;
            if (cb == 0 || cb == 2 || cb == 4) {
                if (more) issue_piece(t + 1, b_nxt, cb / 2 + 1);
            } else if (cb == kLast) {
                if (more2) {
                    issue_piece(t + 2, b_nxt2, 0);
                    issue_cin(t + 2, b_nxt2);
                }
            }
            const unsigned char* blk = tile + cb * 32 * kDim;
            const unsigned char* nblk = cb < kLast ? blk + 32 * kDim : tile_n;
            const bool fetch = cb < kLast || more;
            // the block being tested is the previous one: (t, cb - 1), or the last block of tile t - 1
            const int col_prev = cb > 0 ? t * kQTN + (cb - 1) * 32 : (t - 1) * kQTN + kLast * 32;
            static_for<0, 4>([&](auto S) {
                constexpr int s = decltype(S)::value;
                constexpr int u = s >> 1, ks = s & 1;
                const i32x4 xq = bq[s & 3];
                if (s + kAhead < 4) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(blk + slot_off(s + kAhead));
                } else if (fetch) {
                    bq[(s + kAhead) & 3] = *reinterpret_cast<const i32x4*>(nblk + slot_off(s + kAhead - 4));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[par][u][g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xq, aq[g][ks], ks == 0 ? cin[u] : acc[par][u][g], 0, 0, 0);
                if (ks == 1) {
                    if (cb < kLast)
                        cin[u] = *reinterpret_cast<const i32x4*>(cin_t + (32 * (cb + 1) + 16 * u) * 4);
                    else if (more)
                        cin[u] = *reinterpret_cast<const i32x4*>(cin_n + (16 * u) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (cb > 0)
                    test_group(std::integral_constant<int, par ^ 1>{}, S, col_prev, thr);
                else if (t > 0)
                    test_group(std::integral_constant<int, par ^ 1>{}, S, col_prev, thr_prev);
                __builtin_amdgcn_sched_barrier(0);
            });
            if (cb == kLast - 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        });
    }
    if (ntiles > 0) {
        const int col_last = (ntiles - 1) * kQTN + (kQBlk - 1) * 32;
        static_for<0, 4>([&](auto S) { test_group(std::integral_constant<int, (kQBlk - 1) & 1>{}, S, col_last, thr); });
    }
    __syncthreads();
    if (tid < w.list_cnt) cand_cnt[w.row0 + tid] = s_cnt[tid];  // (row e of the tile = wave e / 64, group (e % 64) / 16, c = e % 16: s_cnt's order)
}

// The candidates of a row (match_list_i8_kernel) -> its exact (idx, d1, d2): eight lanes per row, lane e evaluates the
// canonical f32 distance (exact_dist: the k-ascending fma chain of the contract) of candidates e, e + 8, ..., keeping its two
// smallest by (distance, index); a butterfly over the eight lanes merges them.  Rows with more candidates than slots go to
// the exact-f32 fallback list of their job.
__global__ __launch_bounds__(256) void match_rescore_kernel(const MatchJob* __restrict__ jobs, const uint32_t* __restrict__ row_list,
                                                            const int* __restrict__ list_job, int64_t n_rows,
                                                            const uint32_t* __restrict__ cand, const unsigned int* __restrict__ cand_cnt,
                                                            uint32_t* __restrict__ out_idx, float* __restrict__ out_d1, float* __restrict__ out_d2,
                                                            uint32_t* __restrict__ fb_list, unsigned int* __restrict__ fb_count) {
    const int64_t r = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int e0 = threadIdx.x & 7;
    const bool live = r < n_rows;
    const int row = (int)row_list[live ? r : 0];  // (a row within its job)
    const int job = list_job[live ? r : 0];
    const MatchJob& jb = jobs[job];
    const int64_t slot = jb.out_off + row;
    const unsigned int cnt = live ? cand_cnt[r] : 0u;
    float b = INFINITY, s2 = INFINITY;
    int bi = 0x7fffffff;
    if (live && cnt <= (unsigned)kCandCap) {
        const float* pa = jb.PA + (size_t)row * kDim;
        const float a2 = jb.sqA[row];
        for (unsigned int e = e0; e < cnt; e += 8u) {
            const int id = (int)jb.permB[cand[(size_t)r * kCandCap + e]];  // (the list pass names positions in the sorted column order)
            const float d = exact_dist(pa, jb.PB + (size_t)id * kDim, a2, jb.sqB[id]);
            top2_merge(b, bi, s2, d, id, INFINITY);
        }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
        const float ob = __shfl_xor(b, off), os = __shfl_xor(s2, off);
        const int oi = __shfl_xor(bi, off);
        top2_merge(b, bi, s2, ob, oi, os);
    }
    if (live && e0 == 0) {
        if (cnt > (unsigned)kCandCap || cnt == 0u) {  // (no candidate cannot happen for nB >= 1; the exact kernel decides then)
            const unsigned int p = atomicAdd(fb_count + job, 1u);
            fb_list[jb.out_off + p] = (uint32_t)slot;
        } else {
            out_idx[slot] = (uint32_t)bi + 1u;
            out_d1[slot] = b;
            out_d2[slot] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// ratio / threshold / uniqueness  (matchFeaturesScratch.m:170-211)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t order_f32(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_f32(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

struct FilterJob {
    int64_t row_off;  // first row slot (same as MatchJob::out_off)
    int64_t col_off;  // first slot of this job's B columns in the winner table
    int nA;
    int nB;
};

// row -> job lookup by binary search over row_off
__device__ __forceinline__ int find_job(const FilterJob* __restrict__ fj, int njobs, int64_t slot) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (fj[mid].row_off <= slot)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

__global__ void filter_mark_kernel(const FilterJob* __restrict__ fj, int njobs, int64_t total_rows,
                                   const uint32_t* __restrict__ idx, const float* __restrict__ d1,
                                   const float* __restrict__ d2, double r2, double thr, int unique,
                                   unsigned long long* __restrict__ keys,  // per row, ~0 if dropped
                                   unsigned long long* __restrict__ winner) {
    const int64_t slot = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (slot >= total_rows) return;
    const double b = (double)d1[slot], s = (double)d2[slot];
    // ratioOK = dBest <= r2*dSecond; threshOK = dBest <= MatchThreshold; both finite (:173-178)
    const bool keep = idx[slot] != 0 && (b <= r2 * s) && (b <= thr) && isfinite(b) && isfinite(s);
    unsigned long long key = ~0ull;
    if (keep) {  // (the row -> job search, eleven dependent loads, only for the few rows that pass)
        const FilterJob f = fj[find_job(fj, njobs, slot)];
        const uint32_t row = (uint32_t)(slot - f.row_off);
        key = ((unsigned long long)order_f32(d1[slot]) << 32) | row;
        if (unique) atomicMin(&winner[f.col_off + (idx[slot] - 1)], key);
    }
    keys[slot] = key;
}

__global__ void filter_select_kernel(const FilterJob* __restrict__ fj, int njobs,
                                     int64_t total_rows, const uint32_t* __restrict__ idx,
                                     int unique, const unsigned long long* __restrict__ keys,
                                     const unsigned long long* __restrict__ winner,
                                     unsigned long long* __restrict__ packed,  // kept keys, dense per job
                                     unsigned long long* __restrict__ job_count) {
    const int64_t slot = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    unsigned long long key = ~0ull;
    if (slot < total_rows) key = keys[slot];
    bool kept = key != ~0ull;
    int j = 0;
    if (kept) {
        j = find_job(fj, njobs, slot);
        if (unique && winner[fj[j].col_off + (idx[slot] - 1)] != key) kept = false;
    }
    // the kept keys of a job form the head of its own row segment (any order: the sort that follows
    // works on distinct keys), so the sort touches the survivors only.  One counter update per wave when its kept rows
    // belong to one job (a wave spans 64 consecutive rows): an overlapping pair keeps thousands of rows, and that many
    // same-address atomics serialised in L2 were most of this kernel's time.
    const unsigned long long m = __ballot(kept);
    if (m == 0ull) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    const int jl = __shfl(j, leader);
    unsigned long long pos;
    if (__ballot(kept && j != jl) == 0ull) {
        unsigned long long base = 0ull;
        if (lane == leader) base = atomicAdd(&job_count[jl], (unsigned long long)__popcll(m));
        base = __shfl(base, leader);
        pos = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
    } else if (kept) {
        pos = atomicAdd(&job_count[j], 1ull);
    }
    if (kept) packed[fj[j].row_off + pos] = unique ? key : (key & 0xffffffffull);  // non-unique lists are in row order
}

__global__ void filter_seg_end_kernel(const FilterJob* __restrict__ fj, int njobs,
                                      const unsigned long long* __restrict__ job_count,
                                      int64_t* __restrict__ seg_begin, int64_t* __restrict__ seg_end) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    seg_begin[j] = fj[j].row_off;
    seg_end[j] = fj[j].row_off + (int64_t)job_count[j];
}

// After sorting each job's key segment ascending, the kept entries are the first job_count[j] keys.
__global__ void filter_emit_kernel(const FilterJob* __restrict__ fj, int njobs,
                                   const unsigned long long* __restrict__ sorted_keys,
                                   const unsigned long long* __restrict__ job_ptr,  // exclusive scan
                                   const uint32_t* __restrict__ idx, int unique,
                                   uint32_t* __restrict__ o1, uint32_t* __restrict__ o2,
                                   float* __restrict__ metric, int64_t cap, int job0) {
    const int j = job0 + blockIdx.y;  // gridDim.y <= 65535: the host walks the jobs in chunks
    const FilterJob f = fj[j];
    const unsigned long long base = job_ptr[j];
    const unsigned long long cnt = job_ptr[j + 1] - base;
    for (unsigned long long e = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; e < cnt;
         e += (unsigned long long)gridDim.x * blockDim.x) {
        if ((int64_t)(base + e) >= cap) continue;
        const unsigned long long key = sorted_keys[f.row_off + e];
        const uint32_t row = (uint32_t)(key & 0xffffffffu);
        o1[base + e] = row + 1;
        o2[base + e] = idx[f.row_off + row];
        metric[base + e] = unorder_f32((uint32_t)(key >> 32));
    }
}

// The non-unique list is emitted in row order: filter_select_kernel reduced its keys to the row, the metric is
// re-read from d1.
__global__ void filter_emit_rows_kernel(const FilterJob* __restrict__ fj,
                                        const unsigned long long* __restrict__ sorted_keys,
                                        const unsigned long long* __restrict__ job_ptr,
                                        const uint32_t* __restrict__ idx,
                                        const float* __restrict__ d1, uint32_t* __restrict__ o1,
                                        uint32_t* __restrict__ o2, float* __restrict__ metric,
                                        int64_t cap, int job0) {
    const int j = job0 + blockIdx.y;
    const FilterJob f = fj[j];
    const unsigned long long base = job_ptr[j];
    const unsigned long long cnt = job_ptr[j + 1] - base;
    for (unsigned long long e = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; e < cnt;
         e += (unsigned long long)gridDim.x * blockDim.x) {
        if ((int64_t)(base + e) >= cap) continue;
        const uint32_t row = (uint32_t)(sorted_keys[f.row_off + e] & 0xffffffffu);
        o1[base + e] = row + 1;
        o2[base + e] = idx[f.row_off + row];
        metric[base + e] = d1[f.row_off + row];
    }
}


// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// The operand arrays of the sets of ONE matching call come out of one slab (round 6): 64 sets x 16 arrays were 1024 trips
// through the workspace pool's best-fit scan per call, ~0.4 ms of host time between the probe and the first preparation
// launch.  An Arena in counting mode only adds up the sizes (the dry run that sizes the slab).
struct Arena {
    Ws<unsigned char> slab;
    Ws<unsigned char> sorted;  // the sets' column sides in divisor order (sort_columns)
    size_t off = 0, cap = 0;
    bool counting = false;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~size_t(255);
        void* p = counting ? nullptr : static_cast<void*>(slab.get() + off);
        APS_REQUIRE(counting || off + bytes <= cap, APS_E_INTERNAL, "descriptor arena too small");
        off += bytes;
        return p;
    }
};
template <class T>
struct Buf {  // an array of a prepared set: a piece of the call's arena, or its own workspace block
    T* p = nullptr;
    Ws<T> own;
    void alloc(size_t n, Arena* a) {
        if (a) {
            p = static_cast<T*>(a->take((n ? n : 1) * sizeof(T)));
        } else {
            own.alloc(n);
            p = own.get();
        }
    }
    T* get() const { return p; }
    operator T*() const { return p; }
};
struct Prepared {
    Buf<float> P, sq, dn, maxsq;  // maxsq[0] = max ||x||^2, [1] = max ||x - f16(x)||, [2], [3] = aug residuals, [4..7] int8 screen
    Buf<unsigned short> H;
    Buf<uint4> aug;
    Buf<signed char> QA, QB;  // int8 screening copies (row side / column side) and their side data (see MatchJob)
    Buf<float> dnq, invs;
    Buf<int> sumq;
    Buf<signed char> QX;  // round 6: the exact integer codes (q8_desc_rows), the rows' divisors, 128 x the code sums (n_pad entries)
    Buf<float> tt;
    Buf<int> cin;
    // ... and the set as a COLUMN set in ascending order of the divisors (sort_columns; arena batches only - a set without
    // these has its "no exact code" word forced, so that its jobs take the rounded codes)
    signed char* BXs = nullptr;
    int* cins = nullptr;
    float4* tsc = nullptr;
    uint32_t* perm = nullptr;
    int64_t n_pad = 0;
    Buf<unsigned> part;  // per-workgroup maxima of prep_desc / q8_desc (PrepJob::part)
    int nb1 = 0, nb2 = 0;
    float* stat = nullptr;  // the kStatWords statistics words: own (maxsq) or a slice of the caller's block (one fill for many sets)
    int64_t n = 0;
};

// `st`: the stream the set's launches go to (the caller's own stream, or one of its auxiliary streams when many sets are
// prepared side by side - each set is a chain of small launches that leaves most of the chip idle)
// the buffers of a set (and, when the set owns its statistics words, their zero fill on `st`)
static void prepare_alloc(int64_t n, Prepared& out, hipStream_t st, float* stat_ext, Arena* arena = nullptr);

// An experiment switch read from the environment reaches the device when its value differs from what the device holds
// (also back to 0 when the variable is unset again), by a synchronous copy from a live variable - same-process A/Bs see
// the value they set, and no copy is queued from a stack slot that is gone when it runs.
// (the symbols are per DEVICE and process-wide, so what a device holds is tracked per (device, symbol) for the
// whole process under a lock - a per-thread note went stale when another thread of the same device, or the same thread on
// another device, had changed the value.)
static void sync_device_switch(const char* env, const void* symbol) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, int> held;
    const char* e = std::getenv(env);
    const int want = e ? std::atoi(e) : 0;
    int dev = 0;
    APS_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    int& h = held[{dev, symbol}];  // (0 on first sight: the symbols' initial value)
    if (want == h) return;
    APS_HIP(hipStreamSynchronize(stream()));  // kernels in flight keep the value they were launched under
    APS_HIP(hipMemcpyToSymbol(symbol, &want, sizeof want, 0, hipMemcpyHostToDevice));
    h = want;
}
static void sync_q8_symmetric_switch() {
    sync_device_switch("APS_Q8_SYMMETRIC", &g_q8_symmetric);
    sync_device_switch("APS_MATCH_NO_EXACT", &g_q8_noexact);
#ifdef APS_MATCH_TIMING
    sync_device_switch("APS_Q8_CENTER", &g_q8_center);
#endif
}

static void prepare(const float* X_dev, int64_t n, int64_t ld, int layout, bool normalize,
                    Prepared& out, hipStream_t st = nullptr, bool bracket = true, float* stat_ext = nullptr) {
    if (!st) st = stream();
    prepare_alloc(n, out, st, stat_ext);
    float* const qstat = out.stat + 4;
    const int64_t n_pad = (std::max<int64_t>(n, 1) + kTNB - 1) / kTNB * kTNB;
    sync_q8_symmetric_switch();  // A/B switch APS_Q8_SYMMETRIC: column code without the offset
    if (n == 0) return;
    struct MaybeProf {  // (event brackets live on the caller's own stream: a forked section is bracketed as a whole)
        Prof* p = nullptr;
        explicit MaybeProf(bool on) { if (on) p = new Prof("match_prep"); }
        ~MaybeProf() { delete p; }
    } prof(bracket);
    prep_desc_kernel<<<out.nb1, kPrepThreads, 0, st>>>(X_dev, n, ld, layout, normalize ? 1 : 0, out.P, out.sq, out.H, out.dn, out.part);
    prep_stats_kernel<<<1, 256, 0, st>>>(out.part, out.nb1, out.nb2, out.stat, (const unsigned short*)out.H != nullptr ? 1 : 0, 1);
    q8_desc_kernel<<<out.nb2, 256, 0, st>>>(out.P, n, out.QA, out.QB, out.dnq, out.invs, out.sumq, qstat, out.sq, out.dn, n_pad, out.stat,
                                            out.aug, out.stat + 2, out.part + (size_t)5 * out.nb1, out.QX, out.tt, out.cin);
    prep_stats_kernel<<<1, 256, 0, st>>>(out.part, out.nb1, out.nb2, out.stat, 1, 2);
    // (no sorted column side here - sort_columns runs for arena batches only: the set's jobs take the rounded codes)
    APS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out.stat + 10), 1, 1, st));
    check_launch("prep_desc_kernel");
}

// ---- the column side of the exact codes in ascending order of the divisors (round 6) -------------------------------------
// The exact screen's only slack is the range of the column divisors it has to assume.  Over a whole set that range is ~0.7 %
// (SIFT: t = 512 +- 0.35 %), and ONE odd column - a descriptor saturated at 255, say - would widen it for every row; sorted by
// t, a 256-column tile spans 1/78 of it and an outlier only spoils its own tile.  Per set: keys (the divisor's bit pattern;
// all-zero rows, which fit any divisor, and padding last) -> rocPRIM segmented sort with the row index as the value ->
// gather of the code bytes and of 128 sum p in sorted order, and per tile the divisor range of its non-zero columns with the
// reciprocals rounded outwards.
struct SortSet {
    const float* tt;
    const signed char* QX;
    const int* cin;
    int n, off;  // rows, first slot in the batch-wide key arrays
    signed char* BXs;
    int* cins;
    float4* tsc;
};
constexpr uint32_t kSortLast = 0x7f7fffffu;  // (FLT_MAX: behind every real divisor)
__global__ __launch_bounds__(256) void sort_keys_kernel(const SortSet* __restrict__ sets, const int* __restrict__ blk_ptr, int n_sets,
                                                        unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int j = prep_find_job(blk_ptr, n_sets, (int)blockIdx.x);
    const SortSet S = sets[j];
    const int k = ((int)blockIdx.x - blk_ptr[j]) * 256 + (int)threadIdx.x;
    if (k >= S.n) return;
    // (set index above the divisor's bit pattern: ONE radix sort orders every set of the batch - rocPRIM's segmented sort took
    // 0.44 ms for 64 segments of 20 k keys, the plain sort of the 1.27 M composite keys a quarter of that)
    const uint32_t tb = S.cin[k] == -2097152 ? kSortLast : __float_as_uint(S.tt[k]);  // (128 sum p = -128^3: an all-zero row)
    keys[S.off + k] = ((unsigned long long)j << 32) | tb;
    vals[S.off + k] = (uint32_t)k;
}
__global__ __launch_bounds__(256) void sort_gather_kernel(const SortSet* __restrict__ sets, const int* __restrict__ blk_ptr, int n_sets,
                                                          const unsigned long long* __restrict__ keys_sorted, const uint32_t* __restrict__ perm) {
    const int j = prep_find_job(blk_ptr, n_sets, (int)blockIdx.x);
    const SortSet S = sets[j];
    const int g = ((int)blockIdx.x - blk_ptr[j]) * 256 + (int)threadIdx.x;
    const int k = g >> 3, part = g & 7;  // eight lanes per row, 16 bytes each
    if (k >= S.n) return;
    const uint32_t src = perm[S.off + k];
    typedef __attribute__((address_space(1))) signed char GI8;
    *(GU32x4*)((GI8*)S.BXs + (size_t)k * kDim + 16 * part) = *(const GU32x4*)((const GI8*)S.QX + (size_t)src * kDim + 16 * part);
    if (part != 0) return;
    S.cins[k] = S.cin[src];
    const uint32_t key = (uint32_t)keys_sorted[S.off + k];
    if (key == kSortLast) return;
    const float t = __uint_as_float(key);
    const int T = k >> 8;
    float* sc = reinterpret_cast<float*>(S.tsc + T);
    if ((k & 255) == 0) {  // the tile's first column: its smallest divisor (the key is real, so the tile has non-zero columns)
        sc[0] = __fdiv_rn(1.0f, t) * 1.0000004f;  // (rounded to nearest, then pushed up by 3 ulp)
        sc[2] = t;
    }
    if ((k & 255) == 255 || k == S.n - 1 || (uint32_t)keys_sorted[S.off + k + 1] == kSortLast) {  // ... and its last non-zero column
        sc[1] = __fdiv_rn(1.0f, t) * 0.9999996f;  // (... down)
        sc[3] = t;
    }
}

// Several sets in four launches on the caller's stream (prep_desc_batch_kernel, q8_desc_batch_kernel, each followed by the fold of
// its workgroups' maxima, prep_stats_batch_kernel).  stat_ext[s]: the
// set's eight zeroed statistics words (the caller's block).
struct PrepRequest {
    const float* X;
    int64_t n, ld;
    bool normalize;
    Prepared* out;
    float* stat_ext;
};
// The sorted column sides of a batch (see sort_keys_kernel).  Planned on the host before the batch's first launch - buffers, set
// table, block tables - so that the whole preparation is ONE upload and one chain of launches (the bench's matching runs
// beside the previous panorama's 737 MB download: every extra small upload or host synchronisation inside the chain waited
// its turn behind that copy, 1 ms became 3).
struct SortPlan {
    std::vector<SortSet> sets;
    std::vector<int> bk{0}, bg{0};
    size_t total = 0, tmp_bytes = 0;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    uint32_t *vals_in = nullptr, *vals_out = nullptr;
    unsigned char* tsc0 = nullptr;
    size_t tsc_bytes = 0;
    int key_bits = 33;
    Ws<unsigned char> tmp;
};
static void sort_plan(const std::vector<PrepRequest>& req, Arena& arena, SortPlan& sp) {
    size_t bytes = 0;
    auto up = [](size_t v) { return (v + 255) & ~size_t(255); };
    for (const PrepRequest& r : req) {
        if (r.n <= 0) continue;
        const Prepared& o = *r.out;
        sp.sets.push_back(SortSet{o.tt, o.QX, o.cin, (int)r.n, (int)sp.total, nullptr, nullptr, nullptr});
        sp.total += (size_t)r.n;  // (dense: the sorted keys of set j occupy [off_j, off_j + n_j))
        sp.bk.push_back(sp.bk.back() + (int)cdiv((size_t)r.n, 256));
        sp.bg.push_back(sp.bg.back() + (int)cdiv((size_t)r.n * 8, 256));
        bytes += up((size_t)r.n * kDim) + up((size_t)o.n_pad * sizeof(int)) + up(cdiv((size_t)r.n, 256) * sizeof(float4));
    }
    if (sp.sets.empty()) return;
    APS_REQUIRE(sp.total < ((size_t)1 << 31), APS_E_DIM, "too many descriptor rows in one batch (%zu)", sp.total);
    const size_t key_bytes = up(sp.total * sizeof(unsigned long long)), val_bytes = up(sp.total * sizeof(uint32_t));
    arena.sorted.alloc(bytes + 2 * key_bytes + 2 * val_bytes);
    unsigned char* base = arena.sorted.get();
    sp.keys_in = reinterpret_cast<unsigned long long*>(base);
    sp.keys_out = reinterpret_cast<unsigned long long*>(base + key_bytes);
    sp.vals_in = reinterpret_cast<uint32_t*>(base + 2 * key_bytes);
    sp.vals_out = reinterpret_cast<uint32_t*>(base + 2 * key_bytes + val_bytes);  // = the sets' perm arrays, back to back
    size_t at = 2 * key_bytes + 2 * val_bytes;
    size_t si = 0;
    for (const PrepRequest& r : req) {
        if (r.n <= 0) continue;
        Prepared& o = *r.out;
        SortSet& S = sp.sets[si++];
        S.BXs = o.BXs = reinterpret_cast<signed char*>(base + at);
        at += up((size_t)r.n * kDim);
        S.cins = o.cins = reinterpret_cast<int*>(base + at);
        at += up((size_t)o.n_pad * sizeof(int));
        o.perm = sp.vals_out + S.off;
    }
    sp.tsc0 = base + at;
    si = 0;
    for (const PrepRequest& r : req) {  // (the tile tables last, back to back: one zero fill)
        if (r.n <= 0) continue;
        Prepared& o = *r.out;
        SortSet& S = sp.sets[si++];
        S.tsc = o.tsc = reinterpret_cast<float4*>(base + at);
        at += up(cdiv((size_t)r.n, 256) * sizeof(float4));
    }
    sp.tsc_bytes = (size_t)(base + at - sp.tsc0);
    int set_bits = 1;
    while (((size_t)1 << set_bits) < sp.sets.size()) ++set_bits;
    sp.key_bits = 32 + set_bits;
    APS_HIP(rocprim::radix_sort_pairs(nullptr, sp.tmp_bytes, sp.keys_in, sp.keys_out, sp.vals_in, sp.vals_out, sp.total, 0, sp.key_bits, stream()));
    sp.tmp.alloc(sp.tmp_bytes);
}
// (dsets / dbk / dbg: the plan's tables on the device, part of the batch's one upload)
static void sort_launch(SortPlan& sp, const SortSet* dsets, const int* dbk, const int* dbg) {
    if (sp.sets.empty()) return;
    APS_HIP(hipMemsetAsync(sp.tsc0, 0, sp.tsc_bytes, stream()));
    // (the 128 sum p of sorted positions past a set's rows are never used: the kernels clamp to n - 1 / mask by nB)
    sort_keys_kernel<<<(unsigned)sp.bk.back(), 256, 0, stream()>>>(dsets, dbk, (int)sp.sets.size(), sp.keys_in, sp.vals_in);
    APS_HIP(rocprim::radix_sort_pairs(sp.tmp.get(), sp.tmp_bytes, sp.keys_in, sp.keys_out, sp.vals_in, sp.vals_out, sp.total, 0, sp.key_bits, stream()));
    sort_gather_kernel<<<(unsigned)sp.bg.back(), 256, 0, stream()>>>(dsets, dbg, (int)sp.sets.size(), sp.keys_out, sp.vals_out);
    check_launch("sort_gather_kernel");
}

static void prepare_batch(const std::vector<PrepRequest>& req, int layout, Arena* arena = nullptr) {
    std::vector<PrepJob> jobs;
    std::vector<int> bp{0}, bq{0};
    if (arena) {  // the dry run that sizes the slab (stat_ext is set for every set of a batch)
        Arena count;
        count.counting = true;
        Prepared dry;
        for (const PrepRequest& r : req) prepare_alloc(r.n, dry, stream(), r.stat_ext, &count);
        arena->slab.alloc(count.off);
        arena->cap = count.off;
        arena->off = 0;
    }
    for (const PrepRequest& r : req) {
        prepare_alloc(r.n, *r.out, stream(), r.stat_ext, arena);
        if (r.n == 0) continue;
        const Prepared& o = *r.out;
        const int64_t n_pad = (std::max<int64_t>(r.n, 1) + kTNB - 1) / kTNB * kTNB;
        jobs.push_back(PrepJob{r.X, r.n, r.ld, n_pad, layout, r.normalize ? 1 : 0, o.P, o.sq, o.dn, o.stat, o.H, o.QA, o.QB, o.dnq, o.invs, o.sumq, o.aug,
                               o.part, o.nb1, o.nb2, o.QX, o.tt, o.cin});
        bp.push_back(bp.back() + o.nb1);
        bq.push_back(bq.back() + o.nb2);
    }
    sync_q8_symmetric_switch();
    if (jobs.empty()) return;
    SortPlan sp;
    if (arena) sort_plan(req, *arena, sp);
    std::vector<int> bq1(jobs.size() + 1);  // the FIRST q8 workgroup of every set (the exact codes' probe below)
    for (size_t j = 0; j <= jobs.size(); ++j) bq1[j] = (int)j;
    // every host table of the batch in ONE upload: [PrepJob][SortSet][bp][bq][bq1][bk][bg]
    auto up16 = [](size_t v) { return (v + 15) & ~size_t(15); };
    const size_t o_jobs = 0, o_sets = o_jobs + up16(jobs.size() * sizeof(PrepJob)), o_bp = o_sets + up16(sp.sets.size() * sizeof(SortSet)),
                 o_bq = o_bp + up16(bp.size() * sizeof(int)), o_bq1 = o_bq + up16(bq.size() * sizeof(int)),
                 o_bk = o_bq1 + up16(bq1.size() * sizeof(int)), o_bg = o_bk + up16(sp.bk.size() * sizeof(int)),
                 o_end = o_bg + up16(sp.bg.size() * sizeof(int));
    std::vector<unsigned char> blob(o_end, 0);
    std::memcpy(&blob[o_jobs], jobs.data(), jobs.size() * sizeof(PrepJob));
    if (!sp.sets.empty()) std::memcpy(&blob[o_sets], sp.sets.data(), sp.sets.size() * sizeof(SortSet));
    std::memcpy(&blob[o_bp], bp.data(), bp.size() * sizeof(int));
    std::memcpy(&blob[o_bq], bq.data(), bq.size() * sizeof(int));
    std::memcpy(&blob[o_bq1], bq1.data(), bq1.size() * sizeof(int));
    std::memcpy(&blob[o_bk], sp.bk.data(), sp.bk.size() * sizeof(int));
    std::memcpy(&blob[o_bg], sp.bg.data(), sp.bg.size() * sizeof(int));
    Ws<unsigned char> dblob(o_end);
    APS_HIP(hipMemcpyAsync(dblob, blob.data(), o_end, hipMemcpyHostToDevice, stream()));
    const PrepJob* dj = reinterpret_cast<const PrepJob*>(dblob.get() + o_jobs);
    const SortSet* dsets = reinterpret_cast<const SortSet*>(dblob.get() + o_sets);
    const int* dbp = reinterpret_cast<const int*>(dblob.get() + o_bp);
    const int* dbq = reinterpret_cast<const int*>(dblob.get() + o_bq);
    const int* dbq1 = reinterpret_cast<const int*>(dblob.get() + o_bq1);
    const int* dbk = reinterpret_cast<const int*>(dblob.get() + o_bk);
    const int* dbg = reinterpret_cast<const int*>(dblob.get() + o_bg);
    {
        Prof prof("match_prep");
        prep_desc_batch_kernel<<<(unsigned)bp.back(), kPrepThreads, 0, stream()>>>(dj, dbp, (int)jobs.size());
        prep_stats_batch_kernel<<<(unsigned)jobs.size(), 256, 0, stream()>>>(dj, 1);
        // The exact codes' search (q8_desc_rows) is cheap where it succeeds at the first or second try and expensive where it
        // cannot succeed at all - ordinary float descriptors: 24 tries per row, +2 ms for 64 sets of 20 k rows.  The first 32
        // rows of every set go first (one workgroup per set, the same kernel: it rewrites what it wrote): a set of floats has
        // lost by the time the full pass starts, and that pass skips the search for it.
        q8_desc_batch_kernel<<<(unsigned)jobs.size(), 256, 0, stream()>>>(dj, dbq1, (int)jobs.size());
        q8_desc_batch_kernel<<<(unsigned)bq.back(), 256, 0, stream()>>>(dj, dbq, (int)jobs.size());
        prep_stats_batch_kernel<<<(unsigned)jobs.size(), 256, 0, stream()>>>(dj, 2);
        if (arena) {
            sort_launch(sp, dsets, dbk, dbg);
        } else {  // no sorted column side: the sets' jobs must take the rounded codes ("some row has no exact code", word [10])
            for (const PrepRequest& r : req)
                if (r.n > 0) APS_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(r.out->stat + 10), 1, 1, stream()));
        }
    }
    check_launch("prep_desc_batch_kernel");
    APS_HIP(hipStreamSynchronize(stream()));  // the host tables (and the device blob, the sort's scratch) must outlive the launches
}

static void prepare_alloc(int64_t n, Prepared& out, hipStream_t st, float* stat_ext, Arena* arena) {
    Arena* const a = arena;
    out.n = n;
    const size_t rows = (size_t)std::max<int64_t>(n, 1);
    out.P.alloc(rows * kDim, a);
    out.sq.alloc(rows, a);
    out.H.alloc(rows * kDim, a);
    out.dn.alloc(rows, a);
    if (stat_ext) {  // zeroed by the caller
        out.stat = stat_ext;
    } else {
        APS_REQUIRE(a == nullptr, APS_E_INTERNAL, "a set in an arena takes its statistics words from the caller's block");
        out.maxsq.alloc(kStatWords, nullptr);  // [0..3] the f16 path's set statistics, [4..7] the int8 screen's, [8..10] its exact codes' (one fill)
        APS_HIP(hipMemsetAsync(out.maxsq, 0, kStatWords * sizeof(float), st));
        out.stat = out.maxsq;
    }
    const int64_t n_pad = (std::max<int64_t>(n, 1) + kTNB - 1) / kTNB * kTNB;
    out.aug.alloc((size_t)n_pad, a);
    out.QA.alloc(rows * kDim, a);
    out.QB.alloc(rows * kDim, a);
    out.dnq.alloc(rows, a);
    out.invs.alloc(rows, a);
    out.sumq.alloc(rows, a);
    out.QX.alloc(rows * kDim, a);
    out.tt.alloc(rows, a);
    out.cin.alloc((size_t)n_pad, a);
    out.n_pad = n_pad;
    out.nb1 = (int)cdiv(rows, kPrepRows);
    out.nb2 = (int)cdiv((size_t)n_pad * 8, 256);
    out.part.alloc((size_t)5 * out.nb1 + (size_t)kQ8Part * out.nb2, a);
}

static MatchJob make_job(const Prepared& a, const Prepared& b, int nA, int nB, int64_t out_off) {
    MatchJob j;
    j.PA = a.P;
    j.sqA = a.sq;
    j.PB = b.P;
    j.sqB = b.sq;
    j.nA = nA;
    j.nB = nB;
    j.out_off = out_off;
    j.AF = a.H;
    j.BF = b.H;
    j.dnA = a.dn;
    j.augB = b.aug;
    j.augresB = b.stat + 2;
    j.maxsqB = b.stat;
    j.maxdnB = b.stat + 1;
    j.AQ = a.QA;
    j.BQ = b.QB;
    j.dnqA = a.dnq;
    j.invsA = a.invs;
    j.sumqA = a.sumq;
    j.qstatB = b.stat + 4;
    j.AX = a.QX;
    j.BX = b.QX;
    j.ttA = a.tt;
    j.cinA = a.cin;
    j.cinB = b.cin;
    j.xstatA = reinterpret_cast<const unsigned*>(a.stat + 8);
    j.xstatB = reinterpret_cast<const unsigned*>(b.stat + 8);
    j.ncinB = (int)b.n_pad;
    j.BXs = b.BXs;
    j.cinBs = b.cins;
    j.tscB = b.tsc;
    j.permB = b.perm;
    return j;
}

// max |x| of one descriptor set into *d_slot (device); the caller reads all slots back in one copy
static void absmax_async(const float* X_dev, int64_t n, int64_t ld, int layout, float* d_slot) {
    if (n <= 0) return;
    const unsigned grid = std::min<unsigned>(cdiv((size_t)n * kDim, 256), 2048);
    if (layout == APS_ROWMAJOR && ld == kDim && (reinterpret_cast<uintptr_t>(X_dev) & 15) == 0)
        absmax_flat_kernel<<<std::min<unsigned>(cdiv((size_t)n * kDim / 4, 256), 512), 256, 0, stream()>>>(
            reinterpret_cast<const float4*>(X_dev), n * (kDim / 4), d_slot);
    else
        absmax_kernel<<<grid, 256, 0, stream()>>>(X_dev, n, ld, kDim, layout, d_slot);
    check_launch("absmax_kernel");
}

static float absmax(const float* X_dev, int64_t n, int64_t ld, int layout, float* d_slot) {
    APS_HIP(hipMemsetAsync(d_slot, 0, sizeof(float), stream()));
    absmax_async(X_dev, n, ld, layout, d_slot);
    float h = 0.f;
    APS_HIP(hipMemcpyAsync(&h, d_slot, sizeof(float), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    return h;
}

static void check_desc_args(const void* p, int64_t n, int64_t ld, int dim, int layout,
                            const char* name) {
    APS_REQUIRE(dim == kDim, APS_E_DIM, "%s: descriptor length %d not supported (this path is built for %d-D SIFT)",
                name, dim, kDim);
    APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "%s: unknown layout %d",
                name, layout);
    APS_REQUIRE(n >= 0, APS_E_ARG, "%s: negative row count", name);
    APS_REQUIRE(n == 0 || p != nullptr, APS_E_ARG, "%s: NULL data with %lld rows", name, (long long)n);
    if (layout == APS_ROWMAJOR)
        APS_REQUIRE(ld >= dim, APS_E_DIM, "%s: leading dimension %lld < dim %d", name, (long long)ld, dim);
    else
        APS_REQUIRE(ld >= n, APS_E_DIM, "%s: leading dimension %lld < rows %lld", name, (long long)ld,
                    (long long)n);
    APS_REQUIRE(n < (int64_t)1 << 31, APS_E_DIM, "%s: more than 2^31-1 rows", name);
}

// Matching mode: "split" (default) = bf16x3 candidate search + exact rescoring + exact-f32 fallback rows;
// "f32" = the all-f32 MFMA kernel on every row.  Both produce bit-identical (idx, d1, d2).
static bool use_split_path() {
    const char* e = std::getenv("APS_MATCH_MODE");
    return !(e && std::strcmp(e, "f32") == 0);
}

// job j's list segment (at its first output slot) -> its place in the pooled list
__global__ void fb_compact_kernel(const MatchJob* __restrict__ jobs, const uint32_t* __restrict__ fb_list,
                                  const unsigned int* __restrict__ fb_count, const long long* __restrict__ dst,
                                  uint32_t* __restrict__ pool) {
    const int j = blockIdx.x;
    const unsigned int cnt = fb_count[j];
    const uint32_t* src = fb_list + jobs[j].out_off;
    uint32_t* out = pool + dst[j];
    for (unsigned int e = threadIdx.x; e < cnt; e += blockDim.x) out[e] = src[e];
}

// entries [from[j], to[j]) of job j's list segment (at seg_off[j]) -> the pooled list at dst[j] ..., with their job ids
// (from == nullptr: from 0)
__global__ void list_pool_kernel(const int64_t* __restrict__ seg_off, const uint32_t* __restrict__ list, const unsigned int* __restrict__ from,
                                 const unsigned int* __restrict__ to, const long long* __restrict__ dst, uint32_t* __restrict__ pool,
                                 int* __restrict__ pool_job) {
    const int j = blockIdx.x;
    const unsigned int f0 = from ? from[j] : 0u, cnt = to[j] - f0;
    const uint32_t* src = list + seg_off[j] + f0;
    for (unsigned int e = threadIdx.x; e < cnt; e += blockDim.x) {
        pool[dst[j] + e] = src[e];
        pool_job[dst[j] + e] = j;
    }
}

// Pools the list segments of jobs that share a B set (same operand pointer and column count) into common tiles of kTMB
// rows: returns the tiles (WgJob{a job of the group, start in the pooled list, rows}); pool / pool_job are filled on the
// stream.  from / to: host copies of the per-job entry ranges (from may be empty = zeros); d_from / d_to: device copies.
static std::vector<WgJob> pool_lists(const std::vector<MatchJob>& jobs, const std::vector<int64_t>& seg_off,
                                     const uint32_t* d_list, const std::vector<unsigned int>& from, const std::vector<unsigned int>& to,
                                     const unsigned int* d_from, const unsigned int* d_to, Ws<uint32_t>& pool, Ws<int>& pool_job) {
    const size_t nj = jobs.size();
    std::vector<int> order(nj);
    for (size_t j = 0; j < nj; ++j) order[j] = (int)j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return std::less<const void*>()(jobs[a].BF, jobs[b].BF) || (jobs[a].BF == jobs[b].BF && jobs[a].nB < jobs[b].nB);
    });
    std::vector<long long> h_dst(nj, 0);
    std::vector<WgJob> tiles;
    size_t total = 0;
    for (size_t a = 0; a < nj;) {
        size_t b = a;
        const size_t g0 = total;
        while (b < nj && jobs[order[b]].BF == jobs[order[a]].BF && jobs[order[b]].nB == jobs[order[a]].nB) {
            const int j = order[b];
            h_dst[j] = (long long)total;
            total += to[j] - (from.empty() ? 0u : from[j]);
            ++b;
        }
        if (jobs[order[a]].nB > 0)
            for (size_t r = g0; r < total; r += kTMB) tiles.push_back({order[a], (int)r, (int)std::min<size_t>(kTMB, total - r), 0, 0, 0});
        a = b;
    }
    if (total == 0) return tiles;
    APS_REQUIRE(total < ((size_t)1 << 31), APS_E_DIM, "pooled row list too long (%zu)", total);
    pool.alloc(total);
    pool_job.alloc(total);
    Ws<long long> d_dst(nj);
    Ws<int64_t> d_seg(nj);
    APS_HIP(hipMemcpyAsync(d_dst, h_dst.data(), nj * sizeof(long long), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_seg, seg_off.data(), nj * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
    list_pool_kernel<<<(unsigned)nj, 128, 0, stream()>>>(d_seg, d_list, d_from, d_to, d_dst, pool, pool_job);
    check_launch("list_pool_kernel");
    APS_HIP(hipStreamSynchronize(stream()));  // h_dst / seg_off and their device copies go out of scope
    if (std::getenv("APS_POOL_DEBUG")) {  // consistency of the pooled list with the host's view of it
        std::vector<int> hj(total);
        std::vector<uint32_t> hr(total);
        std::vector<unsigned int> dto(nj);
        APS_HIP(hipMemcpy(hj.data(), pool_job, total * sizeof(int), hipMemcpyDeviceToHost));
        APS_HIP(hipMemcpy(hr.data(), pool, total * sizeof(uint32_t), hipMemcpyDeviceToHost));
        APS_HIP(hipMemcpy(dto.data(), d_to, nj * sizeof(unsigned int), hipMemcpyDeviceToHost));
        for (size_t j = 0; j < nj; ++j)
            if (dto[j] != to[j]) std::fprintf(stderr, "[aps pool] job %zu: device count %u, host count %u\n", j, dto[j], to[j]);
        size_t bad = 0;
        for (const WgJob& t : tiles)
            for (int e = 0; e < t.list_cnt; ++e) {
                const int j = hj[t.row0 + e];
                if (j < 0 || j >= (int)nj || jobs[j].BF != jobs[t.job].BF || hr[t.row0 + e] >= (uint32_t)jobs[j].nA) {
                    if (bad++ < 5) std::fprintf(stderr, "[aps pool] tile at %d entry %d: job %d row %u (tile job %d)\n", t.row0, e, j, hr[t.row0 + e], t.job);
                }
            }
        std::fprintf(stderr, "[aps pool] %zu entries in %zu tiles, %zu bad\n", total, tiles.size(), bad);
    }
    return tiles;
}

static thread_local int64_t g_screen_rows = 0, g_screen_surv = 0;  // aps_match_screen_stats
static thread_local int64_t g_screen_jobs = 0, g_screen_exact = 0;  // aps_match_screen_exact_jobs

// A/B switch: APS_SCREEN_SHAPE=32 runs the screening pass on v_mfma_i32_32x32x32_i8 (rounds 2-3) instead of 16x16x64
static bool screen_shape_32() {
    const char* e = std::getenv("APS_SCREEN_SHAPE");
    return e && std::atoi(e) == 32;
}

// The co-residency rule: waves of other kernels that shared a SIMD with int8-MFMA waves came back with
// different bits, so the screening kernels claim the SIMD's whole register file - 512 threads per workgroup = two waves
// per SIMD, 256 registers each (the `v_mov_b32 v255` in their first lines).  That holds only while the compiler really
// allocates 256: checked against the loaded code object before the first launch of a process, and by tests/test_abi.py
// against the code object's metadata on the CPU.
static void screen_regs(int shape32, int bounds, int* num_regs, int* max_threads) {
    hipFuncAttributes fa;
    const void* f = shape32 ? (bounds ? reinterpret_cast<const void*>(&match_screen_i8_kernel<true>) : reinterpret_cast<const void*>(&match_screen_i8_kernel<false>))
                            : (bounds ? reinterpret_cast<const void*>(&match_screen_i8x16_kernel<true>) : reinterpret_cast<const void*>(&match_screen_i8x16_kernel<false>));
    APS_HIP(hipFuncGetAttributes(&fa, f));
    *num_regs = fa.numRegs;
    *max_threads = fa.maxThreadsPerBlock;
}
static void require_whole_simd(int shape32, int bounds = 0) {
    static std::atomic<int> ok[4];  // (one code object for every device of the process: the answer is the same on all of them)
    if (ok[2 * shape32 + bounds].load(std::memory_order_relaxed)) return;
    int regs = 0, thr = 0;
    screen_regs(shape32, bounds, &regs, &thr);
    APS_REQUIRE((regs + 7) / 8 * 8 >= 256, APS_E_INTERNAL,
                "the int8 screening kernel holds %d registers per lane, not 256: other kernels' waves could share its SIMDs (DESIGN.md section 5)", regs);
    ok[2 * shape32 + bounds].store(1, std::memory_order_relaxed);
}
static void require_whole_simd_list() {  // (the same rule for the exact list pass)
    static std::atomic<int> ok{0};
    if (ok.load(std::memory_order_relaxed)) return;
    hipFuncAttributes fa;
    APS_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&match_list_i8_kernel)));
    APS_REQUIRE((fa.numRegs + 7) / 8 * 8 >= 256, APS_E_INTERNAL,
                "the int8 list kernel holds %d registers per lane, not 256: other kernels' waves could share its SIMDs (DESIGN.md section 5)", fa.numRegs);
    ok.store(1, std::memory_order_relaxed);
}
// tile k of a dense pass = rows [r, r + rows_per_tile) of job j, where off[j] <= k < off[j + 1] and r = (k - off[j]) * rows_per_tile
__global__ void expand_tiles_kernel(const int* __restrict__ off, int n_jobs, int n_tiles, int rows_per_tile, WgJob* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_tiles) return;
    int lo = 0, hi = n_jobs - 1;  // the largest j with off[j] <= k (jobs without rows have empty ranges)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= k) lo = mid; else hi = mid - 1;
    }
    out[k] = WgJob{lo, (k - off[lo]) * rows_per_tile, 0, 0, 0, 0};
}
// Runs the 2-NN search for a list of jobs whose operands are already prepared on the device.
// prune_r2 > 0: the caller will apply the ratio / threshold filter with these constants, so rows that cannot pass it
// may come back as idx 0 / inf without an exact evaluation (see match_cand_f16_kernel's tail)
static void run_match_jobs(const std::vector<MatchJob>& jobs, uint32_t* idx, float* d1, float* d2, float prune_r2 = 0.f,
                           float prune_thr = 0.f) {
    g_screen_rows = g_screen_surv = g_screen_jobs = g_screen_exact = 0;
    bool any_rows = false;
    for (const MatchJob& j : jobs) any_rows = any_rows || j.nA > 0;
    if (!any_rows) return;
    Ws<MatchJob> djobs(jobs.size());
    APS_HIP(hipMemcpyAsync(djobs, jobs.data(), jobs.size() * sizeof(MatchJob), hipMemcpyHostToDevice,
                           stream()));
    if (!use_split_path()) {
        // (the 128-row tile table of the all-f32 kernel: 312 k entries for the 64 x 4K scene - built and uploaded here only;
        // until round 4 every call paid for it, ~1 ms of host time and a 7.5 MB pageable upload the split path never read)
        std::vector<WgJob> wgs;
        for (int j = 0; j < (int)jobs.size(); ++j)
            for (int r = 0; r < jobs[j].nA; r += kTM) wgs.push_back({j, r, 0});
        Ws<WgJob> dwgs(wgs.size());
        APS_HIP(hipMemcpyAsync(dwgs, wgs.data(), wgs.size() * sizeof(WgJob), hipMemcpyHostToDevice,
                               stream()));
        {
            Prof prof("match2nn");
            match2nn_kernel<false><<<(unsigned)wgs.size(), 256, 0, stream()>>>(djobs, dwgs, nullptr, (int)jobs.size(), idx, d1, d2);
        }
        check_launch("match2nn_kernel");
        APS_HIP(hipStreamSynchronize(stream()));  // the pageable host vectors must outlive the copies
        return;
    }
    const int64_t total_rows = jobs.back().out_off + jobs.back().nA;
    // uncertified rows: one list segment per job (at the job's first output slot) and one counter per job, so the
    // host only reads the counters back - no sort, no second copy of the list
    Ws<uint32_t> fb_list((size_t)total_rows);
    Ws<unsigned int> fb_count(jobs.size());
    APS_HIP(hipMemsetAsync(fb_count, 0, jobs.size() * sizeof(unsigned int), stream()));
    // the dense tile table {job, first row} of the screening / candidate kernels, expanded on the device from the jobs' tile
    // offsets (round 6: 78 k entries for the 64 x 4K scene were built on the host and uploaded from pageable memory, ~0.2 ms)
    std::vector<int> wg_off(jobs.size() + 1, 0);
    for (size_t j = 0; j < jobs.size(); ++j) wg_off[j + 1] = wg_off[j] + (jobs[j].nA + kTMB - 1) / kTMB;
    struct { size_t n; size_t size() const { return n; } } bw{(size_t)wg_off.back()};
    Ws<WgJob> dbw(std::max<size_t>(bw.size(), 1));
    Ws<int> d_wg_off(wg_off.size());
    APS_HIP(hipMemcpyAsync(d_wg_off, wg_off.data(), wg_off.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
    if (bw.size() > 0) {
        expand_tiles_kernel<<<(unsigned)cdiv(bw.size(), 256), 256, 0, stream()>>>(d_wg_off, (int)jobs.size(), (int)bw.size(), kTMB, dbw);
        check_launch("expand_tiles_kernel");
    }
#ifdef APS_MATCH_TIMING  // ablation bits are honoured by timing builds only (they invalidate the results)
    const char* ab = std::getenv("APS_MATCH_ABLATE");
    const int ablate = ab ? std::atoi(ab) : 0;
#else
    const int ablate = 0;
#endif
    // Filtered callers: the int8 screen dismisses the rows that provably fail the filter, the candidate kernel then runs
    // on the survivors only (row lists per job, 512 to a workgroup).  APS_MATCH_NO_SCREEN=1: every row takes the f16 path.
    const bool screen = prune_r2 > 0.f && !std::getenv("APS_MATCH_NO_SCREEN");
    if (screen) {
        Ws<uint32_t> surv_list((size_t)total_rows);
        Ws<unsigned int> surv_count(2 * jobs.size());  // per job: survivors, then 1 where the job ran on exact integer codes
        APS_HIP(hipMemsetAsync(surv_count, 0, 2 * jobs.size() * sizeof(unsigned int), stream()));
        {
            Prof prof("match_screen_i8");
#ifdef APS_MATCH_TIMING
            sync_device_switch("APS_SCR_VARIANT", &g_scr_variant);
#endif
            require_whole_simd(screen_shape_32() ? 1 : 0);
            if (screen_shape_32())
                match_screen_i8_kernel<false><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), idx, d1, d2, surv_list,
                                                                                          surv_count, prune_r2, prune_thr, nullptr);
            else
                match_screen_i8x16_kernel<false><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), idx, d1, d2, surv_list,
                                                                                             surv_count, prune_r2, prune_thr, nullptr,
                                                                                             surv_count.get() + jobs.size());
        }
        check_launch("match_screen_i8_kernel");
        const auto S0 = std::chrono::steady_clock::now();
        std::vector<unsigned int> h_surv(2 * jobs.size());
        APS_HIP(hipMemcpyAsync(h_surv.data(), surv_count, 2 * jobs.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        g_screen_jobs = (int64_t)jobs.size();
        g_screen_exact = 0;
        std::vector<char> h_exact(jobs.size());
        for (size_t j = 0; j < jobs.size(); ++j) {
            h_exact[j] = h_surv[jobs.size() + j] ? 1 : 0;
            g_screen_exact += h_exact[j];
        }
        h_surv.resize(jobs.size());
        const auto S1 = std::chrono::steady_clock::now();
        // the survivors of the jobs that share a B set are pooled into common 512-row tiles (APS_MATCH_NO_POOL=1: one list per
        // job, as in rounds 2-3 - 6599 tiles instead of ~5600 for the 64 x 4K scene)
        const bool pooled = !std::getenv("APS_MATCH_NO_POOL");
        std::vector<WgJob> lw;
        size_t n_surv = 0;
        Ws<uint32_t> pool;
        Ws<int> pool_job;
        for (int j = 0; j < (int)jobs.size(); ++j) n_surv += h_surv[j];
        if (pooled) {
            std::vector<int64_t> seg(jobs.size());
            for (size_t j = 0; j < jobs.size(); ++j) seg[j] = jobs[j].out_off;
            // Round 6: the survivors of jobs that ran on exact integer codes take the exact int8 list pass (match_list_i8_kernel +
            // match_rescore_kernel), the others the f16 candidate kernel as before; each kind is pooled on its own.
            // APS_MATCH_NO_LIST_I8=1: the f16 kernel for all (A/B).
            std::vector<unsigned int> cnt_x(jobs.size(), 0u), cnt_g(h_surv);
            size_t n_x = 0;
            if (!std::getenv("APS_MATCH_NO_LIST_I8"))
                for (size_t j = 0; j < jobs.size(); ++j)
                    if (h_exact[j] && jobs[j].nB >= 1) {
                        cnt_x[j] = h_surv[j];
                        cnt_g[j] = 0u;
                        n_x += h_surv[j];
                    }
            if (n_x > 0) {
                Ws<unsigned int> d_cnt_x(jobs.size()), d_cnt_g(jobs.size());
                APS_HIP(hipMemcpyAsync(d_cnt_x, cnt_x.data(), jobs.size() * sizeof(unsigned int), hipMemcpyHostToDevice, stream()));
                APS_HIP(hipMemcpyAsync(d_cnt_g, cnt_g.data(), jobs.size() * sizeof(unsigned int), hipMemcpyHostToDevice, stream()));
                Ws<uint32_t> pool_x;
                Ws<int> pool_job_x;
                const std::vector<WgJob> lx = pool_lists(jobs, seg, surv_list, {}, cnt_x, nullptr, d_cnt_x, pool_x, pool_job_x);
                Ws<WgJob> dlx(lx.size());
                Ws<uint32_t> cand(n_x * kCandCap);
                Ws<unsigned int> cand_cnt(n_x);
                APS_HIP(hipMemcpyAsync(dlx, lx.data(), lx.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
                APS_HIP(hipMemsetAsync(cand_cnt, 0, n_x * sizeof(unsigned int), stream()));
                {
                    Prof prof("match_list_i8");
                    require_whole_simd_list();
                    match_list_i8_kernel<<<(unsigned)lx.size(), 512, 0, stream()>>>(djobs, dlx, (int)lx.size(), pool_x, pool_job_x, d1, cand, cand_cnt);
                }
                {
                    Prof prof("match_rescore");
                    match_rescore_kernel<<<(unsigned)cdiv(n_x, 32), 256, 0, stream()>>>(djobs, pool_x, pool_job_x, (int64_t)n_x, cand, cand_cnt, idx, d1, d2,
                                                                                       fb_list, fb_count);
                }
                check_launch("match_list_i8_kernel");
                lw = pool_lists(jobs, seg, surv_list, {}, cnt_g, nullptr, d_cnt_g, pool, pool_job);
                APS_HIP(hipStreamSynchronize(stream()));  // (lx, the count tables and the candidate lists go out of scope)
            } else {
                lw = pool_lists(jobs, seg, surv_list, {}, h_surv, nullptr, surv_count, pool, pool_job);
            }
        } else {
            for (int j = 0; j < (int)jobs.size(); ++j)
                for (unsigned int r = 0; r < h_surv[j]; r += kTMB) lw.push_back({j, (int)r, (int)std::min<unsigned int>(kTMB, h_surv[j] - r)});
        }
        g_screen_rows = total_rows;
        g_screen_surv = (int64_t)n_surv;
        if (std::getenv("APS_TRACE")) {
            int hist[5] = {0, 0, 0, 0, 0};  // jobs by surviving share: 0, <1 %, <10 %, <50 %, >= 50 %
            for (int j = 0; j < (int)jobs.size(); ++j) {
                const double f = jobs[j].nA ? (double)h_surv[j] / jobs[j].nA : 0.0;
                ++hist[h_surv[j] == 0 ? 0 : f < 0.01 ? 1 : f < 0.1 ? 2 : f < 0.5 ? 3 : 4];
            }
            std::fprintf(stderr, "[aps] int8 screen: %zu of %lld rows survive (%.2f %%), %zu list tiles; jobs by surviving share: none %d, <1%% %d, "
                         "<10%% %d, <50%% %d, >=50%% %d\n", n_surv, (long long)total_rows, 100.0 * (double)n_surv / (double)total_rows,
                         lw.size(), hist[0], hist[1], hist[2], hist[3], hist[4]);
        }
        if (!lw.empty()) {
            Ws<WgJob> dlw(lw.size());
            APS_HIP(hipMemcpyAsync(dlw, lw.data(), lw.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
            {
                Prof prof("match_cand_f16");
                match_cand_f16_kernel<true><<<(unsigned)lw.size(), 512, 0, stream()>>>(djobs, dlw, (int)lw.size(), idx, d1, d2, fb_list, fb_count,
                                                                                        ablate, prune_r2, prune_thr, nullptr, nullptr, nullptr,
                                                                                        pooled ? pool.get() : surv_list.get(),
                                                                                        pooled ? pool_job.get() : nullptr);
            }
            check_launch("match_cand_f16_kernel<list>");
            const auto S3 = std::chrono::steady_clock::now();
            APS_HIP(hipStreamSynchronize(stream()));  // lw must outlive its copy
            if (std::getenv("APS_TRACE")) {
                auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
                    return std::chrono::duration<double, std::milli>(b - a).count();
                };
                std::fprintf(stderr, "[aps] screen: launch -> counts on the host %.2f ms (incl. the wait for the kernel), tiles + pooling on the host %.2f ms, "
                             "list pass (wait) %.2f ms\n", ms(S0, S1), ms(S1, S3), ms(S3, std::chrono::steady_clock::now()));
            }
        }
    } else {
        Prof prof("match_cand_f16");
        match_cand_f16_kernel<false><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), idx, d1, d2, fb_list, fb_count,
                                                                                 ablate, prune_r2, prune_thr, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    check_launch("match_cand_f16_kernel");
    std::vector<unsigned int> h_cnt(jobs.size());
    const auto R0 = std::chrono::steady_clock::now();
    APS_HIP(hipMemcpyAsync(h_cnt.data(), fb_count, jobs.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    const auto R1 = std::chrono::steady_clock::now();
    // Exact f32 kernel in row-list mode.  The uncertified rows of all jobs that share a B set are pooled into common
    // 128-row tiles (a pair leaves ~80 such rows at 20 k features: one mostly empty tile each otherwise): the segments
    // are compacted in group order on the device, the kernel finds every row's job by its output slot.
    std::vector<int> order((size_t)jobs.size());
    for (int j = 0; j < (int)jobs.size(); ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        return std::less<const void*>()(jobs[a].PB, jobs[b].PB) ||
               (jobs[a].PB == jobs[b].PB && jobs[a].nB < jobs[b].nB);
    });
    std::vector<long long> h_dst(jobs.size(), 0);
    std::vector<WgJob> fwgs;
    size_t n_fb = 0;
    for (size_t a = 0; a < order.size();) {
        size_t b = a;
        const size_t g0 = n_fb;
        while (b < order.size() && jobs[order[b]].PB == jobs[order[a]].PB && jobs[order[b]].nB == jobs[order[a]].nB) {
            h_dst[order[b]] = (long long)n_fb;
            n_fb += h_cnt[order[b]];
            ++b;
        }
        for (size_t r = g0; r < n_fb; r += kTM) fwgs.push_back({order[a], (int)r, (int)std::min<size_t>(kTM, n_fb - r), 0, 0, 0});
        a = b;
    }
    if (fwgs.empty()) return;
    // A handful of tiles, each streaming a whole B set through f32 MFMAs, leaves most of the chip idle (63 tiles of ~32
    // rows on the 64 x 4K scene: 1.5 ms): split the columns of every tile into parts and merge the parts' top-2.
    int n_parts = 1;
    if (!std::getenv("APS_MATCH_NO_FB_SPLIT")) n_parts = (int)std::min<size_t>(8, std::max<size_t>(1, 512 / fwgs.size()));
    if (n_parts > 1) {
        std::vector<WgJob> split;
        for (const WgJob& f : fwgs) {
            const int nt = (jobs[f.job].nB + kTN - 1) / kTN;
            const int np = std::min(n_parts, std::max(nt, 1));
            for (int p = 0; p < n_parts; ++p) {  // (every part index exists for every row: parts beyond np cover no tile)
                WgJob g = f;
                g.t0 = p < np ? (int)((long long)nt * p / np) : nt;
                g.t1 = p < np ? (int)((long long)nt * (p + 1) / np) : nt;
                g.part = p;
                split.push_back(g);
            }
        }
        fwgs.swap(split);
    }
    Ws<long long> d_dst(jobs.size());
    Ws<uint32_t> fb_pool(n_fb);
    Ws<WgJob> dfw(fwgs.size());
    APS_HIP(hipMemcpyAsync(d_dst, h_dst.data(), jobs.size() * sizeof(long long), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(dfw, fwgs.data(), fwgs.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
    {
        Prof prof("match2nn_fallback");
        fb_compact_kernel<<<(unsigned)jobs.size(), 64, 0, stream()>>>(djobs, fb_list, fb_count, d_dst, fb_pool);
        if (n_parts > 1) {
            Ws<uint32_t> p_idx(n_fb * n_parts);
            Ws<float> p_d1(n_fb * n_parts), p_d2(n_fb * n_parts);
            match2nn_kernel<true><<<(unsigned)fwgs.size(), 256, 0, stream()>>>(djobs, dfw, fb_pool, (int)jobs.size(), p_idx, p_d1, p_d2,
                                                                              (int64_t)n_fb);
            match2nn_merge_parts_kernel<<<cdiv(n_fb, 256), 256, 0, stream()>>>(fb_pool, (int64_t)n_fb, n_parts, p_idx, p_d1, p_d2, idx, d1, d2);
        } else {
            match2nn_kernel<true><<<(unsigned)fwgs.size(), 256, 0, stream()>>>(djobs, dfw, fb_pool, (int)jobs.size(), idx, d1, d2);
        }
    }
    check_launch("match2nn_kernel<list>");
    const auto R2 = std::chrono::steady_clock::now();
    APS_HIP(hipStreamSynchronize(stream()));
    if (std::getenv("APS_TRACE"))
        std::fprintf(stderr, "[aps] 2-NN: candidates (wait) %.2f ms, fallback list %zu rows: host %.2f ms, kernel (wait) %.2f ms\n",
                     std::chrono::duration<double, std::milli>(R1 - R0).count(), n_fb,
                     std::chrono::duration<double, std::milli>(R2 - R1).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - R2).count());
}

// ratio/threshold/unique for a list of jobs; returns total kept.  Outputs are device pointers.
// job_of[e] for the pooled fallback list: entry e belongs to job j for dst[j] <= e < dst[j] + cnt[j]
__global__ void fb_jobs_kernel(const long long* __restrict__ dst, const unsigned int* __restrict__ fb_count, int n_jobs,
                               int* __restrict__ job_of) {
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const unsigned int cnt = fb_count[j];
    for (unsigned int e = threadIdx.x; e < cnt; e += blockDim.x) job_of[dst[j] + e] = j;
}

}  // namespace aps (reopened below)

namespace aps {

// The screening half of the blocked global k-NN (aps_knn_global on a pool against itself, knn.hip): the pool is cut into
// blocks; for every ordered pair of DIFFERENT blocks (i, j) the candidate kernel finds, for each row of block i, its three
// nearest rows of block j (exact canonical distances, ascending by (distance, index)) and a bound below which no other
// row of block j lies.  Rows the f16 screen cannot certify are recomputed exactly (knn3_rows_kernel).
// Outputs are indexed by slot = job_off[i * nb + j] + local row:  t3_idx[3 slot + e] (1-based within block j, 0 = none),
// t3_d[3 slot + e], t3_b[slot].  Returns the slot count; call with t3_* == nullptr to get it first.
int64_t screened_block_top3(const float* X_dev, int64_t ld, int layout, const std::vector<int64_t>& block_off,
                            std::vector<int64_t>& job_off, uint32_t* t3_idx, float* t3_d, float* t3_b) {
    const int nb = (int)block_off.size() - 1;
    job_off.assign((size_t)nb * nb, -1);
    int64_t slots = 0;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            if (i == j) continue;
            job_off[(size_t)i * nb + j] = slots;
            slots += block_off[i + 1] - block_off[i];
        }
    if (!t3_idx || slots == 0) return slots;
    APS_REQUIRE(slots < ((int64_t)1 << 31), APS_E_DIM, "too many (row, block) pairs for one pass (%lld)", (long long)slots);
    std::vector<Prepared> prep(nb);
    for (int b = 0; b < nb; ++b) {
        const float* xb = layout == APS_ROWMAJOR ? X_dev + (size_t)block_off[b] * ld : X_dev + block_off[b];
        prepare(xb, block_off[b + 1] - block_off[b], ld, layout, false, prep[b]);
    }
    std::vector<MatchJob> jobs;
    std::vector<WgJob> bw;
    for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) {
            if (i == j) continue;
            const int nA = (int)(block_off[i + 1] - block_off[i]), nB = (int)(block_off[j + 1] - block_off[j]);
            for (int r = 0; r < nA; r += kTMB) bw.push_back({(int)jobs.size(), r, 0});
            jobs.push_back(make_job(prep[i], prep[j], nA, nB, job_off[(size_t)i * nb + j]));
        }
    Ws<MatchJob> djobs(jobs.size());
    Ws<WgJob> dbw(bw.size());
    Ws<uint32_t> fb_list((size_t)slots);
    Ws<unsigned int> fb_count(jobs.size());
    APS_HIP(hipMemcpyAsync(djobs, jobs.data(), jobs.size() * sizeof(MatchJob), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(dbw, bw.data(), bw.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemsetAsync(fb_count, 0, jobs.size() * sizeof(unsigned int), stream()));
    {
        Prof prof("match_cand_f16");
        match_cand_f16_kernel<false><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), nullptr, nullptr, nullptr, fb_list,
                                                                                 fb_count, 0, 0.f, 0.f, t3_idx, t3_d, t3_b, nullptr, nullptr);
    }
    check_launch("match_cand_f16_kernel");
    std::vector<unsigned int> h_cnt(jobs.size());
    APS_HIP(hipMemcpyAsync(h_cnt.data(), fb_count, jobs.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    std::vector<long long> h_dst(jobs.size());
    size_t n_fb = 0;
    for (size_t j = 0; j < jobs.size(); ++j) {
        h_dst[j] = (long long)n_fb;
        n_fb += h_cnt[j];
    }
    if (n_fb) {
        Ws<long long> d_dst(jobs.size());
        Ws<uint32_t> pool(n_fb);
        Ws<int> job_of(n_fb);
        APS_HIP(hipMemcpyAsync(d_dst, h_dst.data(), jobs.size() * sizeof(long long), hipMemcpyHostToDevice, stream()));
        Prof prof("knn_fallback3");
        fb_compact_kernel<<<(unsigned)jobs.size(), 64, 0, stream()>>>(djobs, fb_list, fb_count, d_dst, pool);
        fb_jobs_kernel<<<(unsigned)jobs.size(), 64, 0, stream()>>>(d_dst, fb_count, (int)jobs.size(), job_of);
        knn3_rows_kernel<<<(unsigned)n_fb, 64, 0, stream()>>>(djobs, job_of, pool, (int)n_fb, t3_idx, t3_d, t3_b);
        check_launch("knn3_rows_kernel");
    }
    APS_HIP(hipStreamSynchronize(stream()));  // the host tables and the Prepared blocks must outlive the launches
    if (std::getenv("APS_TRACE"))
        std::fprintf(stderr, "[aps] blocked k-NN screen: %d blocks, %zu jobs, %lld slots, %zu uncertified (%.3f %%)\n", nb, jobs.size(),
                     (long long)slots, n_fb, 100.0 * (double)n_fb / (double)slots);
    return slots;
}

// ------------------------------------------------------------------------------------------------
// the pooled matcher (featureMatchingGlobal.m) with its filter in view
// ------------------------------------------------------------------------------------------------
// featureMatchingGlobal keeps query q only if, among its k nearest rows of the pool with q itself and the rows of q's own
// image taken out, at least two remain and dist(first) / max(dist(second), eps) <= ratio (:129-147).  Blocks = images.
// For every OTHER image j the int8 screen gives, per row, L1(j) <= the smallest distance into image j and H1(j), H2(j) >=
// its smallest / second smallest one.  Let c1 <= c2 be the two smallest cross-image distances of q: c1 >= min_j L1(j) and
// c2 <= min(second smallest H1(j), min_j H2(j)).  Whatever the k nearest are, the two cross-image rows the ratio test
// compares are no nearer than c1 and c2 - so  min_j L1(j) > ratio * max(that upper bound, eps)  proves that q is dropped,
// and only the other rows need their exact neighbours.
// A wave's rows lie in at most two images of >= 64 rows (rows are pooled image by image): the lanes of image `iw` append
// to job (iw, j) with one atomic per wave.  `keep` must be false for lanes of another image.
__device__ __forceinline__ void global_list_append(bool keep, int job, int64_t r, const int64_t* __restrict__ job_off,
                                                   uint32_t* __restrict__ row_list, unsigned int* __restrict__ list_count) {
    const unsigned long long m = __ballot(keep);
    if (!m) return;
    const int lane = threadIdx.x & 63;
    unsigned int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&list_count[job], (unsigned int)__popcll(m));
    base = __shfl(base, leader);
    if (keep) row_list[job_off[job] + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)r;
}

// Phase 0 of the pooled search: per row the verdict of the proof (dismissed), the bound-based cut (three distinct rows of
// other images lie within it), and the FIRST images to search: the row's own image and the (up to) three images with the
// smallest upper bound H1 - where its nearest rows most likely are.  Their exact distances then replace the cut's upper
// bounds (phase B, global_phase_b_kernel).
__global__ __launch_bounds__(256) void global_screen_reduce_kernel(const float* __restrict__ bounds, const int64_t* __restrict__ img_off,
                                                                   int n_img, const int64_t* __restrict__ job_off, int64_t f, float ratio,
                                                                   uint8_t* __restrict__ dismissed, float* __restrict__ cut_out,
                                                                   int* __restrict__ first_imgs /* f x 3 */, uint32_t* __restrict__ row_list,
                                                                   unsigned int* __restrict__ list_count, int64_t q_lo) {
    const int64_t q = q_lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // this call's rows: [q_lo, f)
    const bool live = q < f;
    const int64_t qc = live ? q : f - 1;
    int lo = 0, hi = n_img - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (img_off[mid] <= qc) lo = mid; else hi = mid - 1;
    }
    const int i = lo;
    const int64_t r = qc - img_off[i];
    float lmin = INFINITY, h2m = INFINITY;
    float u0 = INFINITY, u1 = INFINITY, u2 = INFINITY;  // three smallest of {H1(j), H2(j)}: distinct columns each
    float g0 = INFINITY, g1 = INFINITY, g2 = INFINITY;  // three smallest H1(j) ...
    int a0 = -1, a1 = -1, a2 = -1;                      // ... and their images
    auto put3 = [&](float h) {
        if (h < u2) {
            if (h < u1) {
                u2 = u1;
                if (h < u0) {
                    u1 = u0;
                    u0 = h;
                } else {
                    u1 = h;
                }
            } else {
                u2 = h;
            }
        }
    };
    for (int j = 0; j < n_img; ++j) {
        if (j == i || img_off[j + 1] == img_off[j]) continue;
        const float* b = bounds + (size_t)(job_off[(size_t)i * n_img + j] + r) * 3;
        const float l1 = b[0], h1 = b[1], h2 = b[2];
        lmin = fminf(lmin, l1);
        h2m = fminf(h2m, h2);
        put3(h1);
        put3(h2);
        if (h1 < g2) {
            if (h1 < g1) {
                g2 = g1, a2 = a1;
                if (h1 < g0) {
                    g1 = g0, a1 = a0;
                    g0 = h1, a0 = j;
                } else {
                    g1 = h1, a1 = j;
                }
            } else {
                g2 = h1, a2 = j;
            }
        }
    }
    const float hsec = fmaxf(fminf(g1, h2m), 1.1920929e-07f);  // (the reference divides by max(second, eps('single')))
    const bool drop = live && hsec < INFINITY && lmin > -INFINITY && lmin < INFINITY &&
                      lmin * (1.0f - 1e-5f) - 1e-30f > ratio * hsec * (1.0f + 1e-5f);
    if (live) {
        dismissed[q] = drop ? 1 : 0;
        cut_out[q] = u2 < INFINITY ? u2 * (1.0f + 1e-5f) + 1e-30f : INFINITY;
        first_imgs[q * 3] = a0;
        first_imgs[q * 3 + 1] = a1;
        first_imgs[q * 3 + 2] = a2;
    }
    const int i_first = __shfl(i, 0), i_last = __shfl(i, 63);
    const bool edge = i != i_first && i != i_last;  // (an image of fewer than 64 rows inside one wave: plain atomics)
    for (int pass = 0; pass < 2; ++pass) {
        const int iw = pass ? i_last : i_first;
        if (pass && i_last == i_first) break;
        for (int j = 0; j < n_img; ++j) {
            if (img_off[j + 1] == img_off[j]) continue;
            const bool keep = live && !drop && i == iw && (j == i || j == a0 || j == a1 || j == a2);
            global_list_append(keep, iw * n_img + j, r, job_off, row_list, list_count);
        }
    }
    if (live && !drop && edge) {
        const int js[4] = {i, a0, a1, a2};
        for (int e = 0; e < 4; ++e)
            if (js[e] >= 0) row_list[job_off[i * n_img + js[e]] + atomicAdd(&list_count[i * n_img + js[e]], 1u)] = (uint32_t)r;
    }
}

// Phase B: with the exact distances of phase A's candidates the third nearest non-self row found so far replaces the
// bound-based cut where it is smaller, and every image not searched yet is listed for the row only if its lower bound
// does not exceed that cut - a row of such an image could still be among the row's four nearest.
__global__ __launch_bounds__(256) void global_phase_b_kernel(const float* __restrict__ bounds, const int64_t* __restrict__ img_off, int n_img,
                                                             const int64_t* __restrict__ job_off, int64_t f, float ratio,
                                                             uint8_t* __restrict__ dismissed, const float* __restrict__ cut_in,
                                                             const int* __restrict__ first_imgs, const uint32_t* __restrict__ t3_idx,
                                                             const float* __restrict__ t3_d, uint32_t* __restrict__ row_list,
                                                             unsigned int* __restrict__ list_count, int64_t q_lo) {
    const int64_t q = q_lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // this call's rows: [q_lo, f)
    const bool live = q < f;
    const int64_t qc = live ? q : f - 1;
    int lo = 0, hi = n_img - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (img_off[mid] <= qc) lo = mid; else hi = mid - 1;
    }
    const int i = lo;
    const int64_t r = qc - img_off[i];
    bool act = live && !dismissed[qc];
    const int a0 = first_imgs[qc * 3], a1 = first_imgs[qc * 3 + 1], a2 = first_imgs[qc * 3 + 2];
    float cut = cut_in[qc];
    if (act) {
        // the three smallest exact distances to rows other than q itself among phase A's (certified) candidates, and the
        // two smallest among those of OTHER images
        float x0 = INFINITY, x1 = INFINITY, x2 = INFINITY, c0 = INFINITY, c1 = INFINITY;
        const int js[4] = {i, a0, a1, a2};
        for (int e = 0; e < 4; ++e) {
            const int j = js[e];
            if (j < 0) continue;
            const int64_t slot = job_off[(size_t)i * n_img + j] + r;
            for (int c = 0; c < 3; ++c) {
                const uint32_t li = t3_idx[slot * 3 + c];
                if (!li || (j == i && (int64_t)li - 1 == r)) continue;  // none / the row itself
                const float d = t3_d[slot * 3 + c];
                if (j != i) {
                    if (d < c0) {
                        c1 = c0;
                        c0 = d;
                    } else if (d < c1) {
                        c1 = d;
                    }
                }
                if (d < x2) {
                    if (d < x1) {
                        x2 = x1;
                        if (d < x0) {
                            x1 = x0;
                            x0 = d;
                        } else {
                            x1 = d;
                        }
                    } else {
                        x2 = d;
                    }
                }
            }
        }
        if (x2 < INFINITY) cut = fminf(cut, x2 * (1.0f + 1e-6f) + 1e-30f);
        // the proof of global_screen_reduce_kernel once more, with the EXACT second cross-image distance found so far in
        // place of its upper bound: c2 <= c1 (found), c1 (true) >= min_j L1(j)
        if (c1 < INFINITY) {
            float lmin = INFINITY;
            for (int j = 0; j < n_img; ++j) {
                if (j == i || img_off[j + 1] == img_off[j]) continue;
                lmin = fminf(lmin, bounds[(size_t)(job_off[(size_t)i * n_img + j] + r) * 3]);
            }
            const float hsec = fmaxf(c1, 1.1920929e-07f);
            // (lmin == +inf: no bound at all - the unscreened mode writes NaN bounds, which fminf skips)
            if (lmin > -INFINITY && lmin < INFINITY && lmin * (1.0f - 1e-5f) - 1e-30f > ratio * hsec * (1.0f + 1e-5f)) {
                dismissed[qc] = 1;
                act = false;
            }
        }
    }
    const int i_first = __shfl(i, 0), i_last = __shfl(i, 63);
    const bool edge = i != i_first && i != i_last;
    for (int pass = 0; pass < 2; ++pass) {
        const int iw = pass ? i_last : i_first;
        if (pass && i_last == i_first) break;
        for (int j = 0; j < n_img; ++j) {
            if (img_off[j + 1] == img_off[j]) continue;
            bool keep = act && i == iw && j != i && j != a0 && j != a1 && j != a2;
            if (keep) {
                const float l1 = bounds[(size_t)(job_off[(size_t)i * n_img + j] + r) * 3];
                keep = !(l1 * (1.0f - 1e-5f) - 1e-30f > cut);  // (NaN / -inf bounds keep the image)
            }
            global_list_append(keep, iw * n_img + j, r, job_off, row_list, list_count);
        }
    }
    if (act && edge) {
        for (int j = 0; j < n_img; ++j) {
            if (img_off[j + 1] == img_off[j] || j == i || j == a0 || j == a1 || j == a2) continue;
            const float l1 = bounds[(size_t)(job_off[(size_t)i * n_img + j] + r) * 3];
            if (!(l1 * (1.0f - 1e-5f) - 1e-30f > cut)) row_list[job_off[i * n_img + j] + atomicAdd(&list_count[i * n_img + j], 1u)] = (uint32_t)r;
        }
    }
}

// (row, image) slots that are not searched: no candidates, and nothing unlisted there can matter (bound = +inf)
__global__ void global_t3_init_kernel(uint32_t* __restrict__ t3_idx, float* __restrict__ t3_d, float* __restrict__ t3_b, int64_t slots) {
    const int64_t s = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (s >= slots) return;
    t3_idx[s * 3] = t3_idx[s * 3 + 1] = t3_idx[s * 3 + 2] = 0u;
    t3_d[s * 3] = t3_d[s * 3 + 1] = t3_d[s * 3 + 2] = INFINITY;
    t3_b[s] = INFINITY;
}

struct GlobalPrep {
    std::vector<Prepared> prep;
    Ws<float> stats;
    bool ready = false;
};
GlobalPrep* global_prep_new() { return new GlobalPrep(); }
void global_prep_free(GlobalPrep* p) { delete p; }

// X_dev: the (normalised) pool; img_off: n + 1 row offsets of the images.  Slots: job (i, j), EVERY j (the diagonal too:
// the rows of q's own image and q itself are candidates of the k nearest), slot = job_off[i n + j] + local row.
// Outputs t3_* as screened_block_top3 for the rows with dismissed[q] == 0 (the other slots are not written).
// Returns the slot count; call with t3_idx == nullptr to get it (and job_off) first.
int64_t screened_global_top3(const float* X_dev, int64_t ld, int layout, const std::vector<int64_t>& img_off, float ratio,
                             std::vector<int64_t>& job_off, uint32_t* t3_idx, float* t3_d, float* t3_b, uint8_t* dismissed,
                             int64_t* n_survivors, int img_a, int img_b, GlobalPrep* keep) {
    // [img_a, img_b): the QUERY images of this call (every image is a column set); a pool whose (row, image) table exceeds
    // 2^31 slots - BASELINE configs[4]: 500 images, 5.4 M rows - is searched in several calls over ranges of query images
    const int n = (int)img_off.size() - 1;
    if (img_b < 0) img_b = n;
    const int64_t q_lo = img_off[img_a], f = img_off[img_b];  // this call's rows (f: their end)
    job_off.assign((size_t)n * n, 0);
    int64_t slots = 0;
    for (int i = img_a; i < img_b; ++i)
        for (int j = 0; j < n; ++j) {
            job_off[(size_t)i * n + j] = slots;
            slots += img_off[i + 1] - img_off[i];
        }
    if (!t3_idx || slots == 0) return slots;
    APS_REQUIRE(slots < ((int64_t)1 << 31), APS_E_DIM, "too many (row, image) pairs for one pass (%lld)", (long long)slots);
    GlobalPrep local;
    GlobalPrep& G = keep ? *keep : local;
    std::vector<Prepared>& prep = G.prep;
    Ws<float>& prep_stats = G.stats;
    if (!G.ready) {  // the images' operand forms: every set in one batch of launches (as in match_pairs_impl)
        prep.resize(n);
        prep_stats.alloc((size_t)kStatWords * std::max(n, 1));
        APS_HIP(hipMemsetAsync(prep_stats, 0, (size_t)kStatWords * std::max(n, 1) * sizeof(float), stream()));
        std::vector<PrepRequest> req;
        for (int b = 0; b < n; ++b) {
            const float* xb = layout == APS_ROWMAJOR ? X_dev + (size_t)img_off[b] * ld : X_dev + img_off[b];
            req.push_back({xb, img_off[b + 1] - img_off[b], ld, false, &prep[b], prep_stats.get() + kStatWords * b});
        }
        prepare_batch(req, layout);
        G.ready = true;
    }
    // ---- the int8 screen over every ordered pair of different images: bounds per (row, image) ----
    std::vector<MatchJob> jobs;
    std::vector<WgJob> bw;
    for (int i = img_a; i < img_b; ++i)
        for (int j = 0; j < n; ++j) {
            const int nA = (int)(img_off[i + 1] - img_off[i]), nB = (int)(img_off[j + 1] - img_off[j]);
            if (i == j || nA == 0 || nB == 0) continue;
            for (int r = 0; r < nA; r += kTMB) bw.push_back({(int)jobs.size(), r, 0});
            jobs.push_back(make_job(prep[i], prep[j], nA, nB, job_off[(size_t)i * n + j]));
        }
    Ws<float> bounds((size_t)slots * 3);
    Ws<int64_t> d_ioff(n + 1), d_joff((size_t)n * n);
    Ws<uint32_t> row_list((size_t)slots);
    Ws<unsigned int> list_count((size_t)n * n);
    APS_HIP(hipMemcpyAsync(d_ioff, img_off.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_joff, job_off.data(), (size_t)n * n * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemsetAsync(list_count, 0, (size_t)n * n * sizeof(unsigned int), stream()));
    const bool screen = !std::getenv("APS_MATCH_NO_SCREEN") && !jobs.empty();
    if (screen) {
        Ws<MatchJob> djobs(jobs.size());
        Ws<WgJob> dbw(bw.size());
        APS_HIP(hipMemcpyAsync(djobs, jobs.data(), jobs.size() * sizeof(MatchJob), hipMemcpyHostToDevice, stream()));
        APS_HIP(hipMemcpyAsync(dbw, bw.data(), bw.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
        {
            Prof prof("match_screen_i8_bounds");
            require_whole_simd(screen_shape_32() ? 1 : 0, 1);
            if (screen_shape_32())
                match_screen_i8_kernel<true><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), nullptr, nullptr, nullptr,
                                                                                         nullptr, nullptr, 0.f, 0.f, bounds);
            else
                match_screen_i8x16_kernel<true><<<(unsigned)bw.size(), 512, 0, stream()>>>(djobs, dbw, (int)bw.size(), nullptr, nullptr, nullptr,
                                                                                            nullptr, nullptr, 0.f, 0.f, bounds, nullptr);
        }
        check_launch("match_screen_i8_kernel (bounds)");
        APS_HIP(hipStreamSynchronize(stream()));  // djobs / dbw go out of scope
    } else {
        // no screen: NaN bounds - every comparison fails, every row survives and is searched in every image
        APS_HIP(hipMemsetAsync(bounds, 0xff, (size_t)slots * 3 * sizeof(float), stream()));
    }
    Ws<float> cut((size_t)f);
    Ws<int> first_imgs((size_t)f * 3);
    {
        Prof prof("global_screen_reduce");
        global_screen_reduce_kernel<<<cdiv(f - q_lo, 256), 256, 0, stream()>>>(bounds, d_ioff, n, d_joff, f, ratio, dismissed, cut, first_imgs,
                                                                               row_list, list_count, q_lo);
        global_t3_init_kernel<<<cdiv(slots, 256), 256, 0, stream()>>>(t3_idx, t3_d, t3_b, slots);
    }
    check_launch("global_screen_reduce_kernel");
    std::vector<unsigned int> h_a((size_t)n * n), h_b((size_t)n * n);
    APS_HIP(hipMemcpyAsync(h_a.data(), list_count, (size_t)n * n * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    int64_t n_surv = 0;
    for (int i = 0; i < n; ++i) n_surv += h_a[(size_t)i * n + i];  // (the diagonal job lists every surviving row of image i)
    if (n_survivors) *n_survivors = n_surv;
    if (n_surv == 0) return slots;
    // ---- the exact three nearest rows (with a certified bound) of the listed (row, image) pairs, in two phases ----
    std::vector<MatchJob> cj;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            cj.push_back(make_job(prep[i], prep[j], (int)(img_off[i + 1] - img_off[i]), (int)(img_off[j + 1] - img_off[j]),
                                  job_off[(size_t)i * n + j]));  // (job index == i n + j)
    Ws<MatchJob> dcj(cj.size());
    Ws<uint32_t> fb_list(1);
    Ws<unsigned int> fb_count(cj.size());
    APS_HIP(hipMemcpyAsync(dcj, cj.data(), cj.size() * sizeof(MatchJob), hipMemcpyHostToDevice, stream()));
    const bool pooled = !std::getenv("APS_MATCH_NO_POOL");
    Ws<unsigned int> d_from((size_t)n * n);
    auto search = [&](const std::vector<unsigned int>& from, const std::vector<unsigned int>& to) {  // list entries [from, to) of every job
        std::vector<WgJob> lw;
        Ws<uint32_t> pool;
        Ws<int> pool_job;
        if (pooled) {  // the lists of the jobs (i, j) that share the column image j in common tiles (list_count holds `to`)
            APS_HIP(hipMemcpyAsync(d_from, from.data(), (size_t)n * n * sizeof(unsigned int), hipMemcpyHostToDevice, stream()));
            lw = pool_lists(cj, job_off, row_list, from, to, d_from, list_count, pool, pool_job);
        } else {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    if (img_off[i + 1] == img_off[i] || img_off[j + 1] == img_off[j]) continue;
                    const size_t job = (size_t)i * n + j;
                    for (unsigned int r = from[job]; r < to[job]; r += kTMB) lw.push_back({(int)job, (int)r, (int)std::min<unsigned int>(kTMB, to[job] - r)});
                }
        }
        if (lw.empty()) return;
        Ws<WgJob> dlw(lw.size());
        APS_HIP(hipMemcpyAsync(dlw, lw.data(), lw.size() * sizeof(WgJob), hipMemcpyHostToDevice, stream()));
        {
            Prof prof("match_cand_f16");
            match_cand_f16_kernel<true><<<(unsigned)lw.size(), 512, 0, stream()>>>(dcj, dlw, (int)lw.size(), nullptr, nullptr, nullptr, fb_list,
                                                                                    fb_count, 0, 0.f, 0.f, t3_idx, t3_d, t3_b,
                                                                                    pooled ? pool.get() : row_list.get(),
                                                                                    pooled ? pool_job.get() : nullptr);
        }
        check_launch("match_cand_f16_kernel (pooled, list mode)");
        APS_HIP(hipStreamSynchronize(stream()));  // lw / dlw go out of scope
    };
    const std::vector<unsigned int> zero((size_t)n * n, 0u);
    search(zero, h_a);  // phase A: the own image and the three most promising ones
    {
        Prof prof("global_screen_reduce");
        global_phase_b_kernel<<<cdiv(f - q_lo, 256), 256, 0, stream()>>>(bounds, d_ioff, n, d_joff, f, ratio, dismissed, cut, first_imgs, t3_idx,
                                                                         t3_d, row_list, list_count, q_lo);
    }
    check_launch("global_phase_b_kernel");
    APS_HIP(hipMemcpyAsync(h_b.data(), list_count, (size_t)n * n * sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    bounds.reset();
    int64_t n_a = 0, n_b = 0;
    for (size_t job = 0; job < (size_t)n * n; ++job) {
        n_a += h_a[job];
        n_b += h_b[job] - h_a[job];
    }
    if (std::getenv("APS_TRACE"))
        std::fprintf(stderr, "[aps] pooled matcher screen: %lld of %lld rows survive (%.2f %%); (row, image) searches: %lld first + %lld more of %lld (%.2f %%)\n",
                     (long long)n_surv, (long long)(f - q_lo), 100.0 * (double)n_surv / (double)std::max<int64_t>(f - q_lo, 1), (long long)n_a, (long long)n_b,
                     (long long)slots, 100.0 * (double)(n_a + n_b) / (double)slots);
    search(h_a, h_b);  // phase B: every other image that can still hold one of the four nearest
    return slots;
}

static int64_t run_filter(const std::vector<FilterJob>& fjobs, int64_t total_rows, int64_t total_cols,
                          const uint32_t* idx, const float* d1, const float* d2,
                          const aps_match_opts& o, unsigned long long* d_job_ptr /* njobs+1 */,
                          uint32_t* o1, uint32_t* o2, float* metric, int64_t cap) {
    const int njobs = (int)fjobs.size();
    APS_HIP(hipMemsetAsync(d_job_ptr, 0, (njobs + 1) * sizeof(unsigned long long), stream()));
    if (total_rows == 0 || njobs == 0) return 0;
    Ws<FilterJob> dfj(njobs);
    APS_HIP(hipMemcpyAsync(dfj, fjobs.data(), njobs * sizeof(FilterJob), hipMemcpyHostToDevice,
                           stream()));
    Ws<unsigned long long> keys(total_rows), packed(total_rows), sorted(total_rows),
        winner(std::max<int64_t>(total_cols, 1)), cnt(njobs + 1);
    APS_HIP(hipMemsetAsync(winner, 0xff, std::max<int64_t>(total_cols, 1) * sizeof(unsigned long long),
                           stream()));
    APS_HIP(hipMemsetAsync(cnt, 0, (njobs + 1) * sizeof(unsigned long long), stream()));
    const double r2 = o.max_ratio * o.max_ratio;  // opt.MaxRatio^2 in double (matchFeaturesScratch.m:170-173)
    Prof prof("match_filter");
    const unsigned grid = cdiv(total_rows, 256);
    filter_mark_kernel<<<grid, 256, 0, stream()>>>(dfj, njobs, total_rows, idx, d1, d2, r2,
                                                    o.match_threshold, o.unique, keys, winner);
    check_launch("filter_mark_kernel");
    filter_select_kernel<<<grid, 256, 0, stream()>>>(dfj, njobs, total_rows, idx, o.unique, keys,
                                                      winner, packed, cnt);
    check_launch("filter_select_kernel");
    {  // job_ptr = exclusive scan of the per-job counts, with the total in slot njobs (cnt[njobs] is a zero pad)
        size_t sb = 0;
        APS_HIP(rocprim::exclusive_scan(nullptr, sb, cnt.get(), d_job_ptr, 0ull, (size_t)njobs + 1,
                                        rocprim::plus<unsigned long long>(), stream()));
        Ws<char> stmp(sb);
        APS_HIP(rocprim::exclusive_scan(stmp.get(), sb, cnt.get(), d_job_ptr, 0ull, (size_t)njobs + 1,
                                        rocprim::plus<unsigned long long>(), stream()));
    }

    // segmented ascending sort of each job's kept keys (the head of its row segment)
    Ws<int64_t> dseg(2 * (int64_t)njobs);
    filter_seg_end_kernel<<<cdiv(njobs, 256), 256, 0, stream()>>>(dfj, njobs, cnt, dseg.get(), dseg.get() + njobs);
    check_launch("filter_seg_end_kernel");
    size_t tmp_bytes = 0;
    APS_HIP(rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes, packed.get(), sorted.get(),
                                               (unsigned)total_rows, (unsigned)njobs, dseg.get(),
                                               dseg.get() + njobs, 0, 64, stream()));
    Ws<char> tmp(tmp_bytes);
    APS_HIP(rocprim::segmented_radix_sort_keys(tmp.get(), tmp_bytes, packed.get(), sorted.get(),
                                               (unsigned)total_rows, (unsigned)njobs, dseg.get(),
                                               dseg.get() + njobs, 0, 64, stream()));
    unsigned long long total = 0;
    APS_HIP(hipMemcpyAsync(&total, d_job_ptr + njobs, sizeof total, hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));  // also keeps fjobs alive long enough
    if ((int64_t)total > cap) return (int64_t)total;
    if (total > 0) {
        for (int job0 = 0; job0 < njobs; job0 += 65535) {  // gridDim.y is capped at 65535 (500 images = 124750 pairs)
            const dim3 g(64, std::min(65535, njobs - job0));
            if (o.unique)
                filter_emit_kernel<<<g, 256, 0, stream()>>>(dfj, njobs, sorted, d_job_ptr, idx, 1, o1, o2, metric, cap, job0);
            else
                filter_emit_rows_kernel<<<g, 256, 0, stream()>>>(dfj, sorted, d_job_ptr, idx, d1, o1, o2, metric, cap, job0);
        }
        check_launch("filter_emit_kernel");
    }
    APS_HIP(hipStreamSynchronize(stream()));
    return (int64_t)total;
}

static aps_match_opts default_opts() {
    aps_match_opts o;
    o.max_ratio = 0.6;
    o.match_threshold = 3.5;
    o.unique = 1;
    o.normalize = 2;
    return o;
}

}  // namespace aps

using namespace aps;

extern "C" {

int aps_match_2nn_ssd(const float* A, int64_t n1, int64_t lda, const float* B, int64_t n2,
                      int64_t ldb, int dim, int layout, uint32_t* idx2, float* d1, float* d2) {
    return guarded([&] {
        check_desc_args(A, n1, lda, dim, layout, "A");
        check_desc_args(B, n2, ldb, dim, layout, "B");
        APS_REQUIRE(n1 == 0 || (idx2 && d1 && d2), APS_E_ARG, "NULL output");
        ctx();
        if (n1 == 0) return;
        const size_t a_elems = layout == APS_ROWMAJOR ? (size_t)(n1 - 1) * lda + dim : (size_t)(dim - 1) * lda + n1;
        const size_t b_elems = n2 == 0 ? 0 : (layout == APS_ROWMAJOR ? (size_t)(n2 - 1) * ldb + dim : (size_t)(dim - 1) * ldb + n2);
        In<float> dA(A, a_elems), dB(B, b_elems);
        Out<uint32_t> oi(idx2, n1);
        Out<float> o1(d1, n1), o2(d2, n1);
        Prepared pa, pb;
        prepare(dA, n1, lda, layout, false, pa);
        prepare(dB, n2, ldb, layout, false, pb);
        if (n2 == 0) {
            // no candidates: idx 0, distances inf
            APS_HIP(hipMemsetAsync(oi.get(), 0, n1 * sizeof(uint32_t), stream()));
            std::vector<float> inf(n1, std::numeric_limits<float>::infinity());
            APS_HIP(hipMemcpyAsync(o1.get(), inf.data(), n1 * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemcpyAsync(o2.get(), inf.data(), n1 * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
        } else {
            std::vector<MatchJob> jobs(1);
            jobs[0] = make_job(pa, pb, (int)n1, (int)n2, 0);
            run_match_jobs(jobs, oi, o1, o2);
        }
        oi.commit();
        o1.commit();
        o2.commit();
    });
}

int aps_match_features(const float* F1, int64_t n1, int64_t ld1, const float* F2, int64_t n2,
                       int64_t ld2, int dim, int layout, const aps_match_opts* opts,
                       uint32_t* idx1, uint32_t* idx2, float* metric, int64_t cap, int64_t* count) {
    return guarded([&] {
        const float* descs[2] = {F1, F2};
        const int64_t counts[2] = {n1, n2};
        const int64_t lds[2] = {ld1, ld2};
        int64_t pair_ptr[2] = {0, 0};
        APS_REQUIRE(count != nullptr, APS_E_ARG, "count is NULL");
        const int rc = aps_match_pairwise(descs, counts, lds, 2, dim, layout, opts, pair_ptr, idx1,
                                          idx2, metric, cap, count);
        if (rc != APS_OK) throw Error(rc, aps_last_error());
    });
}

}  // extern "C"

static void match_pairs_impl(const float* const* desc, const int64_t* counts, const int64_t* ld, int n_img,
                             int dim, int layout, const std::vector<int32_t>& pa, const std::vector<int32_t>& pb,
                             const aps_match_opts* opts, int64_t* pair_ptr, uint32_t* idx_i, uint32_t* idx_j,
                             float* metric, int64_t cap, int64_t* count) {
    APS_REQUIRE(n_img >= 0, APS_E_ARG, "negative image count");
    APS_REQUIRE(count != nullptr && pair_ptr != nullptr, APS_E_ARG, "count/pair_ptr is NULL");
    APS_REQUIRE(n_img == 0 || (desc && counts && ld), APS_E_ARG, "NULL descriptor table");
    APS_REQUIRE(cap >= 0, APS_E_ARG, "negative capacity");
    const aps_match_opts o = opts ? *opts : default_opts();
    APS_REQUIRE(o.max_ratio > 0.0 && o.max_ratio <= 1.0, APS_E_ARG, "MaxRatio must be in (0,1]");
    APS_REQUIRE(o.match_threshold >= 0.0, APS_E_ARG, "MatchThreshold must be >= 0");
    APS_REQUIRE(o.normalize >= 0 && o.normalize <= 2, APS_E_ARG, "normalize must be 0, 1 or 2");
    for (int i = 0; i < n_img; ++i) check_desc_args(desc[i], counts[i], ld[i], dim, layout, "desc");
    const int64_t n_pairs = (int64_t)pa.size();
    for (int64_t p = 0; p < n_pairs; ++p)
        APS_REQUIRE(pa[p] >= 0 && pa[p] < n_img && pb[p] >= 0 && pb[p] < n_img && pa[p] != pb[p], APS_E_ARG,
                    "pair %lld = (%d,%d) is not a valid pair of distinct images", (long long)p, pa[p], pb[p]);
    ctx();
    *count = 0;
    if (n_pairs <= 0) {
        if (is_device_ptr(pair_ptr)) {
            const int64_t z = 0;
            APS_HIP(hipMemcpy(pair_ptr, &z, sizeof z, hipMemcpyHostToDevice));
        } else {
            pair_ptr[0] = 0;
        }
        return;
    }
    const bool trace = std::getenv("APS_TRACE") != nullptr;
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto T0 = t_now();
    // upload + probe (only images that take part in some pair)
    std::vector<char> used(n_img, 0);
    for (int64_t p = 0; p < n_pairs; ++p) used[pa[p]] = used[pb[p]] = 1;
    std::vector<In<float>> din(n_img);
    std::vector<float> amax(n_img, 0.f);
    Ws<float> slots((size_t)std::max(n_img, 1));
    APS_HIP(hipMemsetAsync(slots, 0, (size_t)std::max(n_img, 1) * sizeof(float), stream()));
    bool all_flat = layout == APS_ROWMAJOR;
    for (int i = 0; i < n_img; ++i) {
        if (!used[i]) continue;
        const int64_t n = counts[i];
        const size_t elems = n == 0 ? 0 : (layout == APS_ROWMAJOR ? (size_t)(n - 1) * ld[i] + dim : (size_t)(dim - 1) * ld[i] + n);
        din[i].bind(desc[i], elems);
        all_flat = all_flat && (n == 0 || (ld[i] == kDim && (reinterpret_cast<uintptr_t>((const float*)din[i]) & 15) == 0));
    }
    Ws<AbsmaxJob> d_probe;
    if (o.normalize == 2 && all_flat) {  // one launch for all sets
        std::vector<AbsmaxJob> pj(n_img, AbsmaxJob{nullptr, 0});
        int64_t n4max = 0;
        for (int i = 0; i < n_img; ++i)
            if (used[i] && counts[i] > 0) {
                pj[i] = AbsmaxJob{reinterpret_cast<const float4*>((const float*)din[i]), counts[i] * (kDim / 4)};
                n4max = std::max(n4max, pj[i].n4);
            }
        if (n4max > 0) {
            d_probe.alloc(n_img);
            APS_HIP(hipMemcpyAsync(d_probe, pj.data(), (size_t)n_img * sizeof(AbsmaxJob), hipMemcpyHostToDevice, stream()));
            absmax_batch_kernel<<<dim3(std::min<unsigned>(cdiv((size_t)n4max, 256), 64), n_img), 256, 0, stream()>>>(d_probe, slots);
            check_launch("absmax_batch_kernel");
        }
    } else if (o.normalize == 2) {
        for (int i = 0; i < n_img; ++i)
            if (used[i]) absmax_async(din[i], counts[i], ld[i], layout, (float*)slots + i);
    }
    if (o.normalize == 2) {  // one read-back for all images (a round trip per image cost ~50 us each)
        APS_HIP(hipMemcpyAsync(amax.data(), slots, (size_t)n_img * sizeof(float), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
    }
    // which variants (raw / normalised) of each image are needed: the reference decides per pair
    // (matchFeaturesScratch.m:105: max|A|>2 || max|B|>2)
    auto big = [&](int i) { return o.normalize == 1 || (o.normalize == 2 && amax[i] > 2.f); };
    std::vector<Prepared> raw(n_img), nrm(n_img);
    Arena prep_arena;  // (outlives the sets)
    Ws<float> prep_stats((size_t)2 * kStatWords * std::max(n_img, 1));
    std::vector<char> need_raw(n_img, 0), need_nrm(n_img, 0);
    for (int64_t p = 0; p < n_pairs; ++p) {
        const bool norm = big(pa[p]) || big(pb[p]);
        (norm ? need_nrm : need_raw)[pa[p]] = 1;
        (norm ? need_nrm : need_raw)[pb[p]] = 1;
    }
    {  // every set of the batch in four launches (prepare_batch); APS_MATCH_PREP_STREAMS=1 keeps the older form, one chain of
       // launches per set on eight forked streams
        constexpr int kPrepStreams = 8;
        APS_HIP(hipMemsetAsync(prep_stats, 0, (size_t)2 * kStatWords * std::max(n_img, 1) * sizeof(float), stream()));  // all sets' statistics
        if (!std::getenv("APS_MATCH_PREP_STREAMS")) {  // round 4: every set in two launches (see prep_desc_batch_kernel)
            std::vector<PrepRequest> req;
            for (int i = 0; i < n_img; ++i) {
                if (need_raw[i]) req.push_back({din[i], counts[i], ld[i], false, &raw[i], prep_stats.get() + 2 * kStatWords * i});
                if (need_nrm[i]) req.push_back({din[i], counts[i], ld[i], true, &nrm[i], prep_stats.get() + 2 * kStatWords * i + kStatWords});
            }
            prepare_batch(req, layout, &prep_arena);
        } else {
            AuxScope fork(kPrepStreams);
            std::vector<hipStream_t>& aux = fork.streams;
            Prof prof("match_prep");
            int k = 0;
            for (int i = 0; i < n_img; ++i) {
                if (need_raw[i]) prepare(din[i], counts[i], ld[i], layout, false, raw[i], aux[k++ % kPrepStreams], false, prep_stats.get() + 2 * kStatWords * i);
                if (need_nrm[i]) prepare(din[i], counts[i], ld[i], layout, true, nrm[i], aux[k++ % kPrepStreams], false, prep_stats.get() + 2 * kStatWords * i + kStatWords);
            }
            fork.join();
        }
    }
    const auto T1 = t_now();
    std::vector<MatchJob> jobs;
    std::vector<FilterJob> fjobs;
    int64_t rows = 0, cols = 0;
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int i = pa[p], j = pb[p];
        const bool norm = big(i) || big(j);
        const Prepared& a = norm ? nrm[i] : raw[i];
        const Prepared& b = norm ? nrm[j] : raw[j];
        // an empty side makes the pair empty (the reference's validateattributes would reject empty
        // float inputs; callers skip such images)
        const int nA = counts[j] == 0 ? 0 : (int)counts[i];
        jobs.push_back(make_job(a, b, nA, (int)counts[j], rows));
        fjobs.push_back({rows, cols, nA, (int)counts[j]});
        rows += nA;
        cols += counts[j];
    }
    APS_REQUIRE(rows < ((int64_t)1 << 31), APS_E_DIM, "too many pair-rows for one batch (%lld)", (long long)rows);
    Ws<uint32_t> idx(std::max<int64_t>(rows, 1));
    Ws<float> d1(std::max<int64_t>(rows, 1)), d2(std::max<int64_t>(rows, 1));
    const auto T2 = t_now();
    // the filter below keeps a row iff d1 <= r^2 d2 and d1 <= MatchThreshold (f64): the candidate kernel may dismiss rows
    // that provably fail it.  The constants are rounded so that the kernel's f32 tests err on the side of keeping a row.
    const bool prune = !std::getenv("APS_MATCH_NO_PRUNE");
    const float pr2 = prune ? std::nextafter((float)(o.max_ratio * o.max_ratio), INFINITY) : 0.f;
    const float pthr = std::nextafter((float)o.match_threshold, INFINITY);
    run_match_jobs(jobs, idx, d1, d2, pr2, pthr);
    const auto T3 = t_now();

    Out<uint32_t> oi(idx_i, cap), oj(idx_j, cap);
    Out<float> om(metric, cap);
    Ws<unsigned long long> job_ptr(n_pairs + 1);
    const int64_t total = run_filter(fjobs, rows, cols, idx, d1, d2, o, job_ptr, oi, oj, om, cap);
    *count = total;
    std::vector<unsigned long long> hp(n_pairs + 1);
    APS_HIP(hipMemcpyAsync(hp.data(), job_ptr, (n_pairs + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    if (is_device_ptr(pair_ptr)) {
        std::vector<int64_t> tmp(hp.begin(), hp.end());
        APS_HIP(hipMemcpy(pair_ptr, tmp.data(), tmp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    } else {
        for (int64_t p = 0; p <= n_pairs; ++p) pair_ptr[p] = (int64_t)hp[p];
    }
    if (total > cap) fail(APS_E_CAP, "output capacity %lld < %lld matches", (long long)cap, (long long)total);
    APS_REQUIRE(total == 0 || (idx_i && idx_j && metric), APS_E_ARG, "NULL output with matches present");
    oi.commit(total);
    oj.commit(total);
    om.commit(total);
    if (trace)
        std::fprintf(stderr, "[aps] match_pairs: probe+prepare %.2f ms, jobs %.2f, 2-NN %.2f, filter+out %.2f\n", t_ms(T0, T1),
                     t_ms(T1, T2), t_ms(T2, T3), t_ms(T3, t_now()));
}

extern "C" {

int aps_match_screen_stats(int64_t* rows, int64_t* survivors) {
    if (rows) *rows = g_screen_rows;
    if (survivors) *survivors = g_screen_surv;
    return APS_OK;
}

int aps_match_screen_exact_jobs(int64_t* jobs, int64_t* exact_jobs) {
    if (jobs) *jobs = g_screen_jobs;
    if (exact_jobs) *exact_jobs = g_screen_exact;
    return APS_OK;
}

int aps_match_pairwise(const float* const* desc, const int64_t* counts, const int64_t* ld,
                       int n_img, int dim, int layout, const aps_match_opts* opts,
                       int64_t* pair_ptr, uint32_t* idx_i, uint32_t* idx_j, float* metric,
                       int64_t cap, int64_t* count) {
    return guarded([&] {
        APS_REQUIRE(n_img >= 0, APS_E_ARG, "negative image count");
        // the reference's pair order (featureMatchingPairwise.m:48: column-major triu)
        std::vector<int32_t> pa, pb;
        for (int j = 1; j < n_img; ++j)
            for (int i = 0; i < j; ++i) {
                pa.push_back(i);
                pb.push_back(j);
            }
        match_pairs_impl(desc, counts, ld, n_img, dim, layout, pa, pb, opts, pair_ptr, idx_i, idx_j, metric, cap, count);
    });
}

int aps_match_pairs(const float* const* desc, const int64_t* counts, const int64_t* ld, int n_img,
                    int dim, int layout, const int32_t* pair_a, const int32_t* pair_b, int64_t n_pairs,
                    const aps_match_opts* opts, int64_t* pair_ptr, uint32_t* idx_a, uint32_t* idx_b,
                    float* metric, int64_t cap, int64_t* count) {
    return guarded([&] {
        APS_REQUIRE(n_pairs >= 0, APS_E_ARG, "negative pair count");
        APS_REQUIRE(n_pairs == 0 || (pair_a && pair_b), APS_E_ARG, "NULL pair list");
        std::vector<int32_t> pa(pair_a, pair_a + n_pairs), pb(pair_b, pair_b + n_pairs);
        match_pairs_impl(desc, counts, ld, n_img, dim, layout, pa, pb, opts, pair_ptr, idx_a, idx_b, metric, cap, count);
    });
}

}  // extern "C"

extern "C" int aps_match_set_stats(const float* X, int64_t n, int64_t ld, int layout, int normalize, float* stats) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(X != nullptr && stats != nullptr, APS_E_ARG, "NULL argument");
        APS_REQUIRE(n >= 1, APS_E_DIM, "need at least one row");
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_ARG, "unknown layout");
        APS_REQUIRE(layout == APS_ROWMAJOR ? ld >= kDim : ld >= n, APS_E_DIM, "leading dimension too small");
        ctx();
        In<float> dx;
        dx.bind(X, layout == APS_ROWMAJOR ? (size_t)(n - 1) * ld + kDim : (size_t)(kDim - 1) * ld + n);
        Prepared p;
        prepare(dx, n, ld, layout, normalize != 0, p);
        float raw[8];
        APS_HIP(hipMemcpyAsync(raw, p.stat, sizeof raw, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        // the int8 words are stored as order-preserving integers (complemented where a minimum is kept as a maximum)
        auto bits = [&](int k) { unsigned u; std::memcpy(&u, &raw[k], 4); return u; };
        auto unord = [](unsigned k) { const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
        auto as_f = [](unsigned u) { float f; std::memcpy(&f, &u, 4); return f; };
        stats[0] = raw[0];
        stats[1] = raw[1];
        stats[2] = raw[2];
        stats[3] = raw[3];
        stats[4] = unord(bits(4));
        stats[5] = raw[5];
        stats[6] = as_f(~bits(6));
        stats[7] = unord(~bits(7));
    });
}

extern "C" int aps_match_screen_kernel_regs(int shape, int bounds_pass, int* num_regs, int* max_threads_per_block) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(num_regs && max_threads_per_block && (shape == 16 || shape == 32), APS_E_ARG, "shape is 16 or 32; the outputs must not be null");
        ctx();
        screen_regs(shape == 32, bounds_pass != 0, num_regs, max_threads_per_block);
    });
}
