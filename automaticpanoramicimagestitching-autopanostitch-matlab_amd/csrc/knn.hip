// knn.hip — the global (AutoStitch-style) matcher's device side and the binary-descriptor 2-NN.
//
//   aps_knn_global    : [idx, dist] = flann_knn_win(train, query, k, 'flann', trees, checks) for float
//                       descriptors (PP/mex/flann_knn.cpp:118-253, caller PP/featureMatching/
//                       featureMatchingGlobal.m:106-120) with an EXACT search in place of OpenCV's randomized
//                       kd-forest: squared L2 in the canonical f32 arithmetic of match.hip, ascending,
//                       ties -> lower index.
//   aps_global_filter : the per-query loop of featureMatchingGlobal.m:123-161.
//   aps_hamming_2nn   : PP/mex/nearest2HammingExhaustiveMEX.cpp:16-80 (and the OMP twin): brute-force Hamming
//                       2-NN with the reference's tie rule.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "aps_internal.h"

#include <rocprim/rocprim.hpp>

namespace aps {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kDim = 128;
constexpr int kTN = 64;
constexpr int kLdsRow = 132;

// ---- prep: canonical ||x||^2 and the k-permuted copy (same as match.hip's, without normalisation) --------
__global__ void knn_prep_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout,
                                float* __restrict__ P, float* __restrict__ sq) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[kDim];
#pragma unroll
    for (int k = 0; k < kDim; ++k) x[k] = layout == APS_ROWMAJOR ? X[i * ld + k] : X[i + k * ld];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kDim; ++k) s = __fadd_rn(s, __fmul_rn(x[k], x[k]));
    sq[i] = s;
    float* p = P + i * kDim;
#pragma unroll
    for (int k = 0; k < kDim; ++k) p[(k & 1) * 64 + (k >> 1)] = x[k];
}

// ---- exact kNN on v_mfma_f32_32x32x2_f32 ----------------------------------------------------------------
// Operands are swapped as in match_cand_bf16_kernel: the streamed train tile is the MFMA "A" operand, the
// resident query rows are the "B" operand, so a lane owns ONE query row (col = lane & 31) and sees 16 train
// columns per block; its running top-K is 2K registers.  Distances are exact canonical f32, so no rescoring.
template <int K>
__device__ __forceinline__ void topk_insert(float d, int j, float (&v)[K], int (&id)[K]) {
    // ascending (d, j) list; (d, j) goes in front of every entry it precedes lexicographically
    bool lt[K];
#pragma unroll
    for (int e = 0; e < K; ++e) lt[e] = d < v[e] || (d == v[e] && j < id[e]);
#pragma unroll
    for (int e = K - 1; e >= 1; --e) {
        v[e] = lt[e - 1] ? v[e - 1] : (lt[e] ? d : v[e]);
        id[e] = lt[e - 1] ? id[e - 1] : (lt[e] ? j : id[e]);
    }
    v[0] = lt[0] ? d : v[0];
    id[0] = lt[0] ? j : id[0];
}

template <int K>
__global__ __launch_bounds__(256, 2) void knn_f32_kernel(const float* __restrict__ PQ, const float* __restrict__ sqQ,
                                                         int nq, const float* __restrict__ PT,
                                                         const float* __restrict__ sqT, int nt, int k_out,
                                                         uint32_t* __restrict__ idx, float* __restrict__ dist,
                                                         int64_t ldo, int layout, int block_rows) {
    __shared__ __attribute__((aligned(16))) float lds[2 * kTN * kLdsRow];
    __shared__ float s_t2[2][kTN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 31, h = lane >> 5;
    const int row = blockIdx.x * 128 + wave * 32 + c;
    if (block_rows > 0) {
        // blocked form: the workgroup's 128 query rows are searched in their OWN block of the pool only (block_rows is a
        // multiple of 128), indices come out relative to that block
        const int base = (int)((blockIdx.x * 128) / block_rows) * block_rows;
        PT += (size_t)base * kDim;
        sqT += base;
        nt = min(block_rows, nt - base);
    }
    const int qrow = min(row, nq - 1);
    // resident operand (MFMA B): this lane's query row, k = 2s + h
    f32x4 qv[16];
    {
        const f32x4* qp = reinterpret_cast<const f32x4*>(PQ + (size_t)qrow * kDim + h * 64);
#pragma unroll
        for (int q = 0; q < 16; ++q) qv[q] = qp[q];
    }
    const float q2 = sqQ[qrow];
    float v[K];
    int id[K];
#pragma unroll
    for (int e = 0; e < K; ++e) {
        v[e] = INFINITY;
        id[e] = 0x7fffffff;
    }
    const int ntiles = (nt + kTN - 1) / kTN;
    f32x4 stage[8];
    float stage_t2 = 0.f;
    auto load_tile = [&](int t) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;
            const int trow = min(t * kTN + (f >> 5), nt - 1);
            stage[u] = *reinterpret_cast<const f32x4*>(PT + (size_t)trow * kDim + (f & 31) * 4);
        }
        if (tid < kTN) stage_t2 = (t * kTN + tid) < nt ? sqT[t * kTN + tid] : INFINITY;
    };
    auto store_tile = [&](int buf) {
        float* base = lds + buf * (kTN * kLdsRow);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int f = tid + 256 * u;
            *reinterpret_cast<f32x4*>(base + (f >> 5) * kLdsRow + (f & 31) * 4) = stage[u];
        }
        if (tid < kTN) s_t2[buf][tid] = stage_t2;
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) load_tile(t + 1);
        const float* tile = lds + (t & 1) * (kTN * kLdsRow);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f32x4* tp = reinterpret_cast<const f32x4*>(tile + (cb * 32 + c) * kLdsRow + h * 64);
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const f32x4 tv = tp[q];
                // G(train j, query i) = sum_k q_k * t_k in k-ascending fma order: the MFMA's A operand is the
                // train value, B the query value; the product is commutative, the chain order is k
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tv.x, qv[q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tv.y, qv[q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tv.z, qv[q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tv.w, qv[q].w, acc, 0, 0, 0);
            }
            const float* t2p = &s_t2[t & 1][cb * 32 + 4 * h];
            const int jbase = t * kTN + cb * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jj = (r & 3) + 8 * (r >> 2);
                // same arithmetic as the pairwise matcher: (q2 + t2) - 2*G
                const float d = __fsub_rn(__fadd_rn(q2, t2p[jj]), __fmul_rn(2.0f, acc[r]));
                if (__any(d < v[K - 1])) topk_insert<K>(d, jbase + jj, v, id);
            }
        }
        if (t + 1 < ntiles) store_tile((t + 1) & 1);
        __syncthreads();
    }
    // merge the two half-waves (disjoint column sets of the same row)
    float pv[K];
    int pid[K];
#pragma unroll
    for (int e = 0; e < K; ++e) {
        pv[e] = __shfl_xor(v[e], 32);
        pid[e] = __shfl_xor(id[e], 32);
    }
#pragma unroll
    for (int e = 0; e < K; ++e) topk_insert<K>(pv[e], pid[e], v, id);
    if (h == 0 && row < nq) {
#pragma unroll
        for (int e = 0; e < K; ++e) {
            if (e >= k_out) break;
            const bool ok = id[e] != 0x7fffffff && id[e] < nt;
            const int64_t o = layout == APS_ROWMAJOR ? (int64_t)row * ldo + e : (int64_t)e * ldo + row;
            idx[o] = ok ? (uint32_t)id[e] + 1u : 0u;
            dist[o] = ok ? v[e] : INFINITY;
        }
    }
}

// ---- blocked global k-NN of a pool against itself: merge of the per-block lists --------------------------------------
// Row q of block i holds: its exact top-4 inside its own block (self included) and, for every other block j, the three
// nearest rows of j with exact distances plus a bound below which no further row of j lies (match.hip,
// screened_block_top3).  The global top-4 is the (distance, index)-ascending head of their union, and it is CERTIFIED when
// its fourth distance is strictly below every block's bound (no unlisted row can enter or tie).  Uncertified rows - a
// query with three or more near-identical rows in ANOTHER block - are recomputed against the whole pool.
__global__ void knn_merge_kernel(const int64_t* __restrict__ block_off, int nb, const int64_t* __restrict__ job_off,
                                 const uint32_t* __restrict__ self_idx, const float* __restrict__ self_d,
                                 const uint32_t* __restrict__ t3_idx, const float* __restrict__ t3_d,
                                 const float* __restrict__ t3_b, int64_t f, int k_out, uint32_t* __restrict__ idx,
                                 float* __restrict__ dist, int64_t ldo, int layout, uint32_t* __restrict__ unc_list,
                                 unsigned int* __restrict__ unc_count) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= f) return;
    int lo = 0, hi = nb - 1;  // block of q
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_off[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int bi = lo;
    const int64_t r = q - block_off[bi];
    float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    long long id[4] = {-1, -1, -1, -1};
    auto put = [&](float d, long long g) {
        if (g < 0) return;
        bool lt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lt[e] = id[e] < 0 || d < v[e] || (d == v[e] && g < id[e]);
        if (!lt[3]) return;
#pragma unroll
        for (int e = 3; e >= 1; --e) {
            v[e] = lt[e - 1] ? v[e - 1] : d;
            id[e] = lt[e - 1] ? id[e - 1] : g;
            if (!lt[e - 1]) return;
        }
        v[0] = d;
        id[0] = g;
    };
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t li = self_idx[q * 4 + e];
        if (li) put(self_d[q * 4 + e], block_off[bi] + (long long)li - 1);
    }
    float min_bound = INFINITY;
    // the own block's list is exact and complete up to its fourth entry: anything unlisted there is at least that far
    if (self_idx[q * 4 + 3]) min_bound = self_d[q * 4 + 3];
    for (int j = 0; j < nb; ++j) {
        if (j == bi) continue;
        const int64_t slot = job_off[(size_t)bi * nb + j] + r;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const uint32_t li = t3_idx[slot * 3 + e];
            if (li) put(t3_d[slot * 3 + e], block_off[j] + (long long)li - 1);
        }
        min_bound = fminf(min_bound, t3_b[slot]);
    }
    // certified: the k-th distance reported is strictly below what any unlisted row can have.  (The own block's fourth
    // entry may itself be the global fourth: then nothing unlisted of that block matters only if it is strictly farther -
    // ties are resolved by index, which the exact own-block list already did, so that bound is taken non-strictly.)
    const int kk = k_out < 4 ? k_out : 4;
    const float dk = id[kk - 1] >= 0 ? v[kk - 1] : INFINITY;
    float other_bound = INFINITY;
    for (int j = 0; j < nb; ++j)
        if (j != bi) other_bound = fminf(other_bound, t3_b[job_off[(size_t)bi * nb + j] + r]);
    const bool certified = dk < other_bound || !(other_bound < INFINITY);
    (void)min_bound;
    if (certified) {
        for (int e = 0; e < kk; ++e) {
            const int64_t o = layout == APS_ROWMAJOR ? q * ldo + e : (int64_t)e * ldo + q;
            idx[o] = id[e] >= 0 ? (uint32_t)(id[e] + 1) : 0u;
            dist[o] = id[e] >= 0 ? v[e] : INFINITY;
        }
    } else {
        unc_list[atomicAdd(unc_count, 1u)] = (uint32_t)q;
    }
}

// The same merge for the pooled matcher's screened search (screened_global_top3): blocks = images, EVERY image
// (the row's own too) comes as a certified prefix of three with a bound; rows the filter provably drops are written as
// four copies of themselves (featureMatchingGlobal.m:129-141 removes the self matches, fewer than two candidates
// remain, the query is skipped) and are never looked up.
__global__ void knn_merge3_kernel(const int64_t* __restrict__ block_off, int nb, const int64_t* __restrict__ job_off,
                                  const uint8_t* __restrict__ dismissed, const uint32_t* __restrict__ t3_idx,
                                  const float* __restrict__ t3_d, const float* __restrict__ t3_b, int64_t f, int k_out,
                                  uint32_t* __restrict__ idx, float* __restrict__ dist, int64_t ldo, int layout,
                                  uint32_t* __restrict__ unc_list, unsigned int* __restrict__ unc_count, int64_t q_lo) {
    const int64_t q = q_lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;  // this call's rows: [q_lo, f)
    if (q >= f) return;
    const int kk = k_out < 4 ? k_out : 4;
    if (dismissed[q]) {
        for (int e = 0; e < kk; ++e) {
            const int64_t o = layout == APS_ROWMAJOR ? q * ldo + e : (int64_t)e * ldo + q;
            idx[o] = (uint32_t)(q + 1);
            dist[o] = 0.f;
        }
        return;
    }
    int lo = 0, hi = nb - 1;  // image of q
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_off[mid] <= q) lo = mid; else hi = mid - 1;
    }
    const int bi = lo;
    const int64_t r = q - block_off[bi];
    float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    long long id[4] = {-1, -1, -1, -1};
    auto put = [&](float d, long long g) {
        bool lt[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lt[e] = id[e] < 0 || d < v[e] || (d == v[e] && g < id[e]);
        if (!lt[3]) return;
#pragma unroll
        for (int e = 3; e >= 1; --e) {
            v[e] = lt[e - 1] ? v[e - 1] : d;
            id[e] = lt[e - 1] ? id[e - 1] : g;
            if (!lt[e - 1]) return;
        }
        v[0] = d;
        id[0] = g;
    };
    float bound = INFINITY;
    for (int j = 0; j < nb; ++j) {
        if (block_off[j + 1] == block_off[j]) continue;
        const int64_t slot = job_off[(size_t)bi * nb + j] + r;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            const uint32_t li = t3_idx[slot * 3 + e];
            if (li) put(t3_d[slot * 3 + e], block_off[j] + (long long)li - 1);
        }
        bound = fminf(bound, t3_b[slot]);
    }
    const float dk = id[kk - 1] >= 0 ? v[kk - 1] : INFINITY;
    const bool certified = dk < bound || !(bound < INFINITY);
    if (certified) {
        for (int e = 0; e < kk; ++e) {
            const int64_t o = layout == APS_ROWMAJOR ? q * ldo + e : (int64_t)e * ldo + q;
            idx[o] = id[e] >= 0 ? (uint32_t)(id[e] + 1) : 0u;
            dist[o] = id[e] >= 0 ? v[e] : INFINITY;
        }
    } else {
        unc_list[atomicAdd(unc_count, 1u)] = (uint32_t)q;
    }
}

// Exact top-4 of one uncertified query row inside ONE block of the pool (one wave per (row, block); lane j takes the
// block's rows j, j+64, ...).  d = (q2 + t2) - 2 G with G the k-ascending f32 fma chain: the arithmetic of knn_f32_kernel
// (the MFMA computes that chain) on the k-permuted copies.  Each lane keeps its four nearest - a row among the block's
// four nearest has at most three rows ahead of it anywhere, so it is in its lane's list - and lane 0 merges them.
__global__ __launch_bounds__(64) void knn_rows_block_top4_kernel(const float* __restrict__ P, const float* __restrict__ sq,
                                                                 const uint32_t* __restrict__ rows, int n_rows,
                                                                 const int64_t* __restrict__ block_off, int nb,
                                                                 long long* __restrict__ out_i, float* __restrict__ out_d) {
    __shared__ float s_d[64 * 4];
    __shared__ int s_i[64 * 4];
    const int ri = blockIdx.x / nb, b = blockIdx.x % nb;
    if (ri >= n_rows) return;
    const int64_t q = rows[ri];
    const int64_t t0 = block_off[b], nt = block_off[b + 1] - t0;
    const float* pa = P + (size_t)q * kDim;
    const float q2 = sq[q];
    float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int id[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int j = threadIdx.x; j < nt; j += 64) {
        const float* pb = P + (size_t)(t0 + j) * kDim;
        float g = 0.f;
#pragma unroll 4
        for (int s4 = 0; s4 < 16; ++s4) {
            const f32x4 ae = *reinterpret_cast<const f32x4*>(pa + 4 * s4), ao = *reinterpret_cast<const f32x4*>(pa + 64 + 4 * s4);
            const f32x4 be = *reinterpret_cast<const f32x4*>(pb + 4 * s4), bo = *reinterpret_cast<const f32x4*>(pb + 64 + 4 * s4);
            g = fmaf(ae.x, be.x, g);
            g = fmaf(ao.x, bo.x, g);
            g = fmaf(ae.y, be.y, g);
            g = fmaf(ao.y, bo.y, g);
            g = fmaf(ae.z, be.z, g);
            g = fmaf(ao.z, bo.z, g);
            g = fmaf(ae.w, be.w, g);
            g = fmaf(ao.w, bo.w, g);
        }
        const float dj = __fsub_rn(__fadd_rn(q2, sq[t0 + j]), __fmul_rn(2.0f, g));
        if (dj < v[3]) {  // ascending j per lane: a tie stays behind the earlier entry
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
        for (int e2 = 0; e2 < 64 * 4; ++e2) {
            const float dq = s_d[e2];
            const int iq = s_i[e2];
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
        for (int e = 0; e < 4; ++e) {
            out_i[((size_t)ri * nb + b) * 4 + e] = bi[e] < nt ? t0 + bi[e] : -1;
            out_d[((size_t)ri * nb + b) * 4 + e] = bd[e];
        }
    }
}

// global top-k of an uncertified row from its per-block exact top-4 lists
__global__ void knn_rows_merge_kernel(const uint32_t* __restrict__ rows, int n_rows, int nb, const long long* __restrict__ li,
                                      const float* __restrict__ ld_, int k_out, uint32_t* __restrict__ idx,
                                      float* __restrict__ dist, int64_t ldo, int layout) {
    const int ri = blockIdx.x * blockDim.x + threadIdx.x;
    if (ri >= n_rows) return;
    float v[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    long long id[4] = {-1, -1, -1, -1};
    for (int e2 = 0; e2 < nb * 4; ++e2) {
        const long long g = li[(size_t)ri * nb * 4 + e2];
        const float d = ld_[(size_t)ri * nb * 4 + e2];
        if (g < 0) continue;
        int pos = 4;
        while (pos > 0 && (id[pos - 1] < 0 || d < v[pos - 1] || (d == v[pos - 1] && g < id[pos - 1]))) --pos;
        if (pos >= 4) continue;
        for (int e = 3; e > pos; --e) {
            v[e] = v[e - 1];
            id[e] = id[e - 1];
        }
        v[pos] = d;
        id[pos] = g;
    }
    const int64_t q = rows[ri];
    for (int e = 0; e < k_out && e < 4; ++e) {
        const int64_t o = layout == APS_ROWMAJOR ? q * ldo + e : (int64_t)e * ldo + q;
        idx[o] = id[e] >= 0 ? (uint32_t)(id[e] + 1) : 0u;
        dist[o] = id[e] >= 0 ? v[e] : INFINITY;
    }
}

// allDesc ./ sqrt(sum(allDesc.^2, 2) + eps('single'))  (featureMatchingGlobal.m:83-85: eps INSIDE the root); the sum is
// the k-ascending chain of separate multiplies and adds, like every other canonical sum of this library
__global__ void global_normalize_kernel(const float* __restrict__ X, int64_t n, int64_t ld, int layout, float* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x[kDim];
#pragma unroll
    for (int k = 0; k < kDim; ++k) x[k] = layout == APS_ROWMAJOR ? X[i * ld + k] : X[i + k * ld];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kDim; ++k) s = __fadd_rn(s, __fmul_rn(x[k], x[k]));
    const float nrm = sqrtf(__fadd_rn(s, 1.1920928955078125e-07f));
#pragma unroll
    for (int k = 0; k < kDim; ++k) out[i * kDim + k] = __fdiv_rn(x[k], nrm);
}

// ---- featureMatchingGlobal.m:123-161 ------------------------------------------------------------------------
__global__ void global_filter_kernel(const uint32_t* __restrict__ nn_idx, const float* __restrict__ nn_dist,
                                     int64_t f, int k, int64_t ldn, int layout,
                                     const uint32_t* __restrict__ img_idx, const uint32_t* __restrict__ local_idx,
                                     float ratio, unsigned long long* __restrict__ keys /* pair<<40 | q */,
                                     uint32_t* __restrict__ li, uint32_t* __restrict__ lj) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= f) return;
    const uint32_t qi = img_idx[q];
    uint32_t n0 = 0;
    float d0 = 0, d1 = 0;
    int c = 0;
    for (int u = 0; u < k && c < 2; ++u) {
        const int64_t o = layout == APS_ROWMAJOR ? q * ldn + u : (int64_t)u * ldn + q;
        const uint32_t id = nn_idx[o];
        if (id == 0 || id > (uint32_t)f) continue;   // missing neighbour
        if (id == (uint32_t)(q + 1)) continue;       // :130 self
        if (img_idx[id - 1] == qi) continue;         // :135 same image
        if (c == 0) {
            n0 = id;
            d0 = nn_dist[o];
        } else {
            d1 = nn_dist[o];
        }
        ++c;
    }
    unsigned long long key = ~0ull;
    if (c >= 2) {                                    // :140
        const float eps = 1.1920928955078125e-07f;
        const float r = d0 / (d1 > eps ? d1 : eps);  // :145 in single
        if (!(r > ratio)) {
            const uint32_t j = img_idx[n0 - 1];
            const uint32_t a = qi < j ? qi : j, b = qi < j ? j : qi;  // 1-based image ids, a < b
            const unsigned long long pair = (unsigned long long)(b - 1) * (b - 2) / 2 + (a - 1);
            key = (pair << 40) | (unsigned long long)q;
            li[q] = qi < j ? local_idx[q] : local_idx[n0 - 1];
            lj[q] = qi < j ? local_idx[n0 - 1] : local_idx[q];
        }
    }
    keys[q] = key;
}

__global__ void global_emit_kernel(const unsigned long long* __restrict__ sorted, int64_t total,
                                   const uint32_t* __restrict__ li, const uint32_t* __restrict__ lj,
                                   uint32_t* __restrict__ oi, uint32_t* __restrict__ oj,
                                   unsigned long long* __restrict__ pair_count) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= total) return;
    const unsigned long long key = sorted[e];
    const int64_t q = (int64_t)(key & ((1ull << 40) - 1));
    oi[e] = li[q];
    oj[e] = lj[q];
    atomicAdd(&pair_count[key >> 40], 1ull);
}

// ---- Hamming 2-NN ----------------------------------------------------------------------------------------------
// one thread per A row (its bytes in registers), B streamed through LDS in tiles; the scan order over j and the
// strict-< / <= update rule are those of nearest2HammingExhaustiveMEX.cpp:52-69.
template <int NW>  // words of 4 bytes per descriptor
__global__ void hamming_kernel(const uint32_t* __restrict__ A, int64_t n1, const uint32_t* __restrict__ B, int64_t n2,
                               int nbytes, uint32_t* __restrict__ idx2, float* __restrict__ d1, float* __restrict__ d2) {
    __shared__ uint32_t s_b[256 * NW];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    uint32_t a[NW];
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) a[wv] = i < n1 ? A[i * NW + wv] : 0u;
    unsigned best = 0xFFFFu, second = 0xFFFFu;
    int64_t ibest = -1, isecond = -1;
    for (int64_t j0 = 0; j0 < n2; j0 += 256) {
        const int cnt = (int)(n2 - j0 < 256 ? n2 - j0 : 256);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * NW; e += blockDim.x) s_b[e] = B[j0 * NW + e];
        __syncthreads();
        for (int jj = 0; jj < cnt; ++jj) {
            unsigned hsum = 0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) hsum += __popc(a[wv] ^ s_b[jj * NW + wv]);
            const int64_t j = j0 + jj;
            if (hsum < best) {
                second = best;
                isecond = ibest;
                best = hsum;
                ibest = j;
            } else if (hsum <= second && j != ibest) {
                second = hsum;
                isecond = j;
            }
        }
    }
    if (i >= n1) return;
    if (n2 == 1 || isecond == -1) second = (unsigned)(nbytes * 8);  // :71-74
    idx2[i] = (uint32_t)(ibest + 1);
    d1[i] = (float)best;
    d2[i] = (float)second;
}

// ---- Hamming k-NN: the uint8 branches of flann_knn.cpp (:199-223 BFMatcher knnMatch, :235-240 LSH index) ----------
// one thread per query row (its bytes in registers), train rows through LDS; ascending (distance, index) list of K.
// A candidate enters in front of every entry with a LARGER distance: among equal distances the lower index stays first.
template <int NW, int K>
__global__ void hamming_knn_kernel(const uint32_t* __restrict__ Q, int64_t nq, const uint32_t* __restrict__ T, int64_t nt,
                                   int k_out, uint32_t* __restrict__ idx, float* __restrict__ dist, int64_t ldo, int layout) {
    __shared__ uint32_t s_t[256 * NW];
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    uint32_t a[NW];
#pragma unroll
    for (int wv = 0; wv < NW; ++wv) a[wv] = i < nq ? Q[i * NW + wv] : 0u;
    unsigned v[K];
    int id[K];
#pragma unroll
    for (int e = 0; e < K; ++e) {
        v[e] = 0xFFFFFFFFu;
        id[e] = -1;
    }
    for (int64_t j0 = 0; j0 < nt; j0 += 256) {
        const int cnt = (int)(nt - j0 < 256 ? nt - j0 : 256);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * NW; e += blockDim.x) s_t[e] = T[j0 * NW + e];
        __syncthreads();
        for (int jj = 0; jj < cnt; ++jj) {
            unsigned hsum = 0;
#pragma unroll
            for (int wv = 0; wv < NW; ++wv) hsum += __popc(a[wv] ^ s_t[jj * NW + wv]);
            if (hsum < v[K - 1]) {
                const int j = (int)(j0 + jj);
                bool lt[K];
#pragma unroll
                for (int e = 0; e < K; ++e) lt[e] = hsum < v[e];
#pragma unroll
                for (int e = K - 1; e >= 1; --e) {
                    v[e] = lt[e - 1] ? v[e - 1] : (lt[e] ? hsum : v[e]);
                    id[e] = lt[e - 1] ? id[e - 1] : (lt[e] ? j : id[e]);
                }
                v[0] = lt[0] ? hsum : v[0];
                id[0] = lt[0] ? j : id[0];
            }
        }
    }
    if (i >= nq) return;
#pragma unroll
    for (int e = 0; e < K; ++e) {
        if (e >= k_out) break;
        const size_t o = layout == APS_ROWMAJOR ? (size_t)i * ldo + e : (size_t)i + (size_t)e * ldo;
        idx[o] = id[e] >= 0 ? (uint32_t)id[e] + 1u : 0u;           // missing neighbour: index 0 (flann_knn.cpp:217)
        dist[o] = id[e] >= 0 ? (float)v[e] : INFINITY;              // ... and Inf (:218)
    }
}

__global__ void pack_bytes_kernel(const uint8_t* __restrict__ X, int64_t n, int64_t ld, int nbytes, int layout, int nw,
                                  uint32_t* __restrict__ out) {
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= n * nw) return;
    const int64_t i = e / nw;
    const int wv = (int)(e % nw);
    uint32_t v = 0;
    for (int b = 0; b < 4; ++b) {
        const int k = 4 * wv + b;
        if (k < nbytes) v |= (uint32_t)(layout == APS_ROWMAJOR ? X[i * ld + k] : X[i + (int64_t)k * ld]) << (8 * b);
    }
    out[e] = v;
}

}  // namespace aps

using namespace aps;

static thread_local int64_t g_global_rows = 0, g_global_surv = 0;  // aps_knn_global_screen_stats

extern "C" {

int aps_knn_global(const float* train, int64_t ft, int64_t ldt, const float* query, int64_t fq, int64_t ldq,
                   int dim, int layout, int k, uint32_t* idx, float* dist, int64_t ldo) {
    return guarded([&] {
        APS_REQUIRE(dim == kDim, APS_E_DIM, "descriptor length %d not supported (built for %d-D SIFT)", dim, kDim);
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(k > 0, APS_E_ARG, "k must be > 0");                       // flann_knn:k
        APS_REQUIRE(k <= 8, APS_E_ARG, "k <= 8 supported (featureMatchingGlobal uses k = 4)");
        APS_REQUIRE(ft >= 0 && fq >= 0 && ft < (1ll << 31) && fq < (1ll << 31), APS_E_ARG, "bad sizes");
        APS_REQUIRE(fq == 0 || (idx && dist && query), APS_E_ARG, "NULL argument");
        APS_REQUIRE(ft == 0 || train, APS_E_ARG, "NULL train");
        if (layout == APS_ROWMAJOR)
            APS_REQUIRE(ldt >= dim && ldq >= dim && ldo >= k, APS_E_DIM, "leading dimension too small");
        else
            APS_REQUIRE(ldt >= ft && ldq >= fq && ldo >= fq, APS_E_DIM, "leading dimension too small");
        ctx();
        if (fq == 0) return;
        const size_t te = ft == 0 ? 0 : (layout == APS_ROWMAJOR ? (size_t)(ft - 1) * ldt + dim : (size_t)(dim - 1) * ldt + ft);
        const size_t qe = layout == APS_ROWMAJOR ? (size_t)(fq - 1) * ldq + dim : (size_t)(dim - 1) * ldq + fq;
        const size_t oe = layout == APS_ROWMAJOR ? (size_t)(fq - 1) * ldo + k : (size_t)(k - 1) * ldo + fq;
        const bool same = train == query && ft == fq && ldt == ldq;
        In<float> dT(train, te), dQ;
        if (!same) dQ.bind(query, qe);
        Out<uint32_t> oi(idx, oe);
        Out<float> od(dist, oe);
        Ws<float> PT((size_t)std::max<int64_t>(ft, 1) * kDim), sT(std::max<int64_t>(ft, 1)), PQ, sQ;
        if (ft > 0) knn_prep_kernel<<<cdiv(ft, 64), 64, 0, stream()>>>(dT, ft, ldt, layout, PT, sT);
        const float *pq = PT, *sq = sT;
        if (!same) {
            PQ.alloc((size_t)fq * kDim);
            sQ.alloc(fq);
            knn_prep_kernel<<<cdiv(fq, 64), 64, 0, stream()>>>(dQ, fq, ldq, layout, PQ, sQ);
            pq = PQ;
            sq = sQ;
        }
        check_launch("knn_prep_kernel");
        if (ft == 0) {
            std::vector<uint32_t> z(oe, 0u);
            std::vector<float> inf(oe, INFINITY);
            APS_HIP(hipMemcpyAsync(oi.get(), z.data(), oe * sizeof(uint32_t), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemcpyAsync(od.get(), inf.data(), oe * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
        } else if (same && k <= 4 && ft >= 8192 && !(std::getenv("APS_KNN_MODE") && !std::strcmp(std::getenv("APS_KNN_MODE"), "f32"))) {
            // Blocked, screened form (featureMatchingGlobal's call: the pool against itself, k = 4).  Inside its own block a
            // row's top-4 (self included) comes from the exact f32 kernel; against every other block the f16 candidate
            // kernel of the pairwise matcher finds the three nearest with exact distances and a certified bound; the merge
            // yields the global top-4, bit-identical to the all-f32 path (APS_KNN_MODE=f32), in ~1/8 of its matrix time.
            // rows per block: 40 workgroups of the candidate kernel (APS_KNN_BLOCK: smaller blocks for tests)
            const int64_t bs = std::getenv("APS_KNN_BLOCK") ? std::max<int64_t>(512, std::atoll(std::getenv("APS_KNN_BLOCK")) / 128 * 128) : 20480;
            const int nb = (int)((ft + bs - 1) / bs);
            std::vector<int64_t> boff(nb + 1);
            for (int b = 0; b <= nb; ++b) boff[b] = std::min<int64_t>((int64_t)b * bs, ft);
            Ws<uint32_t> self_i((size_t)ft * 4);
            Ws<float> self_d((size_t)ft * 4);
            {
                Prof prof("knn_f32");  // every block against itself, one launch (the row tile tells the block)
                knn_f32_kernel<4><<<cdiv(ft, 128), 256, 0, stream()>>>(PT, sT, (int)ft, PT, sT, (int)ft, 4, self_i, self_d, 4, APS_ROWMAJOR,
                                                                      (int)bs);
            }
            check_launch("knn_f32_kernel");
            std::vector<int64_t> job_off;
            const int64_t slots = screened_block_top3(dT, ldt, layout, boff, job_off, nullptr, nullptr, nullptr);
            Ws<uint32_t> t3i((size_t)std::max<int64_t>(slots, 1) * 3);
            Ws<float> t3d((size_t)std::max<int64_t>(slots, 1) * 3), t3b((size_t)std::max<int64_t>(slots, 1));
            if (nb > 1) screened_block_top3(dT, ldt, layout, boff, job_off, t3i, t3d, t3b);
            Ws<int64_t> d_boff(nb + 1), d_joff((size_t)nb * nb);
            Ws<uint32_t> unc((size_t)ft);
            Ws<unsigned int> unc_n(1);
            APS_HIP(hipMemcpyAsync(d_boff, boff.data(), (nb + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemcpyAsync(d_joff, job_off.data(), (size_t)nb * nb * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemsetAsync(unc_n, 0, sizeof(unsigned int), stream()));
            {
                Prof prof("knn_merge");
                knn_merge_kernel<<<cdiv(ft, 256), 256, 0, stream()>>>(d_boff, nb, d_joff, self_i, self_d, t3i, t3d, t3b, ft, k, oi, od, ldo,
                                                                     layout, unc, unc_n);
            }
            check_launch("knn_merge_kernel");
            unsigned int n_unc = 0;
            APS_HIP(hipMemcpyAsync(&n_unc, unc_n, sizeof n_unc, hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
            if (n_unc) {
                // the few rows the merge could not certify (four or more near-identical rows in another block): their
                // exact top-4 in EVERY block, one wave per (row, block), then the global head of those lists
                Prof prof("knn_exact_rows");
                Ws<long long> li((size_t)n_unc * nb * 4);
                Ws<float> ldv((size_t)n_unc * nb * 4);
                knn_rows_block_top4_kernel<<<(unsigned)((size_t)n_unc * nb), 64, 0, stream()>>>(PT, sT, unc, (int)n_unc, d_boff, nb, li, ldv);
                knn_rows_merge_kernel<<<cdiv(n_unc, 64), 64, 0, stream()>>>(unc, (int)n_unc, nb, li, ldv, k, oi, od, ldo, layout);
                check_launch("knn exact rows");
            }
            if (std::getenv("APS_TRACE"))
                std::fprintf(stderr, "[aps] blocked k-NN: %d blocks, %u of %lld rows recomputed against the whole pool\n", nb, n_unc,
                             (long long)ft);
        } else {
            Prof prof("knn_f32");
            if (k <= 4)
                knn_f32_kernel<4><<<cdiv(fq, 128), 256, 0, stream()>>>(pq, sq, (int)fq, PT, sT, (int)ft, k, oi, od, ldo, layout, 0);
            else
                knn_f32_kernel<8><<<cdiv(fq, 128), 256, 0, stream()>>>(pq, sq, (int)fq, PT, sT, (int)ft, k, oi, od, ldo, layout, 0);
        }
        check_launch("knn_f32_kernel");
        oi.commit();
        od.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_knn_global_screened(const float* pool, int64_t f, int64_t ld, int dim, int layout, const int64_t* img_off, int n_img,
                            float ratio, int k, uint32_t* idx, float* dist, int64_t ldo) {
    return guarded([&] {
        APS_REQUIRE(dim == kDim, APS_E_DIM, "descriptor length %d not supported (built for %d-D SIFT)", dim, kDim);
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(k >= 1 && k <= 4, APS_E_ARG, "k in 1..4 (featureMatchingGlobal uses k = 4)");
        APS_REQUIRE(img_off && n_img >= 1 && img_off[0] == 0 && img_off[n_img] == f, APS_E_ARG, "img_off must span the pool");
        APS_REQUIRE(f >= 0 && f < (1ll << 31) && (f == 0 || (pool && idx && dist)), APS_E_ARG, "bad sizes / NULL argument");
        APS_REQUIRE(ratio > 0.f && std::isfinite(ratio), APS_E_ARG, "ratio must be positive");
        APS_REQUIRE(layout == APS_ROWMAJOR ? (ld >= dim && ldo >= k) : (ld >= f && ldo >= f), APS_E_DIM, "leading dimension too small");
        for (int i = 0; i < n_img; ++i) APS_REQUIRE(img_off[i + 1] >= img_off[i], APS_E_ARG, "img_off must not decrease");
        ctx();
        if (f == 0) return;
        const std::vector<int64_t> ioff(img_off, img_off + n_img + 1);
        std::vector<int64_t> job_off;
        if (f < 8192 || (std::getenv("APS_KNN_MODE") && !std::strcmp(std::getenv("APS_KNN_MODE"), "f32"))) {
            // a tiny pool, or the exact mode asked for: the plain search
            const int rc = aps_knn_global(pool, f, ld, pool, f, ld, dim, layout, k, idx, dist, ldo);
            if (rc != APS_OK) {
                const std::string why = aps_last_error();
                fail(rc, "%s", why.c_str());
            }
            return;
        }
        // The (row, image) tables are addressed with 32-bit slots: a pool with more than 2^31 of them (BASELINE configs[4]:
        // 500 images, 5.4 M rows = 2.7e9 slots) is searched in several passes over ranges of QUERY images - every pass sees
        // all images as columns, so every row still gets its exact neighbours in the whole pool.  APS_KNN_SLOT_CAP (test
        // hook) lowers the budget so that small pools take the chunked path too.
        int64_t slot_cap = ((int64_t)1 << 31) - 2;
        if (const char* e = std::getenv("APS_KNN_SLOT_CAP")) slot_cap = std::max<int64_t>(1, std::atoll(e));
        std::vector<int> cuts{0};  // query image ranges [cuts[c], cuts[c + 1])
        {
            int64_t run = 0;
            for (int i = 0; i < n_img; ++i) {
                const int64_t need = (ioff[i + 1] - ioff[i]) * (int64_t)n_img;
                if (need > ((int64_t)1 << 31) - 2) {
                    // one image alone exceeds the 32-bit slot range: no pass can hold its table - the plain exact search
                    // answers the whole call (slower, same result), as it did before the search was cut into passes
                    const int rc = aps_knn_global(pool, f, ld, pool, f, ld, dim, layout, k, idx, dist, ldo);
                    if (rc != APS_OK) {
                        const std::string why = aps_last_error();
                        fail(rc, "%s", why.c_str());
                    }
                    return;
                }
                if (run > 0 && run + need > slot_cap) {
                    cuts.push_back(i);
                    run = 0;
                }
                run += need;
            }
            cuts.push_back(n_img);
        }
        const size_t te = layout == APS_ROWMAJOR ? (size_t)(f - 1) * ld + dim : (size_t)(dim - 1) * ld + f;
        const size_t oe = layout == APS_ROWMAJOR ? (size_t)(f - 1) * ldo + k : (size_t)(k - 1) * ldo + f;
        In<float> dT(pool, te);
        Out<uint32_t> oi(idx, oe);
        Out<float> od(dist, oe);
        Ws<uint8_t> dismissed((size_t)f);
        Ws<int64_t> d_boff(n_img + 1), d_joff((size_t)n_img * n_img);
        Ws<uint32_t> unc((size_t)f);
        Ws<unsigned int> unc_n(1);
        APS_HIP(hipMemcpyAsync(d_boff, ioff.data(), (n_img + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
        APS_HIP(hipMemsetAsync(unc_n, 0, sizeof(unsigned int), stream()));
        int64_t n_surv = 0;
        struct PrepKeep {  // the images' operand forms are prepared by the first pass and reused by the others
            GlobalPrep* p = global_prep_new();
            ~PrepKeep() { global_prep_free(p); }
        } keep;
        for (size_t c = 0; c + 1 < cuts.size(); ++c) {
            const int ia = cuts[c], ib = cuts[c + 1];
            const int64_t slots = screened_global_top3(nullptr, ld, layout, ioff, ratio, job_off, nullptr, nullptr, nullptr, nullptr, nullptr, ia, ib);
            if (slots == 0) continue;
            Ws<uint32_t> t3i((size_t)slots * 3);
            Ws<float> t3d((size_t)slots * 3), t3b((size_t)slots);
            int64_t n_surv_c = 0;
            screened_global_top3(dT, ld, layout, ioff, ratio, job_off, t3i, t3d, t3b, dismissed, &n_surv_c, ia, ib, keep.p);
            n_surv += n_surv_c;
            APS_HIP(hipMemcpyAsync(d_joff, job_off.data(), (size_t)n_img * n_img * sizeof(int64_t), hipMemcpyHostToDevice, stream()));
            {
                Prof prof("knn_merge");
                knn_merge3_kernel<<<cdiv(ioff[ib] - ioff[ia], 256), 256, 0, stream()>>>(d_boff, n_img, d_joff, dismissed, t3i, t3d, t3b, ioff[ib], k, oi,
                                                                                        od, ldo, layout, unc, unc_n, ioff[ia]);
            }
            check_launch("knn_merge3_kernel");
            APS_HIP(hipStreamSynchronize(stream()));  // job_off is rebuilt by the next pass; the t3 tables go out of scope
        }
        unsigned int n_unc = 0;
        APS_HIP(hipMemcpyAsync(&n_unc, unc_n, sizeof n_unc, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        if (n_unc) {  // (as in aps_knn_global: the exact top-4 of those rows in every image, then the head of the lists)
            Prof prof("knn_exact_rows");
            Ws<float> PT((size_t)f * kDim), sT((size_t)f);
            knn_prep_kernel<<<cdiv(f, 64), 64, 0, stream()>>>(dT, f, ld, layout, PT, sT);
            Ws<long long> li((size_t)n_unc * n_img * 4);
            Ws<float> ldv((size_t)n_unc * n_img * 4);
            knn_rows_block_top4_kernel<<<(unsigned)((size_t)n_unc * n_img), 64, 0, stream()>>>(PT, sT, unc, (int)n_unc, d_boff, n_img, li, ldv);
            knn_rows_merge_kernel<<<cdiv(n_unc, 64), 64, 0, stream()>>>(unc, (int)n_unc, n_img, li, ldv, k, oi, od, ldo, layout);
            check_launch("knn exact rows");
            APS_HIP(hipStreamSynchronize(stream()));
        }
        if (std::getenv("APS_TRACE"))
            std::fprintf(stderr, "[aps] screened pooled k-NN: %d images, %lld of %lld rows searched, %u recomputed exactly\n", n_img,
                         (long long)n_surv, (long long)f, n_unc);
        g_global_rows = f;
        g_global_surv = n_surv;
        oi.commit();
        od.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_knn_global_screen_stats(int64_t* rows, int64_t* survivors) {
    if (rows) *rows = g_global_rows;
    if (survivors) *survivors = g_global_surv;
    return APS_OK;
}

int aps_global_normalize(const float* X, int64_t n, int64_t ld, int dim, int layout, float* out) {
    return guarded([&] {
        APS_REQUIRE(dim == kDim, APS_E_DIM, "descriptor length %d not supported (built for %d-D SIFT)", dim, kDim);
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(n >= 0 && (n == 0 || (X && out)), APS_E_ARG, "NULL argument");
        APS_REQUIRE(layout == APS_ROWMAJOR ? ld >= dim : ld >= n, APS_E_DIM, "leading dimension too small");
        ctx();
        if (n == 0) return;
        const size_t ne = layout == APS_ROWMAJOR ? (size_t)(n - 1) * ld + dim : (size_t)(dim - 1) * ld + n;
        In<float> dX(X, ne);
        Out<float> dO(out, (size_t)n * kDim);
        Prof prof("global_normalize");
        global_normalize_kernel<<<cdiv(n, 64), 64, 0, stream()>>>(dX, n, ld, layout, dO);
        check_launch("global_normalize_kernel");
        dO.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_global_filter(const uint32_t* nn_idx, const float* nn_dist, int64_t f, int k, int64_t ldn, int layout,
                      const uint32_t* img_idx, const uint32_t* local_idx, int n_img, float ratio,
                      int64_t* pair_ptr, uint32_t* idx_i, uint32_t* idx_j, int64_t cap, int64_t* count) {
    return guarded([&] {
        APS_REQUIRE(count && pair_ptr, APS_E_ARG, "count/pair_ptr is NULL");
        APS_REQUIRE(f >= 0 && k > 0 && n_img >= 0 && cap >= 0, APS_E_ARG, "bad sizes");
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(f < (1ll << 40), APS_E_DIM, "too many features");
        APS_REQUIRE(f == 0 || (nn_idx && nn_dist && img_idx && local_idx), APS_E_ARG, "NULL argument");
        ctx();
        const int64_t n_pairs = (int64_t)n_img * (n_img - 1) / 2;
        *count = 0;
        std::vector<int64_t> hp(std::max<int64_t>(n_pairs, 0) + 1, 0);
        if (f > 0 && n_pairs > 0) {
            const size_t ne = layout == APS_ROWMAJOR ? (size_t)(f - 1) * ldn + k : (size_t)(k - 1) * ldn + f;
            In<uint32_t> di(nn_idx, ne), dimg(img_idx, f), dloc(local_idx, f);
            In<float> dd(nn_dist, ne);
            Ws<unsigned long long> keys(f), sorted(f), pc(n_pairs);
            Ws<uint32_t> li(f), lj(f);
            APS_HIP(hipMemsetAsync(pc, 0, n_pairs * sizeof(unsigned long long), stream()));
            Prof prof("global_filter");
            global_filter_kernel<<<cdiv(f, 256), 256, 0, stream()>>>(di, dd, f, k, ldn, layout, dimg, dloc, ratio, keys, li, lj);
            check_launch("global_filter_kernel");
            size_t tb = 0;
            APS_HIP(rocprim::radix_sort_keys(nullptr, tb, keys.get(), sorted.get(), (unsigned)f, 0, 64, stream()));
            Ws<char> tmp(tb);
            APS_HIP(rocprim::radix_sort_keys(tmp.get(), tb, keys.get(), sorted.get(), (unsigned)f, 0, 64, stream()));
            // rejected queries carry key ~0 and sink to the end: count the accepted prefix on the host side
            std::vector<unsigned long long> hs(f);
            APS_HIP(hipMemcpyAsync(hs.data(), sorted, f * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
            const int64_t total = std::lower_bound(hs.begin(), hs.end(), ~0ull) - hs.begin();
            *count = total;
            for (int64_t e = 0; e < total; ++e) hp[(hs[e] >> 40) + 1]++;
            for (int64_t p = 0; p < n_pairs; ++p) hp[p + 1] += hp[p];
            if (total > cap) fail(APS_E_CAP, "output capacity %lld < %lld matches", (long long)cap, (long long)total);
            if (total > 0) {
                APS_REQUIRE(idx_i && idx_j, APS_E_ARG, "NULL output with matches present");
                Out<uint32_t> oi(idx_i, total), oj(idx_j, total);
                global_emit_kernel<<<cdiv(total, 256), 256, 0, stream()>>>(sorted, total, li, lj, oi, oj, pc);
                check_launch("global_emit_kernel");
                oi.commit();
                oj.commit();
                APS_HIP(hipStreamSynchronize(stream()));
            }
        }
        if (is_device_ptr(pair_ptr))
            APS_HIP(hipMemcpy(pair_ptr, hp.data(), hp.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        else
            std::copy(hp.begin(), hp.end(), pair_ptr);
    });
}

int aps_hamming_2nn(const uint8_t* A, int64_t n1, int64_t lda, const uint8_t* B, int64_t n2, int64_t ldb,
                    int nbytes, int layout, uint32_t* idx2, float* d1, float* d2) {
    return guarded([&] {
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(n1 >= 0 && n2 >= 0, APS_E_ARG, "negative size");
        APS_REQUIRE(nbytes > 0 && nbytes <= 64, APS_E_DIM, "byte width must be in 1..64 (ORB 32, BRISK 64)");
        APS_REQUIRE(n1 == 0 || (A && idx2 && d1 && d2), APS_E_ARG, "NULL argument");
        APS_REQUIRE(n2 == 0 || B, APS_E_ARG, "NULL B");
        if (layout == APS_ROWMAJOR)
            APS_REQUIRE(lda >= nbytes && ldb >= nbytes, APS_E_DIM, "Byte width mismatch.");  // hamm2nn:cols
        else
            APS_REQUIRE(lda >= n1 && ldb >= n2, APS_E_DIM, "leading dimension too small");
        ctx();
        if (n1 == 0) return;
        Out<uint32_t> oi(idx2, n1);
        Out<float> o1(d1, n1), o2(d2, n1);
        if (n2 == 0) {  // :42-45
            std::vector<uint32_t> z(n1, 0u);
            std::vector<float> nan(n1, NAN);
            APS_HIP(hipMemcpyAsync(oi.get(), z.data(), n1 * sizeof(uint32_t), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemcpyAsync(o1.get(), nan.data(), n1 * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipMemcpyAsync(o2.get(), nan.data(), n1 * sizeof(float), hipMemcpyHostToDevice, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
        } else {
            const size_t ae = layout == APS_ROWMAJOR ? (size_t)(n1 - 1) * lda + nbytes : (size_t)(nbytes - 1) * lda + n1;
            const size_t be = layout == APS_ROWMAJOR ? (size_t)(n2 - 1) * ldb + nbytes : (size_t)(nbytes - 1) * ldb + n2;
            In<uint8_t> dA(A, ae), dB(B, be);
            const int nw = nbytes <= 32 ? 8 : 16;
            Ws<uint32_t> pa((size_t)n1 * nw), pb((size_t)n2 * nw);
            pack_bytes_kernel<<<cdiv((size_t)n1 * nw, 256), 256, 0, stream()>>>(dA, n1, lda, nbytes, layout, nw, pa);
            pack_bytes_kernel<<<cdiv((size_t)n2 * nw, 256), 256, 0, stream()>>>(dB, n2, ldb, nbytes, layout, nw, pb);
            Prof prof("hamming_2nn");
            if (nw == 8)
                hamming_kernel<8><<<cdiv(n1, 256), 256, 0, stream()>>>(pa, n1, pb, n2, nbytes, oi, o1, o2);
            else
                hamming_kernel<16><<<cdiv(n1, 256), 256, 0, stream()>>>(pa, n1, pb, n2, nbytes, oi, o1, o2);
            check_launch("hamming_kernel");
        }
        oi.commit();
        o1.commit();
        o2.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_knn_hamming(const uint8_t* train, int64_t ft, int64_t ldt, const uint8_t* query, int64_t fq, int64_t ldq, int nbytes,
                    int layout, int k, uint32_t* idx, float* dist, int64_t ldo) {
    return guarded([&] {
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(k > 0, APS_E_ARG, "k must be > 0");  // flann_knn:k
        APS_REQUIRE(k <= 8, APS_E_ARG, "k <= 8 supported (featureMatchingGlobal uses k = 4)");
        APS_REQUIRE(nbytes > 0 && nbytes <= 64, APS_E_DIM, "byte width must be in 1..64 (ORB 32, BRISK 64)");
        APS_REQUIRE(ft >= 0 && fq >= 0 && ft < (1ll << 31) && fq < (1ll << 31), APS_E_ARG, "bad sizes");
        APS_REQUIRE(fq == 0 || (idx && dist && query), APS_E_ARG, "NULL argument");
        APS_REQUIRE(ft == 0 || train, APS_E_ARG, "NULL train");
        if (layout == APS_ROWMAJOR)
            APS_REQUIRE(ldt >= nbytes && ldq >= nbytes && ldo >= k, APS_E_DIM, "query must have same descriptor dimension as train");
        else
            APS_REQUIRE(ldt >= ft && ldq >= fq && ldo >= fq, APS_E_DIM, "leading dimension too small");
        ctx();
        if (fq == 0) return;
        const size_t te = ft == 0 ? 0 : (layout == APS_ROWMAJOR ? (size_t)(ft - 1) * ldt + nbytes : (size_t)(nbytes - 1) * ldt + ft);
        const size_t qe = layout == APS_ROWMAJOR ? (size_t)(fq - 1) * ldq + nbytes : (size_t)(nbytes - 1) * ldq + fq;
        const size_t oe = layout == APS_ROWMAJOR ? (size_t)(fq - 1) * ldo + k : (size_t)(k - 1) * ldo + fq;
        In<uint8_t> dT(train, te), dQ(query, qe);
        Out<uint32_t> oi(idx, oe);
        Out<float> od(dist, oe);
        const int nw = nbytes <= 32 ? 8 : 16;
        Ws<uint32_t> pt((size_t)std::max<int64_t>(ft, 1) * nw), pq((size_t)fq * nw);
        if (ft > 0) pack_bytes_kernel<<<cdiv((size_t)ft * nw, 256), 256, 0, stream()>>>(dT, ft, ldt, nbytes, layout, nw, pt);
        pack_bytes_kernel<<<cdiv((size_t)fq * nw, 256), 256, 0, stream()>>>(dQ, fq, ldq, nbytes, layout, nw, pq);
        {
            Prof prof("hamming_knn");
            const unsigned g = cdiv(fq, 256);
#define APS_HK(NWV, KV) hamming_knn_kernel<NWV, KV><<<g, 256, 0, stream()>>>(pq, fq, pt, ft, k, oi.get(), od.get(), ldo, layout)
            if (nw == 8) {
                if (k <= 2) APS_HK(8, 2); else if (k <= 4) APS_HK(8, 4); else APS_HK(8, 8);
            } else {
                if (k <= 2) APS_HK(16, 2); else if (k <= 4) APS_HK(16, 4); else APS_HK(16, 8);
            }
#undef APS_HK
        }
        check_launch("hamming_knn_kernel");
        oi.commit();
        od.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

}  // extern "C"
