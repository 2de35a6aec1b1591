// sift.hip — SIFT (Gaussian/DoG pyramid, extrema, orientation, 128-D descriptors) on gfx950.
//
// Stands behind PP/featureMatching/getFeaturePoints.m:26-40,71-74 (rgb2gray -> detectSIFTFeatures ->
// extractFeatures).  The toolbox functions are closed; the algorithm is Lowe 2004 in OpenCV cv::SIFT's
// parameterisation, with the order-independence choices documented in oracle/sift_oracle.c (fma-chain
// Gaussians, 2^-20 fixed-point histograms, polynomial exp/atan2/sincos) so that this path and the oracle
// agree bit for bit.
//
// Kernels
//   gray_up_kernel                 : uint8 (HWC or MATLAB planar) -> f32 gray (LDS only) -> 2x bilinear base.
//   blur_kernel<R>                 : separable Gaussian, row pass then column pass fused through an LDS
//                                    tile (reflect-101 border), both passes on v_pk_fma_f32; the blur of plane
//                                    nl also writes the next octave's base (every second pixel) from registers.
//   extrema_kernel                 : 26-neighbour test + Newton refinement + contrast/edge tests.
//   orient_kernel / descr_kernel   : one 64-lane wave per keypoint, histograms in LDS (int64 atomics).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#ifdef APS_DBG
#include <atomic>
#include <map>
#include <mutex>
#endif

#include "aps_internal.h"

#include <rocprim/rocprim.hpp>

namespace aps {

constexpr int kBorder = 5;
constexpr int kMaxInterp = 5;
constexpr int kOriBins = 36;
constexpr float kFix = 1048576.0f;

// (long long)rintf(v * kFix), the fixed-point addend of the orientation / descriptor histograms.  The pyramid comes
// from 8-bit pixels, so every plane stays inside [0, 255] (the blurs are convex combinations), a gradient magnitude is
// at most 255 sqrt(2) and every addend's magnitude is below 361 * 2^20 < 2^31: the one-instruction i32 conversion gives
// the same integer as the generic f32 -> i64 sequence (a dozen VALU instructions, eight times per sample).
__device__ __forceinline__ long long to_fix(float v) { return (long long)(int)rintf(v * kFix); }
constexpr float kFltEps = 1.1920928955078125e-07f;

// ---- elementary functions (identical formulas to oracle/sift_oracle.c) --------------------------
__device__ __forceinline__ float poly_exp2(float f) {
    float p = 1.5252733804059841e-05f;
    p = fmaf(p, f, 1.5403530393381609e-04f);
    p = fmaf(p, f, 1.3333558146428443e-03f);
    p = fmaf(p, f, 9.6181291076284772e-03f);
    p = fmaf(p, f, 5.5504108664821580e-02f);
    p = fmaf(p, f, 2.4022650695910071e-01f);
    p = fmaf(p, f, 6.9314718055994531e-01f);
    p = fmaf(p, f, 1.0f);
    return p;
}
__device__ __forceinline__ float my_exp2(float t) {
    if (t < -125.0f) return 0.0f;
    if (t > 125.0f) t = 125.0f;
    const float n = rintf(t);
    const float p = poly_exp2(t - n);
    return p * __uint_as_float((uint32_t)((int)n + 127) << 23);
}
__device__ __forceinline__ float my_exp(float x) { return my_exp2(x * 1.4426950408889634f); }

__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f;
    const float p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    // one division for both octants: min / (max + eps) is ay / (ax + eps) when ax >= ay and ax / (ay + eps) otherwise
    // (written as two branches, both quotients were computed for every sample)
    const bool steep = !(ax >= ay);
    const float c = (steep ? ax : ay) / ((steep ? ay : ax) + 2.220446049250313e-16f);
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (steep) a = 90.0f - a;
    if (x < 0) a = 180.0f - a;
    if (y < 0) a = 360.0f - a;
    return a;
}

__device__ __forceinline__ void sincos_deg(float a, float& s, float& c) {
    const float k = rintf(a / 90.0f);
    const float r = a - 90.0f * k;
    const float x = r * 0.017453292519943295f, x2 = x * x;
    float sp = 2.7557319223985893e-06f;
    sp = fmaf(sp, x2, -1.9841269841269841e-04f);
    sp = fmaf(sp, x2, 8.3333333333333332e-03f);
    sp = fmaf(sp, x2, -1.6666666666666666e-01f);
    sp = fmaf(sp * x2, x, x);
    float cp = -2.7557319223985888e-07f;
    cp = fmaf(cp, x2, 2.4801587301587302e-05f);
    cp = fmaf(cp, x2, -1.3888888888888889e-03f);
    cp = fmaf(cp, x2, 4.1666666666666664e-02f);
    cp = fmaf(cp, x2, -0.5f);
    cp = fmaf(cp, x2, 1.0f);
    const int q = (((int)k % 4) + 4) % 4;
    if (q == 0) {
        s = sp;
        c = cp;
    } else if (q == 1) {
        s = cp;
        c = -sp;
    } else if (q == 2) {
        s = -sp;
        c = -cp;
    } else {
        s = -cp;
        c = sp;
    }
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0)
            p = -p;
        else
            p = 2 * (n - 1) - p;
    }
    return p;
}

// ---- gray + base upsample -------------------------------------------------------------------------
// Gray conversion and 2x bilinear base in one pass: the gray values of a tile's footprint are computed into LDS and the
// doubled tile is interpolated from there (same expressions, so the same bits); the gray plane is never stored.
constexpr int kGUW = 128, kGUH = 16;  // output tile per 256-thread workgroup
__global__ __launch_bounds__(256) void gray_up_kernel(const uint8_t* __restrict__ img, int h, int w, int c, int layout,
                                                      float* __restrict__ out) {
    constexpr int GW = kGUW / 2 + 2, GH = kGUH / 2 + 2;
    __shared__ float s_g[GH][GW + 1];
    const int x0 = blockIdx.x * kGUW, y0 = blockIdx.y * kGUH;
    const int cx0 = x0 / 2 - 1, cy0 = y0 / 2 - 1;
    for (int e = threadIdx.x; e < GH * GW; e += 256) {
        const int ly = e / GW, lx = e - ly * GW;
        const int y = min(max(cy0 + ly, 0), h - 1), x = min(max(cx0 + lx, 0), w - 1);
        float v;
        if (c == 1) {
            v = (float)(layout == APS_IMG_U8_HWC ? img[(size_t)y * w + x] : img[(size_t)x * h + y]);
        } else {
            uint8_t ch[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                ch[k] = layout == APS_IMG_U8_HWC ? img[((size_t)y * w + x) * 3 + k]
                                                 : img[(size_t)k * h * w + (size_t)x * h + y];
            const double d = 0.298936021293775 * ch[0] + 0.587043074451121 * ch[1] + 0.114020904255103 * ch[2];
            v = (float)floor(d + 0.5);
        }
        s_g[ly][lx] = v;
    }
    __syncthreads();
    const int y = y0 + threadIdx.x / 16, xb = x0 + (threadIdx.x & 15) * 8;
    if (y >= 2 * h || xb >= 2 * w) return;
    float fy = ((float)y + 0.5f) * 0.5f - 0.5f;
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    if (sy < 0) {
        sy = 0;
        fy = 0;
    }
    if (sy >= h - 1) {
        sy = h - 1;
        fy = 0;
    }
    const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
    const float b0 = 1.0f - fy, b1 = fy;
    const float* r0 = s_g[sy - cy0];
    const float* r1 = s_g[sy1 - cy0];
    float res[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int x = xb + j;
        float fx = ((float)x + 0.5f) * 0.5f - 0.5f;
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) {
            sx = 0;
            fx = 0;
        }
        if (sx >= w - 1) {
            sx = w - 1;
            fx = 0;
        }
        const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
        const float a0 = 1.0f - fx, a1 = fx;
        const float h0 = r0[sx - cx0] * a0 + r0[sx1 - cx0] * a1;
        const float h1 = r1[sx - cx0] * a0 + r1[sx1 - cx0] * a1;
        res[j] = h0 * b0 + h1 * b1;
    }
    float* dst = out + (size_t)y * (2 * w) + xb;
    if (xb + 8 <= 2 * w && (w & 1) == 0) {
        reinterpret_cast<float4*>(dst)[0] = make_float4(res[0], res[1], res[2], res[3]);
        reinterpret_cast<float4*>(dst)[1] = make_float4(res[4], res[5], res[6], res[7]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (xb + j < 2 * w) dst[j] = res[j];
    }
}

__global__ void decimate_kernel(const float* __restrict__ in, int h, int w, int oh, int ow,
                                float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow) return;
    out[(size_t)y * ow + x] = in[(size_t)min(2 * y, h - 1) * w + min(2 * x, w - 1)];
}

// ---- separable Gaussian through LDS ------------------------------------------------------------------
struct GaussK {
    float k[64];
    int n;
};

constexpr int kTW = 64, kTH = 32;  // output tile per 256-thread workgroup

// v_pk_fma_f32 with the tap in an SGPR pair (k[t & ~1], k[(t & ~1) + 1]): op_sel / op_sel_hi pick its low or high half
// for BOTH result lanes.  One instruction = the next tap of two independent k-ascending fma chains.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int T>
__device__ __forceinline__ void pk_tap(f32x2& acc, const GaussK& gk, f32x2 v) {
    unsigned long long kk;  // the aligned tap pair holding k[T]
    __builtin_memcpy(&kk, &gk.k[T & ~1], sizeof kk);
    if (T & 1)
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(kk), "v"(v));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(kk), "v"(v));
}

// radii whose tile kernel keeps ONE LDS buffer for both passes (see blur_tile_passes)
template <int R>
constexpr bool kBlurOneBuffer = R >= 7;

// The two passes of blur_kernel / blur_base_kernel on a filled tile (s_in row-interleaved, see blur_kernel).
template <int R, int TW = kTW, int NT = 256>
__device__ __forceinline__ void blur_tile_passes(const float* s_in, float* s_row, const GaussK& gk, int x0, int y0, int h, int w,
                                                 float* __restrict__ out, float* __restrict__ dec, int dh, int dw) {
    constexpr int RP = (R + 3) & ~3, OFF = RP - R;
    constexpr int IW = TW + 2 * RP, IH = kTH + 2 * R;
    constexpr int IP2 = 2 * IW + 4;
    constexpr int RPITCH = TW + 2;
    const int tid = threadIdx.x;
    __syncthreads();
    // row pass: IH/2 row pairs x 8 segments of 8 outputs; consecutive lanes = consecutive row pairs
    // kBlurOneBuffer<R> (the large radii, round 4): s_row IS s_in - a thread's row-pass results (one work item per thread:
    // IH / 2 * 8 <= 256 for every R <= 12) wait in registers until every thread has read its window, then overwrite the
    // tile.  One barrier more, 19 instead of 32 KB of LDS at R = 10: eight workgroups per CU instead of five, which is what
    // a kernel whose fill -> row pass -> column pass -> store chain is latency per workgroup needs; the small radii are
    // memory-bound and lose a few per cent to the extra barrier (round 2's measurement), so they keep two buffers.
    constexpr bool one_buf = kBlurOneBuffer<R>;
    static_assert(!one_buf || (IH / 2) * (TW / 8) <= NT, "one work item per thread");
    for (int u = tid; u < (one_buf ? NT : (IH / 2) * (TW / 8)); u += NT) {
        const bool live = u < (IH / 2) * (TW / 8);
        const int uc = live ? u : 0;
        const int seg = uc / (IH / 2), p = uc - seg * (IH / 2), xb = seg * 8;
        const f32x2* src = reinterpret_cast<const f32x2*>(&s_in[p * IP2 + 2 * (OFF + xb)]);
        f32x2 v[8 + 2 * R], acc[8];
#pragma unroll
        for (int j = 0; j < 8 + 2 * R; ++j) v[j] = src[j];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = f32x2{0.f, 0.f};
        static_for<0, 2 * R + 1>([&](auto T) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk_tap<t>(acc[j], gk, v[j + t]);
        });
        if (one_buf) __syncthreads();  // every window has been read: the tile may be overwritten
        float* d0 = &s_row[(2 * p) * RPITCH + xb];
        if (live) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d0[j] = acc[j].x;
                d0[RPITCH + j] = acc[j].y;
            }
        }
    }
    __syncthreads();
    // column pass: 32 column pairs x kTH/4 groups of 4 rows
    for (int u = tid; u < (TW / 2) * (kTH / 4); u += NT) {
        const int lx = 2 * (u & (TW / 2 - 1)), yb = (u / (TW / 2)) * 4;
        f32x2 v[4 + 2 * R], acc[4];
#pragma unroll
        for (int j = 0; j < 4 + 2 * R; ++j) v[j] = *reinterpret_cast<const f32x2*>(&s_row[(yb + j) * RPITCH + lx]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x2{0.f, 0.f};
        static_for<0, 2 * R + 1>([&](auto T) {
            constexpr int t = decltype(T)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) pk_tap<t>(acc[j], gk, v[j + t]);
        });
        const int gx = x0 + lx;
        if (gx + 1 < w && (w & 1) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + yb + j;
                if (gy < h) *reinterpret_cast<f32x2*>(&out[(size_t)gy * w + gx]) = acc[j];
            }
        } else if (gx < w) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gy = y0 + yb + j;
                if (gy < h) {
                    out[(size_t)gy * w + gx] = acc[j].x;
                    if (gx + 1 < w) out[(size_t)gy * w + gx + 1] = acc[j].y;
                }
            }
        }
        if (dec && (gx >> 1) < dw) {  // gx, y0 + yb are even
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int dy = (y0 + yb + j) >> 1;
                if (dy < dh) dec[(size_t)dy * dw + (gx >> 1)] = acc[j].x;
            }
        }
    }
}

template <int R, int TW = kTW, int NT = 256>
__global__ __launch_bounds__(NT) void blur_kernel(const float* __restrict__ in, int h, int w, GaussK gk,
                                                   float* __restrict__ out, float* __restrict__ dec, int dh, int dw) {
    // dec (optional): the next octave's base plane, out(2y, 2x) for y < dh, x < dw - written from the registers that
    // hold the result instead of by a decimation pass that re-reads the plane.
    // The haloed input tile is fetched in 16-byte pieces: the horizontal halo is rounded up to a multiple of four
    // pixels so that every piece is aligned when the row pitch is (the kernel was instruction-bound on its
    // dword-per-thread tile fill, not on HBM).  Pieces that cross the image border fall back to reflected scalars.
    constexpr int RP = (R + 3) & ~3;
    constexpr int IW = TW + 2 * RP, IH = kTH + 2 * R, NV = IW / 4;
    static_assert(IH % 2 == 0 && kTH % 4 == 0, "row pairs");
    // Both passes run two fma chains per v_pk_fma_f32, and a packed operand must be an aligned register pair.  A pair
    // of horizontally adjacent inputs is aligned for every other tap only, so the passes pair the OTHER direction:
    //   row pass   : one lane = 8 outputs of TWO consecutive rows; the tile is stored row-interleaved
    //                (s_in[(y >> 1)][x][y & 1]), so the window of both rows is one contiguous run of aligned pairs;
    //   column pass: one lane = 4 outputs of TWO adjacent columns of the (plainly stored) row-pass result.
    // Every accumulator still sees its taps in ascending order: the scalar form's chain, bit for bit.
    // Pitches (floats): 4 * odd, so that consecutive row pairs land on distinct 16-byte bank groups.
    constexpr int IP2 = 2 * IW + 4;
    constexpr int RPITCH = TW + 2;  // column pass: 8-byte reads by consecutive lanes; row pass: dword writes, rows 2 apart
    static_assert((IP2 / 4) % 2 == 1, "pitch");
    static_assert(IH * RPITCH <= (IH / 2) * IP2, "the row-pass result fits the tile's buffer");
    __shared__ __attribute__((aligned(16))) float s_in[(IH / 2) * IP2];
    __shared__ __attribute__((aligned(16))) float s_row_own[kBlurOneBuffer<R> ? 1 : IH * RPITCH];
    float* const s_row = kBlurOneBuffer<R> ? s_in : s_row_own;
    // (Measured and dropped, round 5: an XCD-contiguous tile order - XCD k walks the k-th eighth of the tile list, so that
    // the halo columns two neighbours share sit in one L2 - made every octave-0 launch 7-12 us SLOWER (R = 4: 57.7 -> 64.6 us):
    // the round-robin deal spreads a plane's rows over all memory channels at any moment, the contiguous order does not.)
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * kTH;
    const int tid = threadIdx.x;
    const bool vec_ok = (w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    // All of a thread's pieces are requested before the first one is parked in LDS: as a plain loop (request, wait, store,
    // next) every workgroup paid the memory latency NPF times in series before its first barrier.
    constexpr int NPF = (IH * NV + NT - 1) / NT;
    float4 pf[NPF];
    // Interior tiles (the haloed patch lies inside the plane: all but the rim, 94 % of the tiles of a 7680 x 4320 plane) take
    // their pieces at 32-bit offsets from one scalar base, without the reflection, the border tests and the scalar
    // fallback of the general form below - that bookkeeping was ~25 vector instructions per piece, 125 of the 250-470 a
    // thread executes (round 5).
    const bool interior = vec_ok && x0 >= RP && x0 + TW + RP <= w && y0 >= R && y0 + kTH + R <= h;
    if (interior) {
        const float* base = in + ((size_t)(y0 - R) * w + (x0 - RP));
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = tid + NT * q;
            const int ec = e < IH * NV ? e : IH * NV - 1;
            const int ly = ec / NV, v = ec - ly * NV;
            pf[q] = *reinterpret_cast<const float4*>(base + (unsigned)(ly * w + 4 * v));  // (scalar base + 32-bit lane offset)
        }
    } else {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = tid + NT * q;
            const int ec = e < IH * NV ? e : IH * NV - 1;  // (threads past the end re-read the last piece and drop it)
            const int ly = ec / NV, v = ec - ly * NV;
            const int gy = reflect101(y0 + ly - R, h), gx = x0 - RP + 4 * v;
            const float* row = in + (size_t)gy * w;
            if (vec_ok && gx >= 0 && gx + 3 < w) {
                pf[q] = *reinterpret_cast<const float4*>(row + gx);
            } else {
                pf[q].x = row[reflect101(gx, w)];
                pf[q].y = row[reflect101(gx + 1, w)];
                pf[q].z = row[reflect101(gx + 2, w)];
                pf[q].w = row[reflect101(gx + 3, w)];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int e = tid + NT * q;
        if (e < IH * NV) {
            const int ly = e / NV, v = e - ly * NV;
            float* dst = &s_in[(ly >> 1) * IP2 + 8 * v + (ly & 1)];
            dst[0] = pf[q].x;
            dst[2] = pf[q].y;
            dst[4] = pf[q].z;
            dst[6] = pf[q].w;
        }
    }
    blur_tile_passes<R, TW, NT>(s_in, s_row, gk, x0, y0, h, w, out, dec, dh, dw);
}

// Gray plane of the source image as bytes, row-major (round 5): rgb2gray's value is an integer 0..255 (floor(d + 0.5) of the
// f64 combination), so a uint8 plane holds it exactly.  blur_base_kernel used to convert the RGB footprint of every tile
// itself - 936 footprint pixels for 512 source pixels, three byte loads and the f64 arithmetic each, a quarter of the
// kernel's vector instructions; now the conversion runs once per source pixel (25 MB read, 8 MB written per 4K view) and
// the tiles read bytes.
__global__ __launch_bounds__(256) void gray_u8_kernel(const uint8_t* __restrict__ img, int h, int w, int c, int layout,
                                                      uint8_t* __restrict__ gray) {
    const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= w) return;
    uint8_t g[4];
    const bool fast = c == 3 && layout == APS_IMG_U8_HWC && (w & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 3) == 0;
    if (fast) {  // 4 pixels = 12 bytes = three aligned dwords
        const uint32_t* p = reinterpret_cast<const uint32_t*>(img + ((size_t)y * w + x4) * 3);
        const uint32_t a = p[0], b = p[1], d_ = p[2];
        const uint8_t ch[12] = {(uint8_t)a, (uint8_t)(a >> 8), (uint8_t)(a >> 16), (uint8_t)(a >> 24), (uint8_t)b, (uint8_t)(b >> 8),
                                (uint8_t)(b >> 16), (uint8_t)(b >> 24), (uint8_t)d_, (uint8_t)(d_ >> 8), (uint8_t)(d_ >> 16), (uint8_t)(d_ >> 24)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double d = 0.298936021293775 * ch[3 * k] + 0.587043074451121 * ch[3 * k + 1] + 0.114020904255103 * ch[3 * k + 2];
            g[k] = (uint8_t)(float)floor(d + 0.5);
        }
        *reinterpret_cast<uint32_t*>(gray + (size_t)y * w + x4) = g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
        return;
    }
    for (int k = 0; k < 4 && x4 + k < w; ++k) {
        const int x = x4 + k;
        float v;
        if (c == 1) {
            v = (float)(layout == APS_IMG_U8_HWC ? img[(size_t)y * w + x] : img[(size_t)x * h + y]);
        } else {
            uint8_t ch[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                ch[q] = layout == APS_IMG_U8_HWC ? img[((size_t)y * w + x) * 3 + q] : img[(size_t)q * h * w + (size_t)x * h + y];
            const double d = 0.298936021293775 * ch[0] + 0.587043074451121 * ch[1] + 0.114020904255103 * ch[2];
            v = (float)floor(d + 0.5);
        }
        gray[(size_t)y * w + x] = (uint8_t)v;
    }
}

// The base plane of octave 0 in one pass: gray conversion, 2x bilinear upsample (gray_up_kernel's expressions, so the same
// bits) and the first blur.  The tile's inputs are interpolated straight into the blur's LDS tile from the gray values of
// the tile's footprint (computed into LDS first); neither the gray plane nor the doubled plane is ever stored - a 4K view
// saves the 133 MB write of gray_up_kernel and the 133 MB read of the blur.  img is sh x sw; the output plane 2sh x 2sw.
template <int R>
__global__ __launch_bounds__(256) void blur_base_kernel(const uint8_t* __restrict__ gray, int sh, int sw,
                                                        GaussK gk, float* __restrict__ out) {
    constexpr int RP = (R + 3) & ~3;
    constexpr int IW = kTW + 2 * RP, IH = kTH + 2 * R, NV = IW / 4;
    constexpr int IP2 = 2 * IW + 4;
    constexpr int RPITCH = kTW + 2;
    constexpr int GW = IW / 2 + 3, GH = IH / 2 + 3;  // gray footprint of the haloed tile (+ the interpolation's neighbours)
    __shared__ __attribute__((aligned(16))) float s_in[(IH / 2) * IP2];
    __shared__ __attribute__((aligned(16))) float s_row_own[kBlurOneBuffer<R> ? 1 : IH * RPITCH];
    float* const s_row = kBlurOneBuffer<R> ? s_in : s_row_own;
    __shared__ float s_g[GH][GW + 1];
    static_assert(GH * IW <= IH * RPITCH && IW % 4 == 0, "the horizontally interpolated gray rows fit the row-pass buffer");
    __shared__ __attribute__((aligned(16))) float s_h_own[kBlurOneBuffer<R> ? GH * IW : 4];
    const int h = 2 * sh, w = 2 * sw;
    const int x0 = blockIdx.x * kTW, y0 = blockIdx.y * kTH;
    const int tid = threadIdx.x;
    // source rows / columns of the doubled pixels [lo, hi] that exist (reflected halo pixels of outputs inside the plane fall
    // into the same range): doubled pixel p reads sources floor(p/2 - 1/4) and the next one
    const int lo_x = max(x0 - RP, 0), lo_y = max(y0 - R, 0);
    const int cx0 = max((lo_x + 1) / 2 - 1, 0), cy0 = max((lo_y + 1) / 2 - 1, 0);
    for (int e = tid; e < GH * GW; e += 256) {
        const int ly = e / GW, lx = e - ly * GW;
        const int y = min(cy0 + ly, sh - 1), x = min(cx0 + lx, sw - 1);
        s_g[ly][lx] = (float)gray[(size_t)y * sw + x];  // (gray_u8_kernel: rgb2gray's integer value)
    }
    // the interpolation's row and column terms (index of the first source in s_g, index of the second, second weight), once
    // per tile row / column instead of once per pixel; same expressions as gray_up_kernel.  (The clamps only act on halo
    // pixels of outputs beyond the plane's edge, whose results are never stored.)
    __shared__ int s_ci[IW + IH][2];
    __shared__ float s_cf[IW + IH];
    if (tid < IW + IH) {
        const bool col = tid < IW;
        const int n = col ? sw : sh, c0 = col ? cx0 : cy0, lim = col ? GW : GH;
        const int p = col ? reflect101(x0 - RP + tid, w) : reflect101(y0 + (tid - IW) - R, h);
        float f = ((float)p + 0.5f) * 0.5f - 0.5f;
        int s0 = (int)floorf(f);
        f -= (float)s0;
        if (s0 < 0) {
            s0 = 0;
            f = 0;
        }
        if (s0 >= n - 1) {
            s0 = n - 1;
            f = 0;
        }
        const int s1 = s0 + 1 < n ? s0 + 1 : n - 1;
        s_ci[tid][0] = min(max(s0 - c0, 0), lim - 1);
        s_ci[tid][1] = min(max(s1 - c0, 0), lim - 1);
        s_cf[tid] = f;
    }
    __syncthreads();
    // The interpolation is separable and its horizontal half depends on (gray row, tile column) only: every gray row is
    // interpolated ONCE along x into s_h (GH x IW values) and a doubled pixel is the vertical combination of two of those
    // - h0 = r0[i0] a0 + r0[i1] a1 was evaluated for every doubled pixel, i.e. 3.5 times per (gray row, column) on average;
    // same expressions on the same values, so the same bits (the kernel is vector-issue bound: 673 instructions per wave,
    // half of them here, profiles/r04d_pmc_blur_base.txt).  s_h lives in the row-pass buffer, which is idle until then.
    float* const s_h = kBlurOneBuffer<R> ? s_h_own : s_row;
    for (int e = tid; e < GH * IW; e += 256) {
        const int gy = e / IW, cx = e - gy * IW;
        const float a1 = s_cf[cx], a0 = 1.0f - a1;
        const float* r = s_g[gy];
        s_h[gy * IW + cx] = r[s_ci[cx][0]] * a0 + r[s_ci[cx][1]] * a1;
    }
    __syncthreads();
    for (int e = tid; e < IH * NV; e += 256) {
        const int ly = e / NV, v = e - ly * NV;
        const float b1 = s_cf[IW + ly], b0 = 1.0f - b1;
        const float4 h0 = *reinterpret_cast<const float4*>(&s_h[s_ci[IW + ly][0] * IW + 4 * v]);
        const float4 h1 = *reinterpret_cast<const float4*>(&s_h[s_ci[IW + ly][1] * IW + 4 * v]);
        float* dst = &s_in[(ly >> 1) * IP2 + 8 * v + (ly & 1)];
        dst[0] = h0.x * b0 + h1.x * b1;
        dst[2] = h0.y * b0 + h1.y * b1;
        dst[4] = h0.z * b0 + h1.z * b1;
        dst[6] = h0.w * b0 + h1.w * b1;
    }
    blur_tile_passes<R>(s_in, s_row, gk, x0, y0, h, w, out, nullptr, 0, 0);
}

// ---- the same blur, marching (experiment, APS_BLUR_MARCH=1; see launch_blur for the measurement) ------------------------
// The tile kernel above fetches (kTW + 2 RP) x (kTH + 2 R) inputs for kTW x kTH outputs - 1.4x (R = 4) to 2.2x (R = 10)
// of the plane through L2 - and row-filters the vertical halo of every tile again.  Here a workgroup owns kSW columns and
// `ch` rows of the plane and walks down them kRS input rows at a time: the next step's rows are requested before the
// current ones are consumed, the row pass runs on the kRS new rows only, its results live in a ring of NR rows in LDS, and
// the column pass produces the kRS output rows whose window the ring now holds.  A chunk re-reads the 2 R rows above it;
// nothing else is read twice.  Same chains (pk_tap, taps ascending), so the same bits as blur_kernel.
constexpr int kSW = 128, kRS = 16;
template <int R>
__global__ __launch_bounds__(256) void blur_march_kernel(const float* __restrict__ in, int h, int w, GaussK gk,
                                                         float* __restrict__ out, float* __restrict__ dec, int dh, int dw, int ch) {
    constexpr int RP = (R + 3) & ~3, OFF = RP - R;
    constexpr int IW = kSW + 2 * RP, NV = IW / 4;
    constexpr int LAG = (2 * R + kRS - 1) / kRS;       // steps between a row's arrival and the output batch that ends at it
    constexpr int NR = kRS * (LAG + 1) <= 32 ? 32 : 64;  // ring rows (a power of two >= kRS (LAG + 1))
    static_assert(kRS * (LAG + 1) <= NR && kRS % 4 == 0, "ring");
    constexpr int IP2 = 2 * IW + 4;   // floats per row PAIR of the interleaved input rows (see blur_kernel)
    constexpr int RPITCH = kSW + 2;
    static_assert((IP2 / 4) % 2 == 1, "pitch");
    __shared__ __attribute__((aligned(16))) float s_in[(kRS / 2) * IP2];
    __shared__ __attribute__((aligned(16))) float s_row[NR * RPITCH];
    const int x0 = blockIdx.x * kSW, y0 = blockIdx.y * ch, y1 = min(y0 + ch, h);
    const int tid = threadIdx.x;
    const bool vec_ok = (w & 3) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0;
    constexpr int NPF = (kRS * NV + 255) / 256;
    float4 pf[NPF];
    // input rows of step t: y0 - R + kRS t .. + kRS - 1 (reflected into the plane)
    auto fetch = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = tid + 256 * q;
            const int ec = e < kRS * NV ? e : kRS * NV - 1;
            const int ly = ec / NV, v = ec - ly * NV;
            const int gy = reflect101(y0 - R + kRS * t + ly, h), gx = x0 - RP + 4 * v;
            const float* row = in + (size_t)gy * w;
            if (vec_ok && gx >= 0 && gx + 3 < w) {
                pf[q] = *reinterpret_cast<const float4*>(row + gx);
            } else {
                pf[q].x = row[reflect101(gx, w)];
                pf[q].y = row[reflect101(gx + 1, w)];
                pf[q].z = row[reflect101(gx + 2, w)];
                pf[q].w = row[reflect101(gx + 3, w)];
            }
        }
    };
    const int n_batches = (y1 - y0 + kRS - 1) / kRS;  // output batches of kRS rows
    const int n_steps = n_batches + LAG;
    fetch(0);
    for (int t = 0; t < n_steps; ++t) {
        // a. park the rows of step t (row-interleaved pairs), request those of step t + 1
#pragma unroll
        for (int q = 0; q < NPF; ++q) {
            const int e = tid + 256 * q;
            if (e < kRS * NV) {
                const int ly = e / NV, v = e - ly * NV;
                float* dst = &s_in[(ly >> 1) * IP2 + 8 * v + (ly & 1)];
                dst[0] = pf[q].x;
                dst[2] = pf[q].y;
                dst[4] = pf[q].z;
                dst[6] = pf[q].w;
            }
        }
        if (t + 1 < n_steps) fetch(t + 1);
        __syncthreads();
        // b. row pass on the kRS new rows: kRS / 2 row pairs x kSW / 8 segments of 8 outputs -> ring rows kRS t ...
        for (int u = tid; u < (kRS / 2) * (kSW / 8); u += 256) {
            const int seg = u / (kRS / 2), p = u - seg * (kRS / 2), xb = seg * 8;
            const f32x2* src = reinterpret_cast<const f32x2*>(&s_in[p * IP2 + 2 * (OFF + xb)]);
            f32x2 v[8 + 2 * R], acc[8];
#pragma unroll
            for (int j = 0; j < 8 + 2 * R; ++j) v[j] = src[j];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = f32x2{0.f, 0.f};
            static_for<0, 2 * R + 1>([&](auto T) {
                constexpr int tt = decltype(T)::value;
#pragma unroll
                for (int j = 0; j < 8; ++j) pk_tap<tt>(acc[j], gk, v[j + tt]);
            });
            const int rel = kRS * t + 2 * p;  // input row index relative to y0 - R
            float* d0 = &s_row[(rel & (NR - 1)) * RPITCH + xb];
            float* d1 = &s_row[((rel + 1) & (NR - 1)) * RPITCH + xb];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d0[j] = acc[j].x;
                d1[j] = acc[j].y;
            }
        }
        __syncthreads();
        // c. column pass for output batch k = t - LAG: output row y0 + o reads ring rows o .. o + 2 R
        const int k = t - LAG;
        if (k >= 0) {
            const int u = tid;  // 64 column pairs x kRS / 4 groups of 4 rows = 256 work items
            const int lx = 2 * (u & (kSW / 2 - 1)), yb = kRS * k + (u / (kSW / 2)) * 4;
            f32x2 v[4 + 2 * R], acc[4];
#pragma unroll
            for (int j = 0; j < 4 + 2 * R; ++j) v[j] = *reinterpret_cast<const f32x2*>(&s_row[((yb + j) & (NR - 1)) * RPITCH + lx]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = f32x2{0.f, 0.f};
            static_for<0, 2 * R + 1>([&](auto T) {
                constexpr int tt = decltype(T)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk_tap<tt>(acc[j], gk, v[j + tt]);
            });
            const int gx = x0 + lx;
            if (gx + 1 < w && (w & 1) == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gy = y0 + yb + j;
                    if (gy < y1) *reinterpret_cast<f32x2*>(&out[(size_t)gy * w + gx]) = acc[j];
                }
            } else if (gx < w) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gy = y0 + yb + j;
                    if (gy < y1) {
                        out[(size_t)gy * w + gx] = acc[j].x;
                        if (gx + 1 < w) out[(size_t)gy * w + gx + 1] = acc[j].y;
                    }
                }
            }
            if (dec && (gx >> 1) < dw) {  // gx, y0 + yb are even (ch and kRS are)
#pragma unroll
                for (int j = 0; j < 4; j += 2) {
                    const int gy = y0 + yb + j, dy = gy >> 1;
                    if (gy < y1 && dy < dh) dec[(size_t)dy * dw + (gx >> 1)] = acc[j].x;
                }
            }
        }
        __syncthreads();  // the next step parks into s_in and writes ring rows this one has read
    }
}

// generic fallback for unusual radii: two plain passes through global memory (same arithmetic)
__global__ void blur_row_generic(const float* __restrict__ in, int h, int w, GaussK gk, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = gk.n / 2;
    float acc = 0.f;
    for (int t = 0; t < gk.n; ++t) acc = fmaf(gk.k[t], in[(size_t)y * w + reflect101(x + t - r, w)], acc);
    out[(size_t)y * w + x] = acc;
}
__global__ void blur_col_generic(const float* __restrict__ tmp, int h, int w, GaussK gk, float* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int r = gk.n / 2;
    float acc = 0.f;
    for (int t = 0; t < gk.n; ++t) acc = fmaf(gk.k[t], tmp[(size_t)reflect101(y + t - r, h) * w + x], acc);
    out[(size_t)y * w + x] = acc;
}

// ---- pyramid description on the device ------------------------------------------------------------------
struct OctaveDesc {
    const float* G[8];  // nl + 3 <= 8 Gaussian planes (DoG_p = G[p+1] - G[p] is formed by its consumers)
    int w, h;
};

struct KpRec {
    unsigned long long key;  // (o << 40) | (layer << 32) | (r << 16) | c
    float xc, xr, xi, contr;
};

// DoG value of plane L at (rr, cc): G[L+1] - G[L].  The DoG planes are never stored: every consumer subtracts
// the two Gaussian planes itself, which is the same single f32 subtraction a stored plane would hold.
#define AT(L, rr, cc) (od.G[(L) + 1][(size_t)(rr) * w + (cc)] - od.G[(L)][(size_t)(rr) * w + (cc)])

__device__ bool adjust_extremum(const OctaveDesc& od, int nl, int o, int layer, int r, int c, float contr_thr,
                                float edge_thr, KpRec& kp) {
    const int w = od.w, h = od.h;
    const float img_scale = 1.0f / 255.0f, deriv_scale = img_scale * 0.5f, second_scale = img_scale,
                cross_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < kMaxInterp; ++i) {
        const int im = layer, pv = layer - 1, nx = layer + 1;
        const float dD0 = (AT(im, r, c + 1) - AT(im, r, c - 1)) * deriv_scale;
        const float dD1 = (AT(im, r + 1, c) - AT(im, r - 1, c)) * deriv_scale;
        const float dD2 = (AT(nx, r, c) - AT(pv, r, c)) * deriv_scale;
        const float v2 = AT(im, r, c) * 2.0f;
        const float dxx = (AT(im, r, c + 1) + AT(im, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(im, r + 1, c) + AT(im, r - 1, c) - v2) * second_scale;
        const float dss = (AT(nx, r, c) + AT(pv, r, c) - v2) * second_scale;
        const float dxy = (AT(im, r + 1, c + 1) - AT(im, r + 1, c - 1) - AT(im, r - 1, c + 1) + AT(im, r - 1, c - 1)) * cross_scale;
        const float dxs = (AT(nx, r, c + 1) - AT(nx, r, c - 1) - AT(pv, r, c + 1) + AT(pv, r, c - 1)) * cross_scale;
        const float dys = (AT(nx, r + 1, c) - AT(nx, r - 1, c) - AT(pv, r + 1, c) + AT(pv, r - 1, c)) * cross_scale;
        float A[3][4] = {{dxx, dxy, dxs, dD0}, {dxy, dyy, dys, dD1}, {dxs, dys, dss, dD2}};
        bool singular = false;
#pragma unroll
        for (int col = 0; col < 3; ++col) {
            if (singular) break;
            int piv = col;
#pragma unroll
            for (int row = col + 1; row < 3; ++row)
                if (fabsf(A[row][col]) > fabsf(A[piv][col])) piv = row;
            if (fabsf(A[piv][col]) < kFltEps) {
                singular = true;
                break;
            }
            if (piv != col) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = A[piv][e];
                    A[piv][e] = A[col][e];
                    A[col][e] = t;
                }
            }
            const float d = -1.0f / A[col][col];
#pragma unroll
            for (int row = col + 1; row < 3; ++row) {
                const float alpha = A[row][col] * d;
#pragma unroll
                for (int e = col + 1; e < 4; ++e) A[row][e] = fmaf(alpha, A[col][e], A[row][e]);
            }
        }
        float X[3] = {0, 0, 0};
        if (!singular) {
#pragma unroll
            for (int row = 2; row >= 0; --row) {
                float s = A[row][3];
#pragma unroll
                for (int e = row + 1; e < 3; ++e) s = s - A[row][e] * X[e];
                X[row] = s / A[row][row];
            }
        }
        xi = -X[2];
        xr = -X[1];
        xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > 7.158278826666667e8f || fabsf(xr) > 7.158278826666667e8f || fabsf(xc) > 7.158278826666667e8f)
            return false;
        c += (int)rintf(xc);
        r += (int)rintf(xr);
        layer += (int)rintf(xi);
        if (layer < 1 || layer > nl || c < kBorder || c >= w - kBorder || r < kBorder || r >= h - kBorder) return false;
    }
    if (i >= kMaxInterp) return false;
    const int im = layer, pv = layer - 1, nx = layer + 1;
    const float dD0 = (AT(im, r, c + 1) - AT(im, r, c - 1)) * deriv_scale;
    const float dD1 = (AT(im, r + 1, c) - AT(im, r - 1, c)) * deriv_scale;
    const float dD2 = (AT(nx, r, c) - AT(pv, r, c)) * deriv_scale;
    const float t = (dD0 * xc + dD1 * xr) + dD2 * xi;
    const float contr = AT(im, r, c) * img_scale + t * 0.5f;
    if (fabsf(contr) * (float)nl < contr_thr) return false;
    const float v2 = AT(im, r, c) * 2.0f;
    const float dxx = (AT(im, r, c + 1) + AT(im, r, c - 1) - v2) * second_scale;
    const float dyy = (AT(im, r + 1, c) + AT(im, r - 1, c) - v2) * second_scale;
    const float dxy = (AT(im, r + 1, c + 1) - AT(im, r + 1, c - 1) - AT(im, r - 1, c + 1) + AT(im, r - 1, c - 1)) * cross_scale;
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return false;
    kp.key = ((unsigned long long)o << 40) | ((unsigned long long)layer << 32) | ((unsigned long long)r << 16) |
             (unsigned long long)c;
    kp.xc = xc;
    kp.xr = xr;
    kp.xi = xi;
    kp.contr = fabsf(contr);
    return true;
}

// 128 x 8 pixel tile per 256-thread workgroup; all nl+2 DoG planes of the tile (plus a 1-pixel halo) are staged
// in LDS once, so every plane is read from HBM ~1.3x instead of 27x per layer through the caches.  Tile shapes
// measured on one 4K view (us per view, all octaves): 32x24 573, 64x16 437, 64x12 420, 128x4 454, 128x8 405, 256x4 403,
// 128x12 416, 128x16 523, 256x8 535 - the rows a workgroup reads should be long, the LDS tile small enough for several
// workgroups per CU.
#ifndef APS_EH
#define APS_EH 8
#endif
#ifndef APS_EW
#define APS_EW 128
#endif
constexpr int kEW = APS_EW, kEH = APS_EH, kEGroups = 256 / kEW, kERows = kEH / kEGroups;  // rows of a column owned by one thread
static_assert(kEH % kEGroups == 0 && 256 % kEW == 0, "row groups");

template <int nl>  // NumLayersInOctave: nl + 3 Gaussian planes, nl + 2 DoG planes (sizes the LDS tile and the loops)
__global__ __launch_bounds__(256) void extrema_kernel(OctaveDesc od, int o, float thr,
                                                      unsigned long long* __restrict__ cells,
                                                      unsigned int* __restrict__ count, unsigned int cap) {
    // LDS rows hold pixels x0-4 .. x0+kEW+3 (the 1-pixel halo rounded out to 16-byte pieces; pitch TW)
    constexpr int TW = kEW + 8, TH = kEH + 2, NV = TW / 4, HX = 3;  // HX: LDS column of pixel x0-1
    constexpr int NG = nl + 3, ND = nl + 2;
    __shared__ __attribute__((aligned(16))) float s_d[ND][TH * TW];
    const int w = od.w, h = od.h;
    const int x0 = blockIdx.x * kEW, y0 = blockIdx.y * kEH;
    const int tid = threadIdx.x;
    // DoG planes of the tile, formed here from the nl + 3 Gaussian planes (they are not stored anywhere)
    const bool vec_ok = (w & 3) == 0;
    for (int e = tid; e < TH * NV; e += 256) {
        const int ly = e / NV, v = e - ly * NV;
        const int gy = min(max(y0 + ly - 1, 0), h - 1), gx = x0 - 4 + 4 * v;
        const size_t rowoff = (size_t)gy * w;
        // All plane loads are issued back to back (planes beyond nl + 3 re-read plane 0 and are ignored): with an
        // early exit between the loads the compiler waits for each one before testing the next, and the seven
        // round trips in series made this kernel latency-bound (83 % of its wave cycles sat in s_waitcnt).
        if (vec_ok && gx >= 0 && gx + 3 < w) {
            float4 g[NG];
#pragma unroll
            for (int p = 0; p < NG; ++p) g[p] = *reinterpret_cast<const float4*>(od.G[p] + rowoff + gx);
#pragma unroll
            for (int p = 0; p < ND; ++p)
                *reinterpret_cast<float4*>(&s_d[p][ly * TW + 4 * v]) =
                    make_float4(g[p + 1].x - g[p].x, g[p + 1].y - g[p].y, g[p + 1].z - g[p].z, g[p + 1].w - g[p].w);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t off = rowoff + min(max(gx + q, 0), w - 1);
                float g[NG];
#pragma unroll
                for (int p = 0; p < NG; ++p) g[p] = od.G[p][off];
#pragma unroll
                for (int p = 0; p < ND; ++p) s_d[p][ly * TW + 4 * v + q] = g[p + 1] - g[p];
            }
        }
    }
    __syncthreads();
    // thread (lx, g) owns column lx and the kERows consecutive rows kERows g .. of the tile.  Per plane: the
    // horizontal 3-max/3-min of the kERows + 2 rows it touches, then the vertical 3-max/3-min per owned pixel = the
    // 3x3 window extrema (centre included).  A pixel is a 26-neighbour maximum iff val >= the max of the three
    // planes' window maxima (val itself is inside its own window, which changes nothing).
    const int lx = tid % kEW, g = tid / kEW;
    const int c = x0 + lx;
    // Candidates are collected per workgroup in LDS and appended with ONE global atomic: at the octaves where the
    // texture lives, tens of thousands of single-address atomics serialised in L2 (the 2 MPix octave took 110 us against
    // 24 us without candidates, longer than the 8 MPix octave).  The order of the cells is irrelevant: they are sorted
    // into the canonical keypoint order later.
    constexpr int kLocalCap = 192;  // (1024 cost a fourth workgroup per CU: 44.5 KB of LDS against 38)
    __shared__ unsigned long long s_cells[kLocalCap];
    __shared__ unsigned int s_n, s_base;
    if (tid == 0) s_n = 0u;
    __syncthreads();
    const bool col_ok = c >= kBorder && c < w - kBorder;
    // The window extrema of three consecutive planes are live at a time (plane p in slot p % 3): layer p - 1 is tested
    // as soon as plane p's are known.
    float wmax[3][kERows], wmin[3][kERows];
#pragma unroll
    for (int p = 0; p < ND; ++p) {
        float hmx[kERows + 2], hmn[kERows + 2];
#pragma unroll
        for (int rr = 0; rr < kERows + 2; ++rr) {
            const float* row = &s_d[p][(kERows * g + rr) * TW + lx + HX];
            hmx[rr] = fmaxf(fmaxf(row[0], row[1]), row[2]);
            hmn[rr] = fminf(fminf(row[0], row[1]), row[2]);
        }
#pragma unroll
        for (int k = 0; k < kERows; ++k) {
            wmax[p % 3][k] = fmaxf(fmaxf(hmx[k], hmx[k + 1]), hmx[k + 2]);
            wmin[p % 3][k] = fminf(fminf(hmn[k], hmn[k + 1]), hmn[k + 2]);
        }
        if (p < 2) continue;
        const int layer = p - 1;
#pragma unroll
        for (int k = 0; k < kERows; ++k) {
            const int ry = kERows * g + k;
            const int r = y0 + ry;
            if (!col_ok || r < kBorder || r >= h - kBorder) continue;
            const float val = s_d[layer][(ry + 1) * TW + lx + HX + 1];
            if (!(fabsf(val) > thr)) continue;
            const float mx = fmaxf(fmaxf(wmax[0][k], wmax[1][k]), wmax[2][k]);
            const float mn = fminf(fminf(wmin[0][k], wmin[1][k]), wmin[2][k]);
            const bool is_max = val > 0 && val >= mx, is_min = val < 0 && val <= mn;
            if (!(is_max || is_min)) continue;
            const unsigned long long cell = ((unsigned long long)o << 40) | ((unsigned long long)layer << 32) |
                                            ((unsigned long long)r << 16) | (unsigned long long)c;
            const unsigned int local = atomicAdd(&s_n, 1u);
            if (local < (unsigned)kLocalCap) {
                s_cells[local] = cell;
            } else {  // more candidates than the local list holds (a degenerate tile): append directly
                const unsigned int slot = atomicAdd(count, 1u);
                if (slot < cap) cells[slot] = cell;
            }
        }
    }
    __syncthreads();
    const unsigned int n_loc = min(s_n, (unsigned)kLocalCap);
    if (n_loc == 0u) return;
    if (tid == 0) s_base = atomicAdd(count, n_loc);
    __syncthreads();
    for (unsigned int e = tid; e < n_loc; e += 256) {
        const unsigned int slot = s_base + e;
        if (slot < cap) cells[slot] = s_cells[e];
    }
}
#undef AT

// ---- sort / dedupe helpers ---------------------------------------------------------------------------
__global__ void rec_keys_kernel(const KpRec* __restrict__ recs, unsigned int n, unsigned long long* __restrict__ keys,
                                unsigned int* __restrict__ idx) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = recs[i].key;
    idx[i] = i;
}
__global__ void unique_flag_kernel(const unsigned long long* __restrict__ keys, unsigned int n,
                                   unsigned int* __restrict__ flag) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}
__global__ void compact_recs_kernel(const KpRec* __restrict__ recs, const unsigned int* __restrict__ idx,
                                    const unsigned int* __restrict__ flag, const unsigned int* __restrict__ pos,
                                    unsigned int n, KpRec* __restrict__ out) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    out[pos[i]] = recs[idx[i]];
}

// ---- orientation: one wave per keypoint ----------------------------------------------------------------
struct PyrTable {
    OctaveDesc oct[16];
    int n_oct, nl;
    float sigma;
};

// ---- the same sweep, marching (default) -----------------------------------------------------------------------------
// scripts/probe/mem_pattern.hip: the seven planes of octave 0 stream at 6.1 TB/s read linearly, at 4.0 TB/s (useful
// bytes) through 128 x 8 tiles with their halo, at 6.2 TB/s through 512-wide strips marched top to bottom - the tiles
// were not slow, they re-read a third of their pixels as halo, and the LDS that holds all DoG planes of a tile kept them
// from growing.  Here a workgroup owns a strip of kMI columns and CH rows of one octave and walks down it four rows at
// a time: the next step's seven float4 per thread are requested before the current step is consumed, the DoG rows live
// in a six-row ring in LDS (the two rows carried over + the four new ones), and a pixel is read ~1.05 times.  Every
// octave of the image is swept by ONE launch (block table in ExtremaPlan).  Same window logic as extrema_kernel, same
// cells (their order is irrelevant: they are sorted into the canonical keypoint order later).
constexpr int kMW = 256;      // columns a workgroup loads per row: 64 float4, the outer four on each side are halo
constexpr int kMI = kMW - 8;  // columns it owns
struct ExtremaPlan {
    int blk_ptr[17];  // first workgroup of octave o
    int nstrip[16];   // strips per row of chunks
    int ch[16];       // rows per chunk (multiple of 4)
};

template <int nl>
__global__ __launch_bounds__(256) void extrema_march_kernel(const PyrTable* __restrict__ pt, ExtremaPlan plan, float thr,
                                                            unsigned long long* __restrict__ cells,
                                                            unsigned int* __restrict__ count, unsigned int cap) {
    constexpr int NG = nl + 3, ND = nl + 2, RING = 6, TW = kMW;
    __shared__ __attribute__((aligned(16))) float s_d[ND][RING * TW];
    constexpr int kLocalCap = 256;
    __shared__ unsigned long long s_cells[kLocalCap];
    __shared__ unsigned int s_n, s_base;
    int o = 0;
    while (o < 15 && (int)blockIdx.x >= plan.blk_ptr[o + 1]) ++o;
    const OctaveDesc& od = pt->oct[o];
    const int w = od.w, h = od.h;
    const int local = (int)blockIdx.x - plan.blk_ptr[o];
    const int strip = local % plan.nstrip[o], chunk = local / plan.nstrip[o];
    const int x0 = strip * kMI - 4, y0 = chunk * plan.ch[o], y1 = min(y0 + plan.ch[o], h);
    const int tid = threadIdx.x;
    if (tid == 0) s_n = 0u;
    const __attribute__((address_space(1))) float* G[NG];
#pragma unroll
    for (int p = 0; p < NG; ++p) G[p] = (const __attribute__((address_space(1))) float*)od.G[p];
    const int lx4 = tid & 63, lr = tid >> 6;
    const int gx = x0 + 4 * lx4;
    const bool vec = (w & 3) == 0 && gx >= 0 && gx + 3 < w;
    auto fetch = [&](int row, float4 g[NG]) __attribute__((always_inline)) {
        const size_t rowoff = (size_t)min(max(row, 0), h - 1) * w;
        if (vec) {
#pragma unroll
            for (int p = 0; p < NG; ++p) {
                typedef float f32x4g __attribute__((ext_vector_type(4)));
                const f32x4g v = *reinterpret_cast<const __attribute__((address_space(1))) f32x4g*>(G[p] + rowoff + gx);
                g[p] = make_float4(v.x, v.y, v.z, v.w);
            }
        } else {
            const size_t o0 = rowoff + min(max(gx, 0), w - 1), o1 = rowoff + min(max(gx + 1, 0), w - 1);
            const size_t o2 = rowoff + min(max(gx + 2, 0), w - 1), o3 = rowoff + min(max(gx + 3, 0), w - 1);
#pragma unroll
            for (int p = 0; p < NG; ++p) g[p] = make_float4(G[p][o0], G[p][o1], G[p][o2], G[p][o3]);
        }
    };
    auto park = [&](int row, const float4 g[NG]) __attribute__((always_inline)) {
        const int slot = (row - (y0 - 1)) % RING;
#pragma unroll
        for (int p = 0; p < ND; ++p)
            *reinterpret_cast<float4*>(&s_d[p][slot * TW + 4 * lx4]) =
                make_float4(g[p + 1].x - g[p].x, g[p + 1].y - g[p].y, g[p + 1].z - g[p].z, g[p + 1].w - g[p].w);
    };
    float4 g[NG];
    if (lr < 2) {  // the two rows above the first step
        fetch(y0 - 1 + lr, g);
        park(y0 - 1 + lr, g);
    }
    fetch(y0 + 1 + lr, g);
    const int lx = tid, c = x0 + lx;
    const int lxc = min(max(lx, 1), kMW - 2);  // (the outermost halo lanes own nothing; keep their reads inside the row)
    const bool col_ok = lx >= 4 && lx < kMW - 4 && c >= kBorder && c < w - kBorder;
    for (int ys = y0; ys < y1; ys += 4) {
        park(ys + 1 + lr, g);                    // rows ys+1 .. ys+4
        if (ys + 4 < y1) fetch(ys + 5 + lr, g);  // the next step's rows: in flight while this step is consumed
        __syncthreads();
        const int base = (ys - y0) % RING;  // ring slot of row ys-1
        float wmax[3][4], wmin[3][4];
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            float hmx[6], hmn[6];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr) {
                int slot = base + rr;
                slot = slot >= RING ? slot - RING : slot;
                const float* row = &s_d[p][slot * TW + lxc - 1];
                hmx[rr] = fmaxf(fmaxf(row[0], row[1]), row[2]);
                hmn[rr] = fminf(fminf(row[0], row[1]), row[2]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wmax[p % 3][k] = fmaxf(fmaxf(hmx[k], hmx[k + 1]), hmx[k + 2]);
                wmin[p % 3][k] = fminf(fminf(hmn[k], hmn[k + 1]), hmn[k + 2]);
            }
            if (p < 2) continue;
            const int layer = p - 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = ys + k;
                if (!col_ok || r >= y1 || r < kBorder || r >= h - kBorder) continue;
                int slot = base + k + 1;
                slot = slot >= RING ? slot - RING : slot;
                const float val = s_d[layer][slot * TW + lx];
                if (!(fabsf(val) > thr)) continue;
                const float mx = fmaxf(fmaxf(wmax[0][k], wmax[1][k]), wmax[2][k]);
                const float mn = fminf(fminf(wmin[0][k], wmin[1][k]), wmin[2][k]);
                const bool is_max = val > 0 && val >= mx, is_min = val < 0 && val <= mn;
                if (!(is_max || is_min)) continue;
                const unsigned long long cell = ((unsigned long long)o << 40) | ((unsigned long long)layer << 32) |
                                                ((unsigned long long)r << 16) | (unsigned long long)c;
                const unsigned int loc = atomicAdd(&s_n, 1u);
                if (loc < (unsigned)kLocalCap) {
                    s_cells[loc] = cell;
                } else {
                    const unsigned int slot2 = atomicAdd(count, 1u);
                    if (slot2 < cap) cells[slot2] = cell;
                }
            }
        }
        __syncthreads();  // the next step parks into the slots this one has read
    }
    const unsigned int n_loc = min(s_n, (unsigned)kLocalCap);
    if (n_loc == 0u) return;
    if (tid == 0) s_base = atomicAdd(count, n_loc);
    __syncthreads();
    for (unsigned int e = tid; e < n_loc; e += 256) {
        const unsigned int slot = s_base + e;
        if (slot < cap) cells[slot] = s_cells[e];
    }
}

// ---- the same sweep, one wave per strip, rows in registers (round 5, the default) ---------------------------------------------
// extrema_march_kernel's counters (profiles/r04d_pmc_extrema.txt): 61 M vector, 54 M scalar and 16 M LDS wave-instructions per
// 4K view, two barriers per four rows, four workgroups per CU - three pipes a third busy each and chained by the barriers; it
// streamed at 2.9 TB/s.  Here nothing is shared between waves: a wave owns a strip of 128 columns (lane l: columns x0 + 2 l and
// x0 + 2 l + 1, the two outer lanes are halo) and walks down its chunk of rows.  Per arriving row a lane subtracts its 2 x (nl + 2)
// DoG values, takes the horizontal 3-max / 3-min with its neighbours' edge values through DPP (wave_shr / wave_shl, no LDS), keeps
// those for the last three rows in registers (slots rotate by a 3x unrolled loop), and tests the centre row.  The loads of the
// row three steps ahead are in flight while a row is consumed.  No LDS traffic but the rare cell records, no barrier.  Same
// window logic, same cells (their order is irrelevant: they are sorted into the canonical keypoint order later).
constexpr int kWS = 124;     // columns a wave owns (64 lanes x 2 - 4)
constexpr int kWCells = 256; // cell records a wave parks in LDS before they go to the global list
struct ExtremaWavePlan {
    int job_ptr[17];  // first wave job of octave o
    int nstrip[16];   // strips per row of chunks
    int ch[16];       // rows per chunk
};

__device__ __forceinline__ float dpp_from_lower_lane(float v) {  // lane l <- lane l - 1 (wave_shr:1); lane 0 keeps its own
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float v) {  // lane l <- lane l + 1 (wave_shl:1); lane 63 keeps its own
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

template <int nl>
__global__ __launch_bounds__(256) void extrema_wave_kernel(const PyrTable* __restrict__ pt, ExtremaWavePlan plan, float thr,
                                                           unsigned long long* __restrict__ cells,
                                                           unsigned int* __restrict__ count, unsigned int cap) {
    constexpr int NG = nl + 3, ND = nl + 2;
    __shared__ unsigned long long s_cells[4][kWCells];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job = (int)blockIdx.x * 4 + wave;
    if (job >= plan.job_ptr[16]) return;  // (whole waves leave: nothing below synchronises across waves)
    int o = 0;
    while (o < 15 && job >= plan.job_ptr[o + 1]) ++o;
    const OctaveDesc& od = pt->oct[o];
    const int w = od.w, h = od.h;
    const int local = job - plan.job_ptr[o];
    const int strip = local % plan.nstrip[o], chunk = local / plan.nstrip[o];
    const int x0 = strip * kWS - 2, y0 = chunk * plan.ch[o], y1 = min(y0 + plan.ch[o], h);
    const int rc0 = max(y0, kBorder), rc1 = min(y1, h - kBorder);  // centre rows [rc0, rc1)
    if (rc0 >= rc1) return;
    const __attribute__((address_space(1))) float* G[NG];
#pragma unroll
    for (int p = 0; p < NG; ++p) G[p] = (const __attribute__((address_space(1))) float*)od.G[p];
    // this lane's column pair; pairs past the image edge are clamped inside the row (their values only reach columns
    // within kBorder of the edge, which are never centres)
    const int c0 = x0 + 2 * lane;
    const int cb = min(max(c0, 0), w - 2);
    const bool own = lane >= 1 && lane <= 62;
    const bool ok0 = own && c0 >= kBorder && c0 < w - kBorder, ok1 = own && c0 + 1 >= kBorder && c0 + 1 < w - kBorder;
    typedef float f32x2g __attribute__((ext_vector_type(2)));
    f32x2g q[3][NG];  // rows in flight, slot = (row - (rc0 - 1)) % 3
    auto fetch = [&](auto S, int row) __attribute__((always_inline)) {
        constexpr int sl = decltype(S)::value;
        const size_t off = (size_t)min(row, h - 1) * w + cb;
#pragma unroll
        for (int p = 0; p < NG; ++p) q[sl][p] = *reinterpret_cast<const __attribute__((address_space(1))) f32x2g*>(G[p] + off);
    };
    float hm[3][ND][2], hn[3][ND][2];  // horizontal 3-max / 3-min of the DoG rows, by row slot
    float dc[3][nl][2];                // the DoG values of the rows themselves, layers 1 .. nl
    // a row's loaded planes -> its slot of hm / hn / dc
    auto reduce_row = [&](auto S) __attribute__((always_inline)) {
        constexpr int sl = decltype(S)::value;
#pragma unroll
        for (int p = 0; p < ND; ++p) {
            const float d0 = q[sl][p + 1].x - q[sl][p].x, d1 = q[sl][p + 1].y - q[sl][p].y;
            const float lf = dpp_from_lower_lane(d1), rt = dpp_from_upper_lane(d0);
            hm[sl][p][0] = fmaxf(fmaxf(lf, d0), d1);
            hm[sl][p][1] = fmaxf(fmaxf(d0, d1), rt);
            hn[sl][p][0] = fminf(fminf(lf, d0), d1);
            hn[sl][p][1] = fminf(fminf(d0, d1), rt);
            if (p >= 1 && p <= nl) {
                dc[sl][p - 1][0] = d0;
                dc[sl][p - 1][1] = d1;
            }
        }
    };
    int qn = 0;  // cell records of this wave so far (wave-uniform)
    auto emit = [&](bool pass, int layer, int r, int c) __attribute__((always_inline)) {
        const unsigned long long m = __ballot(pass);
        if (m == 0ull) return;
        if (pass) {
            const unsigned long long cell = ((unsigned long long)o << 40) | ((unsigned long long)layer << 32) |
                                            ((unsigned long long)r << 16) | (unsigned long long)c;
            const int at = qn + __popcll(m & ((1ull << lane) - 1ull));
            if (at < kWCells) {
                s_cells[wave][at] = cell;
            } else {
                const unsigned int slot2 = atomicAdd(count, 1u);
                if (slot2 < cap) cells[slot2] = cell;
            }
        }
        qn += __popcll(m);
    };
    // centre row r: A = slot of row r - 1, B = slot of row r, C = slot of row r + 1 (just reduced)
    auto test_row = [&](auto SA, auto SB, auto SC, int r) __attribute__((always_inline)) {
        constexpr int a = decltype(SA)::value, b = decltype(SB)::value, c = decltype(SC)::value;
        float wmax[3][2], wmin[3][2];
#pragma unroll
        for (int p = 0; p < ND; ++p) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                wmax[p % 3][k] = fmaxf(fmaxf(hm[a][p][k], hm[b][p][k]), hm[c][p][k]);
                wmin[p % 3][k] = fminf(fminf(hn[a][p][k], hn[b][p][k]), hn[c][p][k]);
            }
            if (p < 2) continue;
            const int layer = p - 1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float val = dc[b][layer - 1][k];
                const float mx = fmaxf(fmaxf(wmax[0][k], wmax[1][k]), wmax[2][k]);
                const float mn = fminf(fminf(wmin[0][k], wmin[1][k]), wmin[2][k]);
                const bool pass = (k ? ok1 : ok0) && fabsf(val) > thr && ((val > 0 && val >= mx) || (val < 0 && val <= mn));
                emit(pass, layer, r, c0 + k);
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // rows rc0 - 1 (slot 0) and rc0 (slot 1) first; from then on the row three ahead is requested as a row is consumed
    fetch(I0{}, rc0 - 1);
    fetch(I1{}, rc0);
    fetch(I2{}, rc0 + 1);
    reduce_row(I0{});
    fetch(I0{}, rc0 + 2);
    reduce_row(I1{});
    fetch(I1{}, rc0 + 3);
    for (int r = rc0; r < rc1; r += 3) {
        // row r + 1 is in slot 2, rows r + 2 / r + 3 are in flight in slots 0 / 1
        reduce_row(I2{});
        fetch(I2{}, r + 4);
        test_row(I0{}, I1{}, I2{}, r);
        if (r + 1 >= rc1) break;
        reduce_row(I0{});
        fetch(I0{}, r + 5);
        test_row(I1{}, I2{}, I0{}, r + 1);
        if (r + 2 >= rc1) break;
        reduce_row(I1{});
        fetch(I1{}, r + 6);
        test_row(I2{}, I0{}, I1{}, r + 2);
    }
    if (qn == 0) return;
    const int n_loc = min(qn, kWCells);
    unsigned int base = 0;
    if (lane == 0) base = atomicAdd(count, (unsigned int)n_loc);
    base = __builtin_amdgcn_readfirstlane(base);
    __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in order: the records above land before these reads)
    for (int e = lane; e < n_loc; e += 64) {
        const unsigned int slot = base + e;
        if (slot < cap) cells[slot] = s_cells[wave][e];
    }
}

// one lane per detected extremum: Newton refinement + contrast/edge tests (dense, no divergence against
// the detection sweep)
__global__ void refine_kernel(const PyrTable* __restrict__ pt, const unsigned long long* __restrict__ cells,
                              const unsigned int* __restrict__ n_cells, unsigned int cells_cap, float contr_thr,
                              float edge_thr, KpRec* __restrict__ recs, unsigned int* __restrict__ count,
                              unsigned int cap) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned int n = min(*n_cells, cells_cap);
    if (i >= n) return;
    const unsigned long long key = cells[i];
    const int o = (int)(key >> 40), layer = (int)((key >> 32) & 0xff), r = (int)((key >> 16) & 0xffff),
              c = (int)(key & 0xffff);
    KpRec kp;
    if (!adjust_extremum(pt->oct[o], pt->nl, o, layer, r, c, contr_thr, edge_thr, kp)) return;
    const unsigned int slot = atomicAdd(count, 1u);
    if (slot < cap) recs[slot] = kp;
}

__device__ __forceinline__ float kp_scale(float sigma, int layer, float xi, int nl) {
    return sigma * my_exp2(((float)layer + xi) / (float)nl);
}

__global__ __launch_bounds__(256) void orient_kernel(const PyrTable* __restrict__ pt,
                                                     const KpRec* __restrict__ kps, unsigned int n,
                                                     unsigned int* __restrict__ ori_count,
                                                     float* __restrict__ ori_angle /* n x 36 */,
                                                     unsigned char* __restrict__ ori_bin /* n x 36 */) {
    __shared__ unsigned long long s_hist[4][kOriBins];
    __shared__ float s_h[4][kOriBins + 4];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned int ki = blockIdx.x * 4 + wv;
    const bool active = ki < n;
    if (lane < kOriBins) s_hist[wv][lane] = 0ull;
    __syncthreads();
    int o = 0, layer = 1, r0 = 0, c0 = 0;
    float xi = 0;
    if (active) {
        const KpRec kp = kps[ki];
        o = (int)(kp.key >> 40);
        layer = (int)((kp.key >> 32) & 0xff);
        r0 = (int)((kp.key >> 16) & 0xffff);
        c0 = (int)(kp.key & 0xffff);
        xi = kp.xi;
    }
    const OctaveDesc& od = pt->oct[o];
    // (global address space stated explicitly: a pointer read from the pyramid table otherwise compiles to flat loads)
    const __attribute__((address_space(1))) float* g = (const __attribute__((address_space(1))) float*)od.G[layer];
    const int w = od.w, h = od.h;
    const float scl = kp_scale(pt->sigma, layer, xi, pt->nl);
    const int radius = (int)rintf(4.5f * scl);
    const float sig = 1.5f * scl;
    const float expf_scale = -1.0f / (2.0f * sig * sig);
    const int side = 2 * radius + 1;
    if (active) {
        for (int s = lane; s < side * side; s += 64) {
            const int i = s / side - radius, j = s % side - radius;
            const int y = r0 + i, x = c0 + j;
            if (y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1) continue;
            const int ctr = y * w + x;
            const float dx = g[ctr + 1] - g[ctr - 1];
            const float dy = g[ctr - w] - g[ctr + w];
            const float wgt = my_exp((float)(i * i + j * j) * expf_scale);
            const float ori = fast_atan2_deg(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy);
            int bin = (int)rintf(((float)kOriBins / 360.0f) * ori);
            if (bin >= kOriBins) bin -= kOriBins;
            if (bin < 0) bin += kOriBins;
            const long long q = to_fix(wgt * mag);
            atomicAdd(&s_hist[wv][bin], (unsigned long long)q);
        }
    }
    __syncthreads();
    if (lane < kOriBins) s_h[wv][lane + 2] = (float)(long long)s_hist[wv][lane] * (1.0f / kFix);
    __syncthreads();
    if (lane < 2) {
        s_h[wv][lane] = s_h[wv][kOriBins + lane];
        s_h[wv][kOriBins + 2 + lane] = s_h[wv][2 + lane];
    }
    __syncthreads();
    float hv = 0.f;
    if (lane < kOriBins)
        hv = (s_h[wv][lane] + s_h[wv][lane + 4]) * (1.0f / 16.0f) + (s_h[wv][lane + 1] + s_h[wv][lane + 3]) * (4.0f / 16.0f) +
             s_h[wv][lane + 2] * (6.0f / 16.0f);
    __syncthreads();
    if (lane < kOriBins) s_h[wv][lane] = hv;
    __syncthreads();
    // peaks (>= 0.8 max, local maxima) in ascending bin order: lane j judges bin j, a ballot prefix orders the survivors
    // (the same tests on the same values as a serial walk over the bins)
    if (active) {
        float omax = lane < kOriBins ? s_h[wv][lane] : -INFINITY;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off));
        const float thr = omax * 0.8f;
        bool peak = false;
        float angle = 0.f;
        if (lane < kOriBins) {
            const int j = lane;
            const int l = j > 0 ? j - 1 : kOriBins - 1, r2 = j < kOriBins - 1 ? j + 1 : 0;
            const float hj = s_h[wv][j], hl = s_h[wv][l], hr = s_h[wv][r2];
            if (hj > hl && hj > hr && hj >= thr) {
                peak = true;
                float bin = (float)j + 0.5f * (hl - hr) / (hl - 2.0f * hj + hr);
                bin = bin < 0 ? (float)kOriBins + bin : (bin >= (float)kOriBins ? bin - (float)kOriBins : bin);
                angle = 360.0f - (360.0f / (float)kOriBins) * bin;
                if (fabsf(angle - 360.0f) < kFltEps) angle = 0.0f;
            }
        }
        const unsigned long long pm = __ballot(peak);
        if (peak) {
            const unsigned int cnt = (unsigned int)__popcll(pm & ((1ull << lane) - 1ull));
            ori_angle[(size_t)ki * kOriBins + cnt] = angle;
            ori_bin[(size_t)ki * kOriBins + cnt] = (unsigned char)lane;
        }
        if (lane == 0) ori_count[ki] = (unsigned int)__popcll(pm);
    }
}

struct OrientedKp {
    unsigned int kp;  // index into the keypoint records
    float angle;
};

__global__ void expand_oriented_kernel(const unsigned int* __restrict__ ori_count,
                                       const unsigned int* __restrict__ ori_pos,
                                       const float* __restrict__ ori_angle, unsigned int n,
                                       OrientedKp* __restrict__ out) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int c = ori_count[i], p = ori_pos[i];
    for (unsigned int e = 0; e < c; ++e) {
        out[p + e].kp = i;
        out[p + e].angle = ori_angle[(size_t)i * kOriBins + e];
    }
}

// ---- descriptor: one wave per oriented keypoint -------------------------------------------------------
constexpr int kD = 4, kN = 8, kHistLen = (kD + 2) * (kD + 2) * (kN + 2);  // 360
constexpr int kDescQueue = 512;  // queued samples per wave between two strided passes over the queue

__global__ __launch_bounds__(256) void descr_kernel(const PyrTable* __restrict__ pt,
                                                    const KpRec* __restrict__ kps,
                                                    const OrientedKp* __restrict__ oks, unsigned int n_out,
                                                    float* __restrict__ desc, int desc_layout, int64_t ldd,
                                                    double* __restrict__ loc, int64_t ldl,
                                                    float* __restrict__ aux, int plain_sweep) {
    __shared__ unsigned long long s_hist[4][kHistLen];
    __shared__ __attribute__((aligned(16))) float s_raw[4][128];
    __shared__ int s_queue[4][kDescQueue];  // per wave: samples that passed the window test, (i << 16) | (j & 0xffff)
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned int oi = blockIdx.x * 4 + wv;
    const bool active = oi < n_out;
    for (int e = lane; e < kHistLen; e += 64) s_hist[wv][e] = 0ull;
    __syncthreads();
    int o = 0, layer = 1, r0 = 0, c0 = 0;
    float xc = 0, xr = 0, xi = 0, contr = 0, kangle = 0;
    if (active) {
        const OrientedKp ok = oks[oi];
        const KpRec kp = kps[ok.kp];
        o = (int)(kp.key >> 40);
        layer = (int)((kp.key >> 32) & 0xff);
        r0 = (int)((kp.key >> 16) & 0xffff);
        c0 = (int)(kp.key & 0xffff);
        xc = kp.xc;
        xr = kp.xr;
        xi = kp.xi;
        contr = kp.contr;
        kangle = ok.angle;
    }
    const OctaveDesc& od = pt->oct[o];
    // (global address space stated explicitly: a pointer read from the pyramid table otherwise compiles to flat loads)
    const __attribute__((address_space(1))) float* g = (const __attribute__((address_space(1))) float*)od.G[layer];
    const int w = od.w, h = od.h;
    const float scl = kp_scale(pt->sigma, layer, xi, pt->nl);
    float ori = 360.0f - kangle;
    if (fabsf(ori - 360.0f) < kFltEps) ori = 0.0f;
    const float ptx = (float)c0 + xc, pty = (float)r0 + xr;
    const int px = (int)rintf(ptx), py = (int)rintf(pty);
    float sin_t, cos_t;
    sincos_deg(ori, sin_t, cos_t);
    const float bins_per_deg = (float)kN / 360.0f;
    const float exp_scale = -1.0f / ((float)(kD * kD) * 0.5f);
    const float hist_width = 3.0f * scl;
    int radius = (int)rintf(hist_width * 1.4142135623730951f * (float)(kD + 1) * 0.5f);
    const int diag = (int)sqrt((double)w * w + (double)h * h);
    if (radius > diag) radius = diag;
    cos_t = cos_t / hist_width;
    sin_t = sin_t / hist_width;
    const int side = 2 * radius + 1;
    (void)side;
    // The samples that contribute lie in a rotated 5 x 5-bin window inside the (2 radius + 1)^2 square - about half of
    // it.  The square is swept row by row with the cheap window test only; passing samples are queued (per wave, in LDS),
    // and the expensive part (atan2, exp, sqrt, trilinear split, eight LDS atomics) runs on full wavefronts of queued
    // samples.  The queue is consumed with a STRIDE: lane l takes entries l * rounds .. + rounds - 1, so that at any
    // moment the lanes of a wave sit ~rounds samples apart along the sweep, i.e. in different histogram cells - 64
    // NEIGHBOURING samples share a handful of bins, and their same-address LDS atomics serialise (the kernel was bound by
    // exactly that: SQ_LDS_BANK_CONFLICT 1.3x the LDS-active cycles).  The histogram is int64 fixed point, so the order
    // of the samples does not matter.
    // (Round 4, measured and dropped - the kernel's vector instructions are not what it waits for: sweeping the square in 8 x 8
    // or 4 x 16 patches culled by their centres - 16 % fewer vector instructions, 282 -> 316-340 us, LDS waits 2.4-3x,
    // whatever the order inside a patch; two histogram copies per wave - 33 KB, four workgroups per CU instead of seven -
    // 325 us.  Requesting the next queue entry before an entry's atomics is the one change that paid, ~3 %.)
    auto accumulate = [&](int i, int j) __attribute__((always_inline)) {
        const float c_rot = (float)j * cos_t - (float)i * sin_t;
        const float r_rot = (float)j * sin_t + (float)i * cos_t;
        float rbin = r_rot + (float)(kD / 2) - 0.5f;
        float cbin = c_rot + (float)(kD / 2) - 0.5f;
        const int r = py + i, c = px + j;
        const int ctr = r * w + c;  // a plane holds < 2^31 floats: 32-bit offsets from the (scalar) plane base
        const float dx = g[ctr + 1] - g[ctr - 1];
        const float dy = g[ctr - w] - g[ctr + w];
        const float wgt = my_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
        const float o_deg = fast_atan2_deg(dy, dx);
        const float mag = sqrtf(dx * dx + dy * dy) * wgt;
        float obin = (o_deg - ori) * bins_per_deg;
        const int rr0 = (int)floorf(rbin), cc0 = (int)floorf(cbin);
        int o0 = (int)floorf(obin);
        rbin -= (float)rr0;
        cbin -= (float)cc0;
        obin -= (float)o0;
        if (o0 < 0) o0 += kN;
        if (o0 >= kN) o0 -= kN;
        const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
        const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
        const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
        const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
        const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
        const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
        const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
        const int idx = ((rr0 + 1) * (kD + 2) + cc0 + 1) * (kN + 2) + o0;
        unsigned long long* hb = &s_hist[wv][idx];
#define ADDQ(off, v) atomicAdd(hb + (off), (unsigned long long)to_fix(v))
        ADDQ(0, v000);
        ADDQ(1, v001);
        ADDQ(kN + 2, v010);
        ADDQ(kN + 3, v011);
        ADDQ((kD + 2) * (kN + 2), v100);
        ADDQ((kD + 2) * (kN + 2) + 1, v101);
        ADDQ((kD + 3) * (kN + 2), v110);
        ADDQ((kD + 3) * (kN + 2) + 1, v111);
#undef ADDQ
    };
    if (active) {
        int* q = s_queue[wv];
        int qn = 0;  // wave-uniform
        auto flush = [&]() __attribute__((always_inline)) {
            const int rounds = (qn + 63) >> 6;
            __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in order: the queue writes land before its reads)
            // (the next entry is requested BEFORE this entry's eight atomics: LDS operations complete in order, so a read
            // issued after them waits for all of them - with same-address conflicts that wait was the round's longest)
            const int e0 = lane * rounds;
            int pk = q[min(e0, kDescQueue - 1)];
            for (int r = 0; r < rounds; ++r) {
                const int e = e0 + r;
                const int pk_next = q[min(e + 1, kDescQueue - 1)];
                if (e < qn) accumulate(pk >> 16, (int)(short)(pk & 0xffff));
                pk = pk_next;
            }
            __builtin_amdgcn_wave_barrier();
            qn = 0;
        };
        // The window test of sample (i, j), exactly as the sweep below evaluates it (the reference's expressions, f32, no
        // contraction): every term is monotone in j for a fixed row i, so the passing samples of a row form ONE interval.
        auto passes = [&](int i, int j) __attribute__((always_inline)) {
            const float c_rot = (float)j * cos_t - (float)i * sin_t;
            const float r_rot = (float)j * sin_t + (float)i * cos_t;
            const float rbin = r_rot + (float)(kD / 2) - 0.5f;
            const float cbin = c_rot + (float)(kD / 2) - 0.5f;
            const int r = py + i, c = px + j;
            return j >= -radius && j <= radius && rbin > -1 && rbin < kD && cbin > -1 && cbin < kD && r > 0 && r < h - 1 && c > 0 && c < w - 1;
        };
        // Round 6: the rows' intervals instead of the whole square.  The square holds (2 radius + 1)^2 points of which about
        // half pass, and testing them 64 to a wave instruction was as many vector instructions as the histogram work itself.
        // Lane l takes row i = -radius + l (+ 64 in a second set: radius <= 63, else the plain sweep): the interval's ends are
        // estimated from the two linear constraints in real arithmetic, widened, and then FOUND with the exact test on the
        // five integers around each estimate; the result is verified with the exact test (the points just outside fail, so by
        // monotonicity nothing further out passes; a row without a passing point must have an empty estimate) - a row that does
        // not verify sends the keypoint through the plain sweep, and so does an orientation within ~0.01 degrees of an axis
        // (|sin| or |cos| below 1e-4 of the scaled units: the estimate's error in j, ~5e-6 / |a|, must stay far below the
        // two-integer search range).  Then every row's samples are queued without any test, 64 to an instruction.  Same set
        // of samples, hence the same histogram bits.
        bool plain = plain_sweep != 0 || radius > 63 || fabsf(sin_t) < 1e-4f || fabsf(cos_t) < 1e-4f;  // (plain_sweep: APS_DESCR_PLAIN=1, A/B)
        int jl_[2] = {0, 0}, len_[2] = {0, 0};
        if (!plain) {
            bool bad = false;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int i = -radius + lane + 64 * half;
                const bool row_ok = i <= radius && py + i > 0 && py + i < h - 1;
                const float fi = (float)i;
                float lo = fmaxf(-(float)radius, (float)(1 - px)), hi = fminf((float)radius, (float)(w - 2 - px));
                auto clip = [&](float a, float lo_v, float hi_v) __attribute__((always_inline)) {  // lo_v < j a < hi_v, |a| >= 1e-4
                    const float p0 = lo_v / a, p1 = hi_v / a;
                    lo = fmaxf(lo, fminf(p0, p1));
                    hi = fminf(hi, fmaxf(p0, p1));
                };
                clip(sin_t, -2.5f - fi * cos_t, 2.5f - fi * cos_t);  // -1 < j sin + i cos + 1.5 < 4
                clip(cos_t, -2.5f + fi * sin_t, 2.5f + fi * sin_t);  // -1 < j cos - i sin + 1.5 < 4
                const int gl = (int)ceilf(lo), gh = (int)floorf(hi);
                int jl = 0x7fffffff, jh = -0x7fffffff;
#pragma unroll
                for (int dlt = 2; dlt >= -2; --dlt)
                    if (passes(i, gl + dlt)) jl = gl + dlt;
#pragma unroll
                for (int dlt = -2; dlt <= 2; ++dlt)
                    if (passes(i, gh + dlt)) jh = gh + dlt;
                const bool none = jl > jh;
                bool good;
                if (none)  // nothing passes within two integers of either estimated end: the estimate itself must be empty
                    good = gl > gh;
                else
                    good = !passes(i, jl - 1) && !passes(i, jh + 1) && passes(i, (jl + jh) / 2) && jl >= gl - 2 && jh <= gh + 2;
                bad = bad || (row_ok && !good);
                jl_[half] = jl;
                len_[half] = (row_ok && !none) ? jh - jl + 1 : 0;
            }
            plain = __any(bad) != 0;
        }
        if (!plain) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned long long rows_m = __ballot(len_[half] > 0);
                while (rows_m) {
                    const int l = __ffsll((long long)rows_m) - 1;
                    rows_m &= rows_m - 1ull;
                    const int rj = __builtin_amdgcn_readlane(jl_[half], l), rn = __builtin_amdgcn_readlane(len_[half], l);
                    const int ri = -radius + l + 64 * half;
                    for (int t0 = 0; t0 < rn; t0 += 64) {
                        const int cnt = min(64, rn - t0);
                        if (lane < cnt) q[qn + lane] = (ri << 16) | ((rj + t0 + lane) & 0xffff);
                        qn += cnt;
                        if (qn > kDescQueue - 64) flush();
                    }
                }
            }
        } else {
            for (int i = -radius; i <= radius; ++i) {
                for (int j0 = -radius; j0 <= radius; j0 += 64) {
                    const int j = j0 + lane;
                    const bool pass = passes(i, j);
                    const unsigned long long m = __ballot(pass);
                    if (m == 0ull) continue;
                    if (pass) q[qn + __popcll(m & ((1ull << lane) - 1ull))] = (i << 16) | (j & 0xffff);
                    qn += __popcll(m);
                    if (qn > kDescQueue - 64) flush();
                }
            }
        }
        if (qn > 0) flush();
    }
    __syncthreads();
    // finalize: 128 outputs, two per lane
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = lane + 64 * half;  // (i*kD + j)*kN + k
        const int k = e & 7, ij = e >> 3, i = ij >> 2, j = ij & 3;
        const int idx = ((i + 1) * (kD + 2) + (j + 1)) * (kN + 2);
        long long v = (long long)s_hist[wv][idx + k];
        if (k < 2) v += (long long)s_hist[wv][idx + kN + k];
        s_raw[wv][e] = (float)v * (1.0f / kFix);
    }
    __syncthreads();
    if (!active) return;
    // the norms are k-ascending fma chains: every lane walks the same chain (LDS broadcast reads).  The clipped and the
    // quantised values are computed once per element (two per lane) and parked in s_raw, so that each chain step is one
    // read and one fma instead of re-deriving the element in every lane.
    // (the chain over s_raw in k order, four elements per LDS read: same fma sequence)
    auto sumsq = [&]() __attribute__((always_inline)) {
        float a = 0;
#pragma unroll 4
        for (int k4 = 0; k4 < 32; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(&s_raw[wv][4 * k4]);
            a = fmaf(v.x, v.x, a);
            a = fmaf(v.y, v.y, a);
            a = fmaf(v.z, v.z, a);
            a = fmaf(v.w, v.w, a);
        }
        return a;
    };
    float nrm2 = sumsq();
    const float thr = sqrtf(nrm2) * 0.2f;
    float clip[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const float v0 = s_raw[wv][lane + 64 * half];
        clip[half] = v0 < thr ? v0 : thr;
    }
    __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations execute in order: all reads above before the writes below)
    s_raw[wv][lane] = clip[0];
    s_raw[wv][lane + 64] = clip[1];
    __builtin_amdgcn_wave_barrier();
    nrm2 = sumsq();
    const float sn = sqrtf(nrm2);
    const float scale = 512.0f / (sn > kFltEps ? sn : kFltEps);
    float qv[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float v = rintf(clip[half] * scale);
        qv[half] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    __builtin_amdgcn_wave_barrier();
    s_raw[wv][lane] = qv[0];
    s_raw[wv][lane + 64] = qv[1];
    __builtin_amdgcn_wave_barrier();
    const float qq = sumsq();
    const float inv = sqrtf(qq);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int e = lane + 64 * half;
        const float outv = inv > 0 ? qv[half] / inv : 0.0f;
        if (desc_layout == APS_ROWMAJOR)
            desc[(size_t)oi * ldd + e] = outv;
        else
            desc[(size_t)e * ldd + oi] = outv;
    }
    if (lane == 0) {
        const float s = ldexpf(1.0f, o) * 0.5f;
        loc[oi] = (double)(ptx * s) + 1.0;
        loc[ldl + oi] = (double)(pty * s) + 1.0;
        if (aux) {
            aux[(size_t)oi * 4 + 0] = scl * s * 2.0f;
            aux[(size_t)oi * 4 + 1] = kangle;
            aux[(size_t)oi * 4 + 2] = contr;
            aux[(size_t)oi * 4 + 3] = (float)(o + 256 * layer);
        }
    }
}

// ---- host side -----------------------------------------------------------------------------------------
static GaussK make_gauss(double sigma) {
    GaussK g;
    int n = (int)std::lrint(sigma * 8.0 + 1.0) | 1;
    if (n > 63) n = 63;
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        g.k[i] = (float)std::exp(s2 * x * x);
        sum += g.k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) g.k[i] = (float)(g.k[i] * sum);
    g.n = n;
    return g;
}

// dec/dh/dw: optional half-resolution copy of the result (the next octave's base); returns false when the radius took the
// generic path, which does not write it.
static bool launch_blur(const float* in, int h, int w, double sigma, float* out, Ws<float>& scratch, float* dec = nullptr,
                        int dh = 0, int dw = 0) {
    const GaussK gk = make_gauss(sigma);
    const int r = gk.n / 2;
    Prof prof("sift_blur");
    const dim3 grid(cdiv(w, kTW), cdiv(h, kTH));
    // The marching form (APS_BLUR_MARCH=1) is an experiment that did NOT pay: bit-identical, but 66 / 75 / 109 us per
    // 33 MPix plane at R = 4 / 5 / 10 against the tile kernel's 60 / 60 / 83 (profiles/r03f_sift_trace.txt).  Unlike the
    // extrema sweep the blur is not a pure stream: at R = 10 its 42 fma per output and ~1.5 GB of LDS traffic per plane
    // already cost what the memory system does, and the march adds three barriers per 16 rows with only 3-6 workgroups
    // per CU to cover them.
    static const bool march = std::getenv("APS_BLUR_MARCH") != nullptr;
    if (march && r >= 1 && r <= 12 && h >= 64 && w >= kSW) {
        const int strips = cdiv(w, kSW);
        int ch = (int)(((long long)h * strips + 1023) / 1024);  // about a thousand workgroups on the large planes
        ch = std::max(4 * kRS, (ch + kRS - 1) / kRS * kRS);
        const dim3 mg(strips, cdiv(h, ch));
        switch (r) {
#define APS_BLUR_CASE(R) \
    case R:              \
        blur_march_kernel<R><<<mg, 256, 0, stream()>>>(in, h, w, gk, out, dec, dh, dw, ch); \
        break;
            APS_BLUR_CASE(1)
            APS_BLUR_CASE(2)
            APS_BLUR_CASE(3)
            APS_BLUR_CASE(4)
            APS_BLUR_CASE(5)
            APS_BLUR_CASE(6)
            APS_BLUR_CASE(7)
            APS_BLUR_CASE(8)
            APS_BLUR_CASE(9)
            APS_BLUR_CASE(10)
            APS_BLUR_CASE(11)
            APS_BLUR_CASE(12)
#undef APS_BLUR_CASE
        }
        check_launch("blur_march_kernel");
        return dec != nullptr;
    }
    // (Measured and dropped, round 5: one wave per 128-column strip marching down its rows with the row-pass results of the last
    // 2R + 2 rows in registers - no barrier, no halo rows recomputed, 0.55 instead of 0.93 wave-instructions per pixel at
    // R = 10, bit-identical - read 59 / 61 / 66 / 94 / 92 / 101 us per octave-0 plane at R = 4 / 5 / 6 / 7 / 8 / 10 against
    // 58 / 62 / 62 / 67 / 63 / 70: a lane's two pixels give two dependent fma chains of 2R + 1 per pass where the tile
    // kernel runs eight side by side on a shared window, and the ring costs 88-170 registers.  profiles/r05j_blur_tile_variants.txt)
    // (Measured and dropped, round 5: 128-column tiles on 512 threads - blur_kernel<R, 128, 512>, fills in rows of 544-608 B
    // instead of 288-352 B - read 58.7 / 70 / 85 us per octave-0 plane at R = 4 / 5 / 10 against 56.8 / 60 / 72:
    // profiles/r05j_blur_tile_variants.txt.  The tile shape stays 64 x 32 on 256 threads.)
    // (Measured and dropped, round 6: 64-ROW tiles on 512 threads for R >= 7 - blur_kernel<R, 64, 512, 64>, vertical halo 1.22-1.38
    // instead of 1.44-1.75 of the tile, one work item per thread in both passes, 30 KB of LDS - same bits, 71.4 / 65.1 us per octave-0
    // plane at R = 10 / 7 against 69.4 / 65.0, the 64-view stage 75.6-76.4 against 74.3-74.9 ms: profiles/r06y_blur_tall_tiles.txt.)
    switch (r) {
#define APS_BLUR_CASE(R) \
    case R:              \
        blur_kernel<R><<<grid, 256, 0, stream()>>>(in, h, w, gk, out, dec, dh, dw); \
        break;
        APS_BLUR_CASE(1)
        APS_BLUR_CASE(2)
        APS_BLUR_CASE(3)
        APS_BLUR_CASE(4)
        APS_BLUR_CASE(5)
        APS_BLUR_CASE(6)
        APS_BLUR_CASE(7)
        APS_BLUR_CASE(8)
        APS_BLUR_CASE(9)
        APS_BLUR_CASE(10)
        APS_BLUR_CASE(11)
        APS_BLUR_CASE(12)
#undef APS_BLUR_CASE
        default: {
            if (scratch.n < (size_t)h * w) scratch.alloc((size_t)h * w);
            blur_row_generic<<<dim3(cdiv(w, 256), h), 256, 0, stream()>>>(in, h, w, gk, scratch);
            blur_col_generic<<<dim3(cdiv(w, 256), h), 256, 0, stream()>>>(scratch, h, w, gk, out);
            dec = nullptr;
        }
    }
    check_launch("blur_kernel");
    return dec != nullptr;
}

// gray + 2x upsample + first blur in one kernel (blur_base_kernel); false when the radius has no tile instantiation (the
// caller then takes gray_up_kernel + launch_blur).  APS_SIFT_NO_BASE_FUSE=1 forces that two-kernel path (same bits).
static bool launch_base_blur(const uint8_t* img, int sh, int sw, int c, int layout, double sigma, float* out, Ws<uint8_t>& gray) {
    static const bool off = [] {
        const char* e = std::getenv("APS_SIFT_NO_BASE_FUSE");
        return e && e[0] == '1';
    }();
    if (off) return false;
    const GaussK gk = make_gauss(sigma);
    const int r = (gk.n - 1) / 2;
    if (r < 3 || r > 8) return false;
    const dim3 grid(cdiv(2 * sw, kTW), cdiv(2 * sh, kTH));
    Prof prof("sift_blur");
    gray.alloc((size_t)sh * sw);
    gray_u8_kernel<<<dim3(cdiv(cdiv(sw, 4), 256), sh), 256, 0, stream()>>>(img, sh, sw, c, layout, gray);
    switch (r) {
#define APS_BLUR_CASE(R) \
    case R:              \
        blur_base_kernel<R><<<grid, 256, 0, stream()>>>(gray, sh, sw, gk, out); \
        break;
        APS_BLUR_CASE(3)
        APS_BLUR_CASE(4)
        APS_BLUR_CASE(5)
        APS_BLUR_CASE(6)
        APS_BLUR_CASE(7)
        APS_BLUR_CASE(8)
#undef APS_BLUR_CASE
        default: return false;
    }
    check_launch("blur_base_kernel");
    return true;
}

static int num_octaves(int H, int W) {
    const int mn = std::min(2 * W, 2 * H);
    return (int)std::lrint(std::log((double)mn) / std::log(2.0) - 2.0) + 1;
}


#ifdef APS_DBG
// Debug builds only (`make debug`): the co-residency experiment of DESIGN.md section 5, victim side.  APS_DBG_REPLAY=R makes
// aps_sift_extract launch the extrema sweep, refine_kernel, orient_kernel and descr_kernel R more times each on the inputs the
// first launch saw and compare the results on the device (order-free checksums for the atomically appended lists, word by
// word for the indexed outputs); the pyramid's checksum is kept per image pointer and compared between calls.
namespace dbg {
enum { kPyrChanged, kExtReplays, kExtDiff, kRefReplays, kRefDiff, kOriReplays, kOriDiff, kOriWords, kDesReplays, kDesDiff, kDesWords, kCalls, kNStat };
static std::atomic<long long> g_stat[kNStat];
static std::mutex g_mu;
static std::map<const void*, unsigned long long> g_pyr;
__global__ void cks_kernel(const unsigned int* __restrict__ p, size_t n_rec, int wpr, int positional, unsigned long long* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long h = 0;
    if (i < n_rec) {
        h = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < wpr; ++k) {
            h ^= p[i * wpr + k];
            h *= 0x100000001B3ull;
            h ^= h >> 29;
        }
        if (positional) h *= (2ull * i + 1ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}
__global__ void diff_kernel(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b, size_t n, unsigned long long* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const unsigned long long m = __ballot(i < n && a[i] != b[i]);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(out, (unsigned long long)__popcll(m));
}
static unsigned long long cks(const void* p, size_t n_rec, int wpr, int positional, unsigned long long* d_slot) {
    APS_HIP(hipMemsetAsync(d_slot, 0, 8, stream()));
    if (n_rec) cks_kernel<<<cdiv(n_rec, 256), 256, 0, stream()>>>((const unsigned int*)p, n_rec, wpr, positional, d_slot);
    unsigned long long h = 0;
    APS_HIP(hipMemcpyAsync(&h, d_slot, 8, hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    return h;
}
static unsigned long long diff(const void* a, const void* b, size_t n_words, unsigned long long* d_slot) {
    APS_HIP(hipMemsetAsync(d_slot, 0, 8, stream()));
    if (n_words) diff_kernel<<<cdiv(n_words, 256), 256, 0, stream()>>>((const unsigned int*)a, (const unsigned int*)b, n_words, d_slot);
    unsigned long long h = 0;
    APS_HIP(hipMemcpyAsync(&h, d_slot, 8, hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));
    return h;
}
}  // namespace dbg
#endif
}  // namespace aps

using namespace aps;

extern "C" {

int aps_sift_extract(const uint8_t* img, int height, int width, int channels, int img_layout,
                     const aps_sift_params* params, float* desc, int desc_layout, int64_t ldd,
                     double* loc, int64_t ldl, float* aux, int64_t cap, int64_t* count) {
    return guarded([&] {
        APS_REQUIRE(img && params && count, APS_E_ARG, "NULL argument");
        APS_REQUIRE(height > 0 && width > 0, APS_E_DIM, "empty image");
        APS_REQUIRE(channels == 1 || channels == 3, APS_E_DIM, "channels must be 1 or 3");
        APS_REQUIRE(img_layout == APS_IMG_U8_HWC || img_layout == APS_IMG_U8_MATLAB, APS_E_TYPE, "unknown image layout");
        APS_REQUIRE(desc_layout == APS_ROWMAJOR || desc_layout == APS_COLMAJOR, APS_E_TYPE, "unknown descriptor layout");
        APS_REQUIRE(params->n_layers >= 1 && params->n_layers <= 5, APS_E_ARG, "NumLayersInOctave must be in 1..5");
        APS_REQUIRE(params->sigma > 0.5, APS_E_ARG, "Sigma must exceed the assumed camera blur 0.5");
        APS_REQUIRE(params->contrast_threshold >= 0 && params->edge_threshold >= 1, APS_E_ARG, "bad thresholds");
        APS_REQUIRE(cap >= 0, APS_E_ARG, "negative capacity");
        APS_REQUIRE(2 * (int64_t)height < 65536 && 2 * (int64_t)width < 65536, APS_E_DIM, "image side must be < 32768");
        ctx();
        *count = 0;
        const int nl = params->n_layers;
        const int H = height, W = width;
        In<uint8_t> dimg(img, (size_t)H * W * channels);
        Ws<float> up, scratch;  // (up: the doubled gray plane, only when the fused base kernel does not apply)
        Ws<uint8_t> gray8;      // the source gray plane as bytes (gray_u8_kernel), read by blur_base_kernel
        const int n_oct = std::min(num_octaves(H, W), 16);
        if (n_oct <= 0) return;
        // pyramid storage
        std::vector<Ws<float>> G((size_t)n_oct * (nl + 3));
        PyrTable table;
        std::memset(&table, 0, sizeof table);
        table.n_oct = n_oct;
        table.nl = nl;
        table.sigma = (float)params->sigma;
        double sig[16];
        sig[0] = params->sigma;
        const double kf = std::pow(2.0, 1.0 / nl);
        for (int i = 1; i < nl + 3; ++i) {
            const double sp = std::pow(kf, (double)(i - 1)) * params->sigma, st = sp * kf;
            sig[i] = std::sqrt(st * st - sp * sp);
        }
        int ow = 2 * W, oh = 2 * H;
        bool base_written = false;  // the previous octave's blur of plane nl wrote this octave's base
        for (int o = 0; o < n_oct; ++o) {
            if (o > 0) {
                ow = std::max(1, ow / 2);
                oh = std::max(1, oh / 2);
            }
            OctaveDesc& od = table.oct[o];
            od.w = ow;
            od.h = oh;
            const size_t px = (size_t)ow * oh;
            for (int i = 0; i < nl + 3; ++i)
                if (!(i == 0 && base_written)) G[o * (nl + 3) + i].alloc(px);
            if (o == 0) {
                double sd = params->sigma * params->sigma - 4.0 * 0.5 * 0.5;
                if (sd < 0.01) sd = 0.01;
                if (!launch_base_blur(dimg, H, W, channels, img_layout, std::sqrt(sd), G[0], gray8)) {
                    up.alloc((size_t)4 * H * W);
                    gray_up_kernel<<<dim3(cdiv(2 * W, kGUW), cdiv(2 * H, kGUH)), 256, 0, stream()>>>(dimg, H, W, channels, img_layout, up);
                    check_launch("gray_up_kernel");
                    launch_blur(up, oh, ow, std::sqrt(sd), G[0], scratch);
                }
            } else if (!base_written) {
                const OctaveDesc& pd = table.oct[o - 1];
                decimate_kernel<<<dim3(cdiv(ow, 256), oh), 256, 0, stream()>>>(G[(o - 1) * (nl + 3) + nl], pd.h, pd.w,
                                                                            oh, ow, G[o * (nl + 3)]);
                check_launch("decimate_kernel");
            }
            base_written = false;
            for (int i = 1; i < nl + 3; ++i) {
                if (i == nl && o + 1 < n_oct) {
                    const int nw = std::max(1, ow / 2), nh = std::max(1, oh / 2);
                    G[(o + 1) * (nl + 3)].alloc((size_t)nw * nh);
                    base_written = launch_blur(G[o * (nl + 3) + i - 1], oh, ow, sig[i], G[o * (nl + 3) + i], scratch,
                                               G[(o + 1) * (nl + 3)], nh, nw);
                } else {
                    launch_blur(G[o * (nl + 3) + i - 1], oh, ow, sig[i], G[o * (nl + 3) + i], scratch);
                }
            }
            for (int i = 0; i < nl + 3; ++i) od.G[i] = G[o * (nl + 3) + i];
        }
#ifdef APS_DBG
        static const int dbg_replay = std::getenv("APS_DBG_REPLAY") ? std::atoi(std::getenv("APS_DBG_REPLAY")) : 0;
        Ws<unsigned long long> dbg_slot(1);
        if (dbg_replay > 0) {
            dbg::g_stat[dbg::kCalls]++;
            unsigned long long h = 0;
            for (int o = 0; o < std::min(n_oct, 3); ++o)
                for (int i = 0; i < nl + 3; ++i)
                    h = h * 1000003ull + dbg::cks(G[o * (nl + 3) + i].get(), (size_t)table.oct[o].w * table.oct[o].h, 1, 1, dbg_slot);
            std::lock_guard<std::mutex> lk(dbg::g_mu);
            auto it = dbg::g_pyr.find((const void*)img);
            if (it == dbg::g_pyr.end())
                dbg::g_pyr[(const void*)img] = h;
            else if (it->second != h)
                dbg::g_stat[dbg::kPyrChanged]++;
        }
#endif
        // extrema: detection sweep per octave -> packed cells; then one dense refinement launch
        Ws<PyrTable> d_table(1);
        APS_HIP(hipMemcpyAsync(d_table, &table, sizeof table, hipMemcpyHostToDevice, stream()));
        const unsigned int cand_cap = (unsigned int)std::min<size_t>(
            params->max_features > 0 ? (size_t)params->max_features : std::max<size_t>((size_t)H * W / 2, 65536), 1u << 26);
        unsigned int cells_cap = (unsigned int)std::min<size_t>(std::max<size_t>((size_t)H * W / 2, 1u << 18), 1u << 27);
        Ws<unsigned long long> cells(cells_cap);
        Ws<KpRec> recs(cand_cap);
        Ws<unsigned int> d_count(2);  // [0] cells, [1] refined records
        const float thr = (float)(int)std::floor(0.5 * params->contrast_threshold / nl * 255.0);
        unsigned int h_counts[2] = {0, 0};
#ifdef APS_SIFT_TIMING  // (timing builds only, results are wrong: APS_SIFT_ABLATE=2 stops after the pyramid, 1 after the extrema sweep)
        static const int sift_ablate = std::getenv("APS_SIFT_ABLATE") ? std::atoi(std::getenv("APS_SIFT_ABLATE")) : 0;
        if (sift_ablate == 2) {
            APS_HIP(hipStreamSynchronize(stream()));
            return;
        }
#endif
        for (int attempt = 0; attempt < 2; ++attempt) {
            APS_HIP(hipMemsetAsync(d_count, 0, 2 * sizeof(unsigned int), stream()));
            if (!std::getenv("APS_EXTREMA_TILES") && !std::getenv("APS_EXTREMA_MARCH")) {
                // one launch over all octaves, one wave per strip and chunk of rows (extrema_wave_kernel)
                ExtremaWavePlan plan;
                std::memset(&plan, 0, sizeof plan);
                static const int ch_env = std::getenv("APS_EXTREMA_CH") ? std::atoi(std::getenv("APS_EXTREMA_CH")) : 0;
                int run = 0;
                for (int o = 0; o < 16; ++o) {
                    plan.job_ptr[o] = run;
                    plan.nstrip[o] = plan.ch[o] = 1;
                    if (o >= n_oct) continue;
                    const OctaveDesc& od = table.oct[o];
                    if (od.w <= 2 * kBorder || od.h <= 2 * kBorder) continue;
                    const int ns = cdiv(od.w, kWS);
                    // rows per chunk (every chunk re-reads two halo rows): ~8000 wave jobs on the largest octave, >= 24 rows
                    int ch = (int)(((long long)od.h * ns + 8191) / 8192);
                    ch = std::max(24, ch);
                    if (ch_env > 0) ch = ch_env;
                    plan.nstrip[o] = ns;
                    plan.ch[o] = ch;
                    run += ns * cdiv(od.h, ch);
                }
                plan.job_ptr[16] = run;
                if (run > 0) {
                    Prof prof("sift_extrema");
                    const int wgs = cdiv(run, 4);
                    auto launch_sweep = [&](unsigned long long* cl, unsigned int* ct) {
                        switch (nl) {
                            case 1: extrema_wave_kernel<1><<<wgs, 256, 0, stream()>>>(d_table, plan, thr, cl, ct, cells_cap); break;
                            case 2: extrema_wave_kernel<2><<<wgs, 256, 0, stream()>>>(d_table, plan, thr, cl, ct, cells_cap); break;
                            case 3: extrema_wave_kernel<3><<<wgs, 256, 0, stream()>>>(d_table, plan, thr, cl, ct, cells_cap); break;
                            case 4: extrema_wave_kernel<4><<<wgs, 256, 0, stream()>>>(d_table, plan, thr, cl, ct, cells_cap); break;
                            default: extrema_wave_kernel<5><<<wgs, 256, 0, stream()>>>(d_table, plan, thr, cl, ct, cells_cap); break;
                        }
                    };
                    launch_sweep(cells, d_count);
                    check_launch("extrema_wave_kernel");
#ifdef APS_DBG
                    if (dbg_replay > 0) {
                        Ws<unsigned long long> cells2(cells_cap);
                        Ws<unsigned int> cnt2(2);
                        unsigned int n1 = 0, n2 = 0;
                        APS_HIP(hipMemcpyAsync(&n1, d_count, 4, hipMemcpyDeviceToHost, stream()));
                        APS_HIP(hipStreamSynchronize(stream()));
                        const unsigned long long c1 = dbg::cks(cells.get(), std::min(n1, cells_cap), 2, 0, dbg_slot);
                        for (int rep = 0; rep < dbg_replay; ++rep) {
                            APS_HIP(hipMemsetAsync(cnt2, 0, 8, stream()));
                            launch_sweep(cells2, cnt2);
                            APS_HIP(hipMemcpyAsync(&n2, cnt2, 4, hipMemcpyDeviceToHost, stream()));
                            APS_HIP(hipStreamSynchronize(stream()));
                            const unsigned long long c2 = dbg::cks(cells2.get(), std::min(n2, cells_cap), 2, 0, dbg_slot);
                            dbg::g_stat[dbg::kExtReplays]++;
                            if (n1 != n2 || c1 != c2) dbg::g_stat[dbg::kExtDiff]++;
                        }
                    }
#endif
                }
            } else if (!std::getenv("APS_EXTREMA_TILES")) {
                // one marching launch over all octaves (extrema_march_kernel; APS_EXTREMA_MARCH=1, rounds 3-4)
                ExtremaPlan plan;
                std::memset(&plan, 0, sizeof plan);
                int run = 0;
                for (int o = 0; o < 16; ++o) {
                    plan.blk_ptr[o] = run;
                    plan.nstrip[o] = plan.ch[o] = 1;
                    if (o >= n_oct) continue;
                    const OctaveDesc& od = table.oct[o];
                    if (od.w <= 2 * kBorder || od.h <= 2 * kBorder) continue;
                    const int ns = cdiv(od.w, kMI);
                    // rows per chunk: about a thousand workgroups for the large octaves, never fewer than 16 rows
                    int ch = (int)(((long long)od.h * ns + 1023) / 1024);
                    ch = std::max(16, (ch + 3) & ~3);
                    plan.nstrip[o] = ns;
                    plan.ch[o] = ch;
                    run += ns * cdiv(od.h, ch);
                }
                plan.blk_ptr[16] = run;
                if (run > 0) {
                    Prof prof("sift_extrema");
                    switch (nl) {
                        case 1: extrema_march_kernel<1><<<run, 256, 0, stream()>>>(d_table, plan, thr, cells, d_count, cells_cap); break;
                        case 2: extrema_march_kernel<2><<<run, 256, 0, stream()>>>(d_table, plan, thr, cells, d_count, cells_cap); break;
                        case 3: extrema_march_kernel<3><<<run, 256, 0, stream()>>>(d_table, plan, thr, cells, d_count, cells_cap); break;
                        case 4: extrema_march_kernel<4><<<run, 256, 0, stream()>>>(d_table, plan, thr, cells, d_count, cells_cap); break;
                        default: extrema_march_kernel<5><<<run, 256, 0, stream()>>>(d_table, plan, thr, cells, d_count, cells_cap); break;
                    }
                    check_launch("extrema_march_kernel");
                }
            } else
            for (int o = 0; o < n_oct; ++o) {
                const OctaveDesc& od = table.oct[o];
                if (od.w <= 2 * kBorder || od.h <= 2 * kBorder) continue;
                Prof prof("sift_extrema");
                const dim3 eg(cdiv(od.w, kEW), cdiv(od.h, kEH));
                switch (nl) {
                    case 1: extrema_kernel<1><<<eg, 256, 0, stream()>>>(od, o, thr, cells, d_count, cells_cap); break;
                    case 2: extrema_kernel<2><<<eg, 256, 0, stream()>>>(od, o, thr, cells, d_count, cells_cap); break;
                    case 3: extrema_kernel<3><<<eg, 256, 0, stream()>>>(od, o, thr, cells, d_count, cells_cap); break;
                    case 4: extrema_kernel<4><<<eg, 256, 0, stream()>>>(od, o, thr, cells, d_count, cells_cap); break;
                    default: extrema_kernel<5><<<eg, 256, 0, stream()>>>(od, o, thr, cells, d_count, cells_cap); break;
                }
                check_launch("extrema_kernel");
            }
            APS_HIP(hipMemcpyAsync(h_counts, d_count, sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));
            if (h_counts[0] <= cells_cap) break;
            // flat or band-limited images can have far more (weak) scale-space extrema than the default room:
            // grow to the exact need and sweep once more
            APS_REQUIRE(attempt == 0, APS_E_INTERNAL, "extrema count changed between identical sweeps");
            cells_cap = h_counts[0];
            cells.alloc(cells_cap);
        }
#ifdef APS_SIFT_TIMING
        if (sift_ablate == 1) return;
#endif
        if (h_counts[0] > 0) {
            Prof prof("sift_refine");
            refine_kernel<<<cdiv(h_counts[0], 256), 256, 0, stream()>>>(d_table, cells, d_count, cells_cap,
                                                                     (float)params->contrast_threshold,
                                                                     (float)params->edge_threshold, recs, d_count.get() + 1,
                                                                     cand_cap);
            check_launch("refine_kernel");
#ifdef APS_DBG
            if (dbg_replay > 0) {
                Ws<KpRec> recs2(cand_cap);
                Ws<unsigned int> cnt2(1);
                unsigned int n1 = 0, n2 = 0;
                APS_HIP(hipMemcpyAsync(&n1, d_count.get() + 1, 4, hipMemcpyDeviceToHost, stream()));
                APS_HIP(hipStreamSynchronize(stream()));
                const unsigned long long c1 = dbg::cks(recs.get(), std::min(n1, cand_cap), 6, 0, dbg_slot);
                for (int rep = 0; rep < dbg_replay; ++rep) {
                    APS_HIP(hipMemsetAsync(cnt2, 0, 4, stream()));
                    refine_kernel<<<cdiv(h_counts[0], 256), 256, 0, stream()>>>(d_table, cells, d_count, cells_cap, (float)params->contrast_threshold,
                                                                             (float)params->edge_threshold, recs2, cnt2, cand_cap);
                    APS_HIP(hipMemcpyAsync(&n2, cnt2, 4, hipMemcpyDeviceToHost, stream()));
                    APS_HIP(hipStreamSynchronize(stream()));
                    const unsigned long long c2 = dbg::cks(recs2.get(), std::min(n2, cand_cap), 6, 0, dbg_slot);
                    dbg::g_stat[dbg::kRefReplays]++;
                    if (n1 != n2 || c1 != c2) dbg::g_stat[dbg::kRefDiff]++;
                }
            }
#endif
        }
        unsigned int n_cand = 0;
        APS_HIP(hipMemcpyAsync(&n_cand, d_count.get() + 1, sizeof n_cand, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        if (n_cand > cand_cap)
            fail(APS_E_CAP, "SIFT found %u extrema, more than the candidate capacity %u (raise params.max_features)", n_cand, cand_cap);
        if (n_cand == 0) return;
        // canonical order + dedupe
        Ws<unsigned long long> keys(n_cand), keys_s(n_cand);
        Ws<unsigned int> idx(n_cand), idx_s(n_cand), flag(n_cand), pos(n_cand);
        rec_keys_kernel<<<cdiv(n_cand, 256), 256, 0, stream()>>>(recs, n_cand, keys, idx);
        size_t tb = 0;
        APS_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.get(), keys_s.get(), idx.get(), idx_s.get(), n_cand, 0, 44, stream()));
        Ws<char> tmp(tb);
        APS_HIP(rocprim::radix_sort_pairs(tmp.get(), tb, keys.get(), keys_s.get(), idx.get(), idx_s.get(), n_cand, 0, 44, stream()));
        unique_flag_kernel<<<cdiv(n_cand, 256), 256, 0, stream()>>>(keys_s, n_cand, flag);
        size_t tb2 = 0;
        APS_HIP(rocprim::exclusive_scan(nullptr, tb2, flag.get(), pos.get(), 0u, n_cand, rocprim::plus<unsigned int>(), stream()));
        Ws<char> tmp2(tb2);
        APS_HIP(rocprim::exclusive_scan(tmp2.get(), tb2, flag.get(), pos.get(), 0u, n_cand, rocprim::plus<unsigned int>(), stream()));
        unsigned int last_pos = 0, last_flag = 0;
        APS_HIP(hipMemcpyAsync(&last_pos, pos.get() + n_cand - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipMemcpyAsync(&last_flag, flag.get() + n_cand - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        const unsigned int n_kp = last_pos + last_flag;
        Ws<KpRec> kps(n_kp);
        compact_recs_kernel<<<cdiv(n_cand, 256), 256, 0, stream()>>>(recs, idx_s, flag, pos, n_cand, kps);
        check_launch("compact_recs_kernel");
        // orientations
        Ws<unsigned int> ocount(n_kp), opos(n_kp);
        Ws<float> oangle((size_t)n_kp * kOriBins);
        Ws<unsigned char> obin((size_t)n_kp * kOriBins);
#ifdef APS_DBG
        if (dbg_replay > 0) {
            APS_HIP(hipMemsetAsync(oangle, 0, (size_t)n_kp * kOriBins * 4, stream()));
            APS_HIP(hipMemsetAsync(obin, 0, (size_t)n_kp * kOriBins, stream()));
        }
#endif
        {
            Prof prof("sift_orient");
            orient_kernel<<<cdiv(n_kp, 4), 256, 0, stream()>>>(d_table, kps, n_kp, ocount, oangle, obin);
        }
        check_launch("orient_kernel");
#ifdef APS_DBG
        if (dbg_replay > 0) {
            Ws<unsigned int> ocount2(n_kp);
            Ws<float> oangle2((size_t)n_kp * kOriBins);
            Ws<unsigned char> obin2(((size_t)n_kp * kOriBins + 3) & ~(size_t)3);
            for (int rep = 0; rep < dbg_replay; ++rep) {
                APS_HIP(hipMemsetAsync(oangle2, 0, (size_t)n_kp * kOriBins * 4, stream()));
                APS_HIP(hipMemsetAsync(obin2, 0, (size_t)n_kp * kOriBins, stream()));
                orient_kernel<<<cdiv(n_kp, 4), 256, 0, stream()>>>(d_table, kps, n_kp, ocount2, oangle2, obin2);
                const unsigned long long d = dbg::diff(ocount.get(), ocount2.get(), n_kp, dbg_slot) +
                                             dbg::diff(oangle.get(), oangle2.get(), (size_t)n_kp * kOriBins, dbg_slot) +
                                             dbg::diff(obin.get(), obin2.get(), (size_t)n_kp * kOriBins / 4, dbg_slot);
                dbg::g_stat[dbg::kOriReplays]++;
                if (d) dbg::g_stat[dbg::kOriDiff]++;
                dbg::g_stat[dbg::kOriWords] += (long long)d;
            }
        }
#endif
        size_t tb3 = 0;
        APS_HIP(rocprim::exclusive_scan(nullptr, tb3, ocount.get(), opos.get(), 0u, n_kp, rocprim::plus<unsigned int>(), stream()));
        Ws<char> tmp3(tb3);
        APS_HIP(rocprim::exclusive_scan(tmp3.get(), tb3, ocount.get(), opos.get(), 0u, n_kp, rocprim::plus<unsigned int>(), stream()));
        unsigned int lp = 0, lc = 0;
        APS_HIP(hipMemcpyAsync(&lp, opos.get() + n_kp - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipMemcpyAsync(&lc, ocount.get() + n_kp - 1, sizeof(unsigned int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        const unsigned int n_out = lp + lc;
        *count = n_out;
        if ((int64_t)n_out > cap) fail(APS_E_CAP, "feature capacity %lld < %u features", (long long)cap, n_out);
        if (n_out == 0) return;
        APS_REQUIRE(desc && loc, APS_E_ARG, "NULL output with features present");
        if (desc_layout == APS_ROWMAJOR)
            APS_REQUIRE(ldd >= 128, APS_E_DIM, "ldd < 128");
        else
            APS_REQUIRE(ldd >= n_out, APS_E_DIM, "ldd < count");
        APS_REQUIRE(ldl >= n_out, APS_E_DIM, "ldl < count");
        Ws<OrientedKp> oks(n_out);
        expand_oriented_kernel<<<cdiv(n_kp, 256), 256, 0, stream()>>>(ocount, opos, oangle, n_kp, oks);
        check_launch("expand_oriented_kernel");
        const size_t desc_elems = desc_layout == APS_ROWMAJOR ? (size_t)(n_out - 1) * ldd + 128 : (size_t)127 * ldd + n_out;
        Out<float> odesc(desc, desc_elems), oaux(aux, (size_t)n_out * 4);
        Out<double> oloc(loc, (size_t)ldl + n_out);
        {
            Prof prof("sift_descr");
            static const int plain_sweep = std::getenv("APS_DESCR_PLAIN") ? 1 : 0;  // (A/B: the whole-square sweep of rounds 1-5; same bits)
            descr_kernel<<<cdiv(n_out, 4), 256, 0, stream()>>>(d_table, kps, oks, n_out, odesc, desc_layout, ldd, oloc, ldl,
                                                               oaux.present() ? oaux.get() : nullptr, plain_sweep);
        }
        check_launch("descr_kernel");
#ifdef APS_DBG
        if (dbg_replay > 0 && desc_layout == APS_ROWMAJOR && ldd == 128) {
            Ws<float> desc2((size_t)n_out * 128), aux2((size_t)n_out * 4);
            Ws<double> loc2((size_t)2 * n_out);
            for (int rep = 0; rep < dbg_replay; ++rep) {
                descr_kernel<<<cdiv(n_out, 4), 256, 0, stream()>>>(d_table, kps, oks, n_out, desc2, desc_layout, 128, loc2, n_out,
                                                                   oaux.present() ? aux2.get() : nullptr, 0);
                unsigned long long d = dbg::diff(odesc.get(), desc2.get(), (size_t)n_out * 128, dbg_slot) +
                                       dbg::diff(oloc.get(), loc2.get(), (size_t)2 * n_out, dbg_slot) +
                                       dbg::diff(oloc.get() + ldl, loc2.get() + n_out, (size_t)2 * n_out, dbg_slot);
                if (oaux.present()) d += dbg::diff(oaux.get(), aux2.get(), (size_t)n_out * 4, dbg_slot);
                dbg::g_stat[dbg::kDesReplays]++;
                if (d) dbg::g_stat[dbg::kDesDiff]++;
                dbg::g_stat[dbg::kDesWords] += (long long)d;
            }
        }
#endif
        odesc.commit();
        oloc.commit();
        oaux.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

}  // extern "C"

#ifdef APS_DBG
// (debug library only) the counters of the replay experiment as text; reset = 1 clears them and the pyramid checksums
extern "C" int aps_dbg_replay_report(char* buf, int cap, int reset) {
    using namespace aps::dbg;
    static const char* names[kNStat] = {"pyramid_changed", "extrema_replays", "extrema_diff", "refine_replays", "refine_diff", "orient_replays",
                                        "orient_diff", "orient_words", "descr_replays", "descr_diff", "descr_words", "calls"};
    int n = 0;
    for (int i = 0; i < kNStat && n < cap; ++i) n += std::snprintf(buf + n, (size_t)(cap - n), "%s=%lld ", names[i], (long long)g_stat[i].load());
    if (reset) {
        for (int i = 0; i < kNStat; ++i) g_stat[i] = 0;
        if (reset > 1) {
            std::lock_guard<std::mutex> lk(g_mu);
            g_pyr.clear();
        }
    }
    return 0;
}
#endif
