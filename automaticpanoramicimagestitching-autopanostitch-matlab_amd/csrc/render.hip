// render.hip — inverse-warp resampling, tile fusion and Laplacian-pyramid multiband blending on gfx950.
//
// Restates PP/renderPanorama/renderPanorama.m:342-425 (tile loop, rays, paint), :825-1060 (fuseTile),
// :1063-1146 (sampleOneTile), :1282-1312 (warpWeights), :1393-1456 (sampleBlock),
// PP/blending/multiBandBlending.m:45-171, PP/blending/linearBlending.m:64-101 and
// PP/imageProcessing/imageWarp.m:39-168.
//
// HBM layout
//   source images : RGBA8, row-major, one 32-bit word per pixel (a bilinear tap is one dword gather);
//                   converted once from the caller's layout (HWC or MATLAB planar column-major).
//   tile layers   : float4 per pixel (r, g, b, w): colour and blend weight travel together because the
//                   reference blurs/downsamples both identically at every pyramid level.
//   numerator pyr : float4 per pixel per level (w unused).
// All tap sums are k-ascending f32 fma chains (the oracle's order); toolbox semantics (interp2,
// imgaussfilt, imresize) are the ones written down in oracle/render_oracle.c.
#include <algorithm>
#include <climits>
#include <cmath>
#include <memory>
#include <vector>

#include <functional>

#include "render_dev.h"

namespace aps {


// ------------------------------------------------------------------------------------------------
// coverage prepass: which images touch which tile (exact: the same predicate as the sampler)
// ------------------------------------------------------------------------------------------------
// Coverage pre-pass of one tile.  Each 32 x 8-pixel block records, per image, whether any of its pixels maps
// into that image: rowmask[img][block row] |= bit(block column >> xshift).  The few-way contended 64-bit ORs
// replace a per-image bounding-box min/max that every wave would fight over; footprint_kernel turns the masks
// into boxes.  The footprint therefore has block granularity (a superset of the mask, which is all it must be).
__global__ __launch_bounds__(256) void cover_kernel(DevCanvas cv, const DevImage* __restrict__ imgs,
                                                     int n_img, int r0, int c0, int ht, int wt,
                                                     float angle_pow, int nby, int xshift, int block_cull,
                                                     unsigned long long* __restrict__ rowmask) {
    __shared__ unsigned long long s_any, s_cand;
    __shared__ float s_dc[3];
    __shared__ int s_cosb;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const bool in_tile = x < wt && y < ht;
    float d[3] = {0.f, 0.f, 1.f};
    if (in_tile) canvas_ray(cv, (float)(c0 + x), (float)(r0 + y), d);
    const float dn = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
    // Block-level cull: the ray of the block's centre pixel and the widest angle theta_b between it and any ray of
    // the block; image i can be seen from the block only if angle(centre, axis_i) <= theta_i + theta_b.  The
    // per-pixel loop then runs over those candidates only (typically 2-6 of 64 images).
    if (threadIdx.x == 0) {
        float dc[3];
        canvas_ray(cv, (float)(c0 + min(blockIdx.x * 32 + 16, (unsigned)wt - 1)), (float)(r0 + min(blockIdx.y * 8 + 4, (unsigned)ht - 1)), dc);
        s_dc[0] = dc[0];
        s_dc[1] = dc[1];
        s_dc[2] = dc[2];
        s_cosb = __float_as_int(1.0f);
    }
    __syncthreads();
    {
        float cb = in_tile ? fmaf(d[2], s_dc[2], fmaf(d[1], s_dc[1], d[0] * s_dc[0])) / fmaxf(dn, 1e-8f) : 1.0f;
        cb = fminf(fmaxf(cb, 0.0f), 1.0f);  // non-negative floats order like their bit patterns
        for (int off = 32; off > 0; off >>= 1) cb = fminf(cb, __shfl_xor(cb, off));
        if ((threadIdx.x & 63) == 0) atomicMin(&s_cosb, __float_as_int(cb));
    }
    __syncthreads();
    const float cosb = fmaxf(__int_as_float(s_cosb) - 1e-6f, 0.0f), sinb = sqrtf(fmaxf(0.0f, 1.0f - cosb * cosb));
    const unsigned long long colbit = 1ull << (blockIdx.x >> xshift);
    for (int base = 0; base < n_img; base += 64) {
        const int cnt = min(64, n_img - base);
        if (threadIdx.x < 64) {
            bool cand = false;
            if (threadIdx.x < cnt) {
                const DevImage& im = imgs[base + threadIdx.x];
                const float ca = fmaf(s_dc[2], im.R[8], fmaf(s_dc[1], im.R[5], s_dc[0] * im.R[2]));
                const float ci = im.cmin, si = sqrtf(fmaxf(0.0f, 1.0f - ci * ci));
                // cos(theta_i + theta_b), valid while the sum stays below 90 degrees; otherwise keep the image
                cand = !block_cull || ci <= 0.0f || cosb <= 0.0f || ci * cosb - si * sinb <= 0.0f ||
                       ca >= (ci * cosb - si * sinb) - 1e-5f;
            }
            const unsigned long long b = __ballot(cand);
            if (threadIdx.x == 0) {
                s_cand = b;
                s_any = 0ull;
            }
        }
        __syncthreads();
        unsigned long long mine = 0ull;  // wave-uniform: images (of this group of 64) seen by this wave
        for (unsigned long long todo = s_cand; todo; todo &= todo - 1) {
            const int i = __ffsll((long long)todo) - 1;
            const DevImage& im = imgs[base + i];
            // necessary condition first (three fmas): the ray must lie inside the image's cone about its optical
            // axis; a wave that is wholly outside skips the projection.  cam2 is project()'s own expression.
            const float cam2 = fmaf(d[2], im.R[8], fmaf(d[1], im.R[5], d[0] * im.R[2]));
            if (!__any(in_tile && cam2 >= im.cmin * dn)) continue;
            float u, v, wa;
            const bool m = in_tile && project(im, d, angle_pow, u, v, wa);
            if (__any(m)) mine |= 1ull << i;
        }
        if ((threadIdx.x & 63) == 0 && mine) atomicOr(&s_any, mine);
        __syncthreads();
        const unsigned long long all = s_any;
        if (threadIdx.x < cnt && ((all >> threadIdx.x) & 1ull))
            atomicOr(&rowmask[(size_t)(base + threadIdx.x) * nby + blockIdx.y], colbit);
        __syncthreads();
    }
}

// rowmask -> {x0, y0, x1, y1} per (tile, image), pixels of the tile, half-open; x1 <= x0 when nothing is covered
__global__ void footprint_kernel(const unsigned long long* __restrict__ rowmask, const int* __restrict__ tile_dims,
                                 int n_img, int nby_max, int total, int* __restrict__ bbox) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // tile * n_img + image
    if (e >= total) return;
    const int t = e / n_img;
    const int ht = tile_dims[4 * t + 0], wt = tile_dims[4 * t + 1], nby = tile_dims[4 * t + 2], xshift = tile_dims[4 * t + 3];
    const unsigned long long* m = rowmask + (size_t)e * nby_max;
    unsigned long long cols = 0ull;
    int y0 = INT_MAX, y1 = 0;
    for (int by = 0; by < nby; ++by) {
        const unsigned long long r = m[by];
        if (r) {
            cols |= r;
            y0 = min(y0, by * 8);
            y1 = max(y1, min(by * 8 + 8, ht));
        }
    }
    int x0 = 0, x1 = 0;
    if (cols) {
        const int lo = __ffsll((long long)cols) - 1, hi = 63 - __clzll((long long)cols);
        x0 = (lo << xshift) * 32;
        x1 = min(((hi + 1) << xshift) * 32, wt);
    } else {
        y0 = 0;
    }
    bbox[4 * e + 0] = x0;
    bbox[4 * e + 1] = y0;
    bbox[4 * e + 2] = x1;
    bbox[4 * e + 3] = y1;
}

// ------------------------------------------------------------------------------------------------
// warp: one layer (tile x image) as float4 (r,g,b,w = Wang*Wf)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void warp_layer_kernel(DevCanvas cv, const DevImage* __restrict__ imgs,
                                                          int img, int r0, int c0, int wt, Rect rc,
                                                          float angle_pow, float wf_floor,
                                                          float4* __restrict__ layer) {
    const int x = rc.x0 + blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = rc.y0 + blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= rc.x1 || y >= rc.y1) return;
    float d[3];
    canvas_ray(cv, (float)(c0 + x), (float)(r0 + y), d);
    const Sample s = sample_one(imgs[img], d, angle_pow);
    float wf = s.wf;
    if (wf_floor > 0.f) {  // 'linear': Wf = max(Wf, 1e-4) (:933)
        if (!isfinite(wf)) wf = 0.f;
        wf = wf > wf_floor ? wf : wf_floor;
    }
    const float wv = s.m ? s.wang * wf : 0.0f;
    layer[(size_t)y * wt + x] = make_float4(s.s[0], s.s[1], s.s[2], wv);
}

// test/diagnostic form: the four outputs of sampleOneTile separately
__global__ __launch_bounds__(256) void warp_tile_kernel(DevCanvas cv, const DevImage* __restrict__ imgs,
                                                         int r0, int c0, int ht, int wt,
                                                         float angle_pow, float* __restrict__ S,
                                                         uint8_t* __restrict__ M,
                                                         float* __restrict__ Wang,
                                                         float* __restrict__ Wf) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= wt || y >= ht) return;
    float d[3];
    canvas_ray(cv, (float)(c0 + x), (float)(r0 + y), d);
    const Sample s = sample_one(imgs[0], d, angle_pow);
    const size_t o = (size_t)y * wt + x;
    S[3 * o + 0] = s.s[0];
    S[3 * o + 1] = s.s[1];
    S[3 * o + 2] = s.s[2];
    M[o] = s.m ? 1 : 0;
    Wang[o] = s.wang;
    Wf[o] = s.wf;
}

// ------------------------------------------------------------------------------------------------
// weight normalisation over the K layers of a tile
// ------------------------------------------------------------------------------------------------
// Up to kMaxK layer pointers travel as a kernel argument (no table upload, no host sync); tiles with more
// contributors take the uploaded-table path.
struct PtrTab {
    float4* p[kMaxK];
};


// fuse_norm: renderPanorama.m:1009-1017 (w *= 1/sum where sum > 1e-8), coverage = any(w > 0)
// mbb_norm : multiBandBlending.m:72-85   (w = max(0,w)/sum where sum > 1e-8)
template <class Tab>
__device__ __forceinline__ float4* tab_get(const Tab& t, int k);
template <>
__device__ __forceinline__ float4* tab_get<PtrTab>(const PtrTab& t, int k) { return t.p[k]; }
template <>
__device__ __forceinline__ float4* tab_get<float4* const*>(float4* const* const& t, int k) { return t[k]; }

// use_rects == 0: every layer covers the whole tile (also the only form for K > kMaxK)
template <class Tab>
__global__ void norm_weights_kernel(Tab layers_tab, RectTab rects, int use_rects, int K, int w, size_t n,
                                    int fuse_norm, int mbb_norm, uint8_t* __restrict__ cov) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int y = (int)(p / (size_t)w), x = (int)(p - (size_t)y * w);
    // bit k: the pixel lies in layer k's footprint (K <= 64 per call: callers chunk above that)
    unsigned long long inside = 0ull;
    for (int k = 0; k < K; ++k)
        if (!use_rects || in_rect(rects.r[k], x, y)) inside |= 1ull << k;
    bool any = false;
    if (fuse_norm) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float wv = (inside >> k) & 1 ? tab_get<Tab>(layers_tab, k)[p].w : 0.f;
            s = s + wv;
            any |= wv > 0.f;
        }
        const float inv = s > 1e-8f ? 1.0f / s : 0.f;
        for (int k = 0; k < K; ++k)
            if ((inside >> k) & 1) tab_get<Tab>(layers_tab, k)[p].w = tab_get<Tab>(layers_tab, k)[p].w * inv;
    }
    if (mbb_norm) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float wv = (inside >> k) & 1 ? tab_get<Tab>(layers_tab, k)[p].w : 0.f;
            s = s + (wv > 0.f ? wv : 0.f);
        }
        for (int k = 0; k < K; ++k) {
            if (!((inside >> k) & 1)) continue;
            const float wv = tab_get<Tab>(layers_tab, k)[p].w > 0.f ? tab_get<Tab>(layers_tab, k)[p].w : 0.f;
            tab_get<Tab>(layers_tab, k)[p].w = s > 1e-8f ? wv / s : 0.f;
        }
    }
    if (cov) cov[p] = any ? 1 : 0;
}


__global__ void blur_v_kernel(const float4* __restrict__ in, int h, int w, Taps tp,
                              float4* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t <= 2 * tp.r; ++t) {
        const int yy = min(max(y + t - tp.r, 0), h - 1);
        a = fma4(tp.k[t], in[(size_t)yy * w + x], a);
    }
    out[(size_t)y * w + x] = a;
}

__global__ void blur_h_kernel(const float4* __restrict__ in, int h, int w, Taps tp,
                              float4* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t <= 2 * tp.r; ++t) {
        const int xx = min(max(x + t - tp.r, 0), w - 1);
        a = fma4(tp.k[t], in[(size_t)y * w + xx], a);
    }
    out[(size_t)y * w + x] = a;
}


// dim 0 (rows): in h x w -> out oh x w
__global__ void resize_rows_kernel(const float4* __restrict__ in, int h, int w, int oh,
                                   float4* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w) return;
    int left;
    float wts[12];
    const int P = resize_taps(h, oh, y, left, wts);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < P; ++t) {
        const int yy = min(max(left + t, 1), h) - 1;
        a = fma4(wts[t], in[(size_t)yy * w + x], a);
    }
    out[(size_t)y * w + x] = a;
}

// dim 1 (cols): in h x w -> out h x ow
__global__ void resize_cols_kernel(const float4* __restrict__ in, int h, int w, int ow,
                                   float4* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= ow) return;
    int left;
    float wts[12];
    const int P = resize_taps(w, ow, x, left, wts);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < P; ++t) {
        const int xx = min(max(left + t, 1), w) - 1;
        a = fma4(wts[t], in[(size_t)y * w + xx], a);
    }
    out[(size_t)y * ow + x] = a;
}

// Num_l += (G - U) .* G.w   (multiBandBlending.m:139-144)
__global__ void lap_accum_kernel(const float4* __restrict__ G, const float4* __restrict__ U, size_t n,
                                 float4* __restrict__ Num) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 g = G[p], u = U[p];
    float4 a = Num[p];
    a.x = a.x + (g.x - u.x) * g.w;
    a.y = a.y + (g.y - u.y) * g.w;
    a.z = a.z + (g.z - u.z) * g.w;
    Num[p] = a;
}

// Num_L += G .* G.w   (:159)
__global__ void coarse_accum_kernel(const float4* __restrict__ G, size_t n, float4* __restrict__ Num) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 g = G[p];
    float4 a = Num[p];
    a.x = a.x + g.x * g.w;
    a.y = a.y + g.y * g.w;
    a.z = a.z + g.z * g.w;
    Num[p] = a;
}

// F = up + Num_l   (:166)
__global__ void add_kernel(const float4* __restrict__ A, const float4* __restrict__ B, size_t n,
                           float4* __restrict__ out) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float4 a = A[p], b = B[p];
    out[p] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, 0.f);
}

__global__ void pack_layer_kernel(const float* __restrict__ C3, const float* __restrict__ W, size_t n,
                                  float4* __restrict__ out) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    out[p] = make_float4(C3[3 * p], C3[3 * p + 1], C3[3 * p + 2], W[p]);
}

__global__ void unpack_clamp_kernel(const float4* __restrict__ in, size_t n, int clamp01,
                                    float* __restrict__ out) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    float v[3] = {in[p].x, in[p].y, in[p].z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = v[c];
        if (clamp01) {
            t = t > 0.f ? t : 0.f;  // max(0,F): NaN -> 0
            t = t < 1.f ? t : 1.f;
        }
        out[3 * p + c] = t;
    }
}

// 'linear' (:916-978) over K layers that already hold w = Wang*max(Wf,1e-4) (0 outside the mask)
template <class Tab>
__global__ void linear_fuse_kernel(Tab layers_tab, RectTab rects, int use_rects, int K, int w, size_t n,
                                   float4* __restrict__ F, uint8_t* __restrict__ cov) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int y = (int)(p / (size_t)w), x = (int)(p - (size_t)y * w);
    float acc[3] = {0.f, 0.f, 0.f}, ws = 0.f, bestw = 0.f, best[3] = {0.f, 0.f, 0.f};
    bool anyv = false;
    for (int k = 0; k < K; ++k) {
        if (use_rects && !in_rect(rects.r[k], x, y)) continue;  // an all-zero term
        const float4 g = tab_get<Tab>(layers_tab, k)[p];
        acc[0] = acc[0] + g.x * g.w;
        acc[1] = acc[1] + g.y * g.w;
        acc[2] = acc[2] + g.z * g.w;
        ws = ws + g.w;
        // inside the mask Wang > 0 and Wf >= 1e-4, so w > 0 <=> M (the reference tests M)
        const bool m = g.w > 0.f;
        anyv |= m;
        if (m && g.w > bestw) {
            bestw = g.w;
            best[0] = g.x;
            best[1] = g.y;
            best[2] = g.z;
        }
    }
    const bool z = ws > 1e-12f;
    float4 o;
    o.x = z ? acc[0] / ws : (anyv ? best[0] : 0.f);
    o.y = z ? acc[1] / ws : (anyv ? best[1] : 0.f);
    o.z = z ? acc[2] / ws : (anyv ? best[2] : 0.f);
    o.w = 0.f;
    F[p] = o;
    cov[p] = ws > 0.f ? 1 : 0;
}

// linearBlending.m:64-101 (sum(I.*W) / max(sum(W), eps('single')))
template <class Tab>
__global__ void linear_blend_kernel(Tab layers_tab, int K, size_t n,
                                    float4* __restrict__ F) {
    const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (p >= n) return;
    float acc[3] = {0.f, 0.f, 0.f}, den = 0.f;
    for (int k = 0; k < K; ++k) {
        const float4 g = tab_get<Tab>(layers_tab, k)[p];
        acc[0] = acc[0] + g.x * g.w;
        acc[1] = acc[1] + g.y * g.w;
        acc[2] = acc[2] + g.z * g.w;
        den = den + g.w;
    }
    const float tiny = 1.1920928955078125e-07f;
    const float d = den > tiny ? den : tiny;
    F[p] = make_float4(acc[0] / d, acc[1] / d, acc[2] / d, 0.f);
}

// 'none' policies (:864-914): images visited in order, one launch per image
__global__ __launch_bounds__(256) void none_fuse_kernel(DevCanvas cv, const DevImage* __restrict__ imgs,
                                                         int img, int r0, int c0, int ht, int wt,
                                                         float angle_pow, int policy,
                                                         float4* __restrict__ F,
                                                         uint8_t* __restrict__ cov) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= wt || y >= ht) return;
    float d[3];
    canvas_ray(cv, (float)(c0 + x), (float)(r0 + y), d);
    const Sample s = sample_one(imgs[img], d, angle_pow);
    const size_t p = (size_t)y * wt + x;
    float4 f = F[p];  // f.w carries bestW for 'maxangle'
    bool upd;
    if (policy == APS_NONE_LAST)
        upd = s.m;
    else if (policy == APS_NONE_FIRST)
        upd = s.m && !cov[p];
    else
        upd = s.m && (s.wang > f.w);
    if (upd) {
        f.x = s.s[0];
        f.y = s.s[1];
        f.z = s.s[2];
        if (policy == APS_NONE_MAXANGLE) f.w = s.wang;
        F[p] = f;
        cov[p] = 1;
    }
}

// paint + uint8 (:408-425): pano = covered ? uint8(round(255*clamp(F))) : canvas colour
__global__ void paint_kernel(const float4* __restrict__ F, const uint8_t* __restrict__ cov, int r0,
                             int c0, int ht, int wt, int H, int W, int white, int out_layout,
                             uint8_t* __restrict__ pano, uint8_t* __restrict__ covered) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= wt) return;
    const size_t p = (size_t)y * wt + x;
    const float4 f = F[p];
    const bool c = cov[p] != 0;
    const float v[3] = {f.x, f.y, f.z};
    const int gy = r0 + y, gx = c0 + x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = roundf(255.0f * v[k]);  // MATLAB round: half away from zero
        if (!(t > 0.f)) t = 0.f;
        if (t > 255.f) t = 255.f;
        const uint8_t b = c ? (uint8_t)t : (white ? 255 : 0);
        if (out_layout == APS_IMG_U8_HWC)
            pano[((size_t)gy * W + gx) * 3 + k] = b;
        else
            pano[(size_t)k * H * W + (size_t)gx * H + gy] = b;
    }
    if (covered) {
        if (out_layout == APS_IMG_U8_HWC)
            covered[(size_t)gy * W + gx] = c ? 1 : 0;
        else
            covered[(size_t)gx * H + gy] = c ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// imageWarp 'bilinear' (imageWarp.m:125-168)
// ------------------------------------------------------------------------------------------------
struct HWarp {
    double A[9];  // adjugate of H/H(3,3)
    double det;
};

// bicubicKernel of imageWarp.m:275-301 (Keys, a = -0.5), |x|^2 and |x|^3 as products, terms left to right
__device__ __forceinline__ double warp_cubic(double x) {
    const double a = fabs(x), a2 = a * a, a3 = a2 * a;
    if (a <= 1.0) return (1.5 * a3 - 2.5 * a2) + 1.0;
    if (a <= 2.0) return ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0;
    return 0.0;
}

// METHOD: APS_WARP_NEAREST (imageWarp.m:109-123), APS_WARP_BILINEAR (:125-168), APS_WARP_BICUBIC (:170-264)
template <class T, int METHOD>
__global__ void image_warp_h_kernel(const T* __restrict__ in, int in_h, int in_w, int C, HWarp hw,
                                    int out_h, int out_w, double x0, double y0, double sx, double sy,
                                    T fill, T* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= out_w) return;
    const double X = x0 + (double)x * sx, Y = y0 + (double)y * sy;
    const double s0 = ((hw.A[0] * X + hw.A[3] * Y) + hw.A[6]) / hw.det;
    const double s1 = ((hw.A[1] * X + hw.A[4] * Y) + hw.A[7]) / hw.det;
    const double s2 = ((hw.A[2] * X + hw.A[5] * Y) + hw.A[8]) / hw.det;
    double wv = fabs(s2) > 1e-12 ? fabs(s2) : 1e-12;
    wv = s2 < 0 ? -wv : (s2 > 0 ? wv : 0.0);
    const double srcx = s0 / wv, srcy = s1 / wv;
    if (METHOD == APS_WARP_NEAREST) {
        const double rx = round(srcx), ry = round(srcy);  // MATLAB round: half away from zero
        const bool valid = rx >= 1 && rx <= in_w && ry >= 1 && ry <= in_h;
        for (int c = 0; c < C; ++c)
            out[((size_t)y * out_w + x) * C + c] = valid ? in[((size_t)((int)ry - 1) * in_w + ((int)rx - 1)) * C + c] : fill;
        return;
    }
    const double fx1 = floor(srcx), fy1 = floor(srcy);
    if (METHOD == APS_WARP_BICUBIC) {
        const bool valid = fx1 >= 2 && fx1 <= in_w - 2 && fy1 >= 2 && fy1 <= in_h - 2;  // :177
        double wxk[4], wyk[4];
        if (valid) {
            const double dx = srcx - fx1, dy = srcy - fy1;
#pragma unroll
            for (int ii = -1; ii <= 2; ++ii) {
                wxk[ii + 1] = warp_cubic((double)ii - dx);
                wyk[ii + 1] = warp_cubic((double)ii - dy);
            }
        }
        for (int c = 0; c < C; ++c) {
            T o = fill;
            if (valid) {
                const int xb = (int)fx1, yb = (int)fy1;
                double v = 0.0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {  // x direction first, taps in ascending order from 0 (:236-243), then y (:246)
                    double xi = 0.0;
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii)
                        xi = xi + (double)in[((size_t)(yb + jj - 2) * in_w + (xb + ii - 2)) * C + c] * wxk[ii];
                    v = v + xi * wyk[jj];
                }
                if (sizeof(T) == 1) {
                    double rr = round(v);
                    rr = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
                    o = (T)rr;
                } else {
                    v = v < 0 ? 0 : (v > 1.0 ? 1.0 : v);  // max(0, min(maxVal, .)) with maxVal = 1 for float images (:216,:254)
                    o = (T)v;
                }
            }
            out[((size_t)y * out_w + x) * C + c] = o;
        }
        return;
    }
    const bool valid = fx1 >= 1 && fx1 + 1 <= in_w && fy1 >= 1 && fy1 + 1 <= in_h;
    for (int c = 0; c < C; ++c) {
        T o = fill;
        if (valid) {
            const int x1 = (int)fx1, y1 = (int)fy1;
            const double wx = srcx - fx1, wy = srcy - fy1;
            const double w11 = (1 - wx) * (1 - wy), w12 = (1 - wx) * wy, w21 = wx * (1 - wy),
                         w22 = wx * wy;
            const double p11 = (double)in[((size_t)(y1 - 1) * in_w + (x1 - 1)) * C + c];
            const double p12 = (double)in[((size_t)(y1)*in_w + (x1 - 1)) * C + c];
            const double p21 = (double)in[((size_t)(y1 - 1) * in_w + (x1)) * C + c];
            const double p22 = (double)in[((size_t)(y1)*in_w + (x1)) * C + c];
            const double v = ((w11 * p11 + w12 * p12) + w21 * p21) + w22 * p22;
            if (sizeof(T) == 1) {
                double rr = round(v);
                rr = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
                o = (T)rr;
            } else {
                o = (T)v;
            }
        }
        out[((size_t)y * out_w + x) * C + c] = o;
    }
}

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
// imresize(in,[oh ow],'bilinear') on float4 images; dimension with the smaller scale first
static void imresize4(const float4* in, int h, int w, int oh, int ow, float4* out, Ws<float4>& tmp) {
    const double sr = (double)oh / h, sc = (double)ow / w;
    if (sr <= sc) {
        tmp.alloc((size_t)oh * w);
        resize_rows_kernel<<<dim3(cdiv(w, 256), oh), 256, 0, stream()>>>(in, h, w, oh, tmp);
        resize_cols_kernel<<<dim3(cdiv(ow, 256), oh), 256, 0, stream()>>>(tmp, oh, w, ow, out);
    } else {
        tmp.alloc((size_t)h * ow);
        resize_cols_kernel<<<dim3(cdiv(ow, 256), h), 256, 0, stream()>>>(in, h, w, ow, tmp);
        resize_rows_kernel<<<dim3(cdiv(ow, 256), oh), 256, 0, stream()>>>(tmp, h, ow, oh, out);
    }
    check_launch("imresize4");
}

// ------------------------------------------------------------------------------------------------
// fused pyramid kernels (same per-stage f32 roundings as the unfused building blocks above)
// ------------------------------------------------------------------------------------------------
// imgaussfilt: column (vertical) pass then row pass through one LDS tile; replicate padding.
constexpr int kBW = 32, kBH = 16;  // output tile of mb_blur_kernel
template <int R>
__global__ __launch_bounds__(256) void mb_blur_kernel(PtrTab ins, RectTab irs, int h, int w, Taps tp, PtrTab outs,
                                                      RectTab ors) {
    const float4* __restrict__ in = ins.p[blockIdx.z];
    float4* __restrict__ out = outs.p[blockIdx.z];
    const Rect ir = irs.r[blockIdx.z], orc = ors.r[blockIdx.z];
    constexpr int IW = kBW + 2 * R, IH = kBH + 2 * R;
    const int x0 = orc.x0 + blockIdx.x * kBW, y0 = orc.y0 + blockIdx.y * kBH, tid = threadIdx.x;
    if (x0 >= orc.x1 || y0 >= orc.y1) return;  // the grid is sized for the largest output rect of the launch
    __shared__ float4 s_in[IH * IW];
    __shared__ float4 s_v[kBH * IW];
    for (int e = tid; e < IH * IW; e += 256) {
        const int ly = e / IW, lx = e - ly * IW;
        const int gy = min(max(y0 + ly - R, 0), h - 1), gx = min(max(x0 + lx - R, 0), w - 1);
        s_in[e] = ld_rect(in, w, ir, gx, gy);
    }
    __syncthreads();
    for (int e = tid; e < kBH * IW; e += 256) {  // vertical pass for every column of the haloed tile
        const int ly = e / IW, lx = e - ly * IW;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], s_in[(ly + t) * IW + lx], a);
        s_v[e] = a;
    }
    __syncthreads();
    for (int e = tid; e < kBH * kBW; e += 256) {  // horizontal pass
        const int ly = e / kBW, lx = e - ly * kBW;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= orc.x1 || gy >= orc.y1) continue;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], s_v[ly * IW + lx + t], a);
        out[(size_t)gy * w + gx] = a;
    }
}

// imresize: both passes in one kernel.  Each thread produces one output pixel by evaluating, for each of its
// second-pass taps, the first-pass result at that intermediate position (an fma chain over the first-pass taps).
// The intermediate value depends only on its own position, so this equals materialising the intermediate image.
// ROWS_FIRST = the reference's rule (smaller scale factor first, ties -> rows).
template <bool ROWS_FIRST>
__device__ __forceinline__ float4 resize_at(const float4* __restrict__ in, int h, int w, const Rect& ir, int Pr, int lr,
                                            const float* wr, int Pc, int lc, const float* wc) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ROWS_FIRST) {  // pass 1 resizes rows (at full width), pass 2 resizes columns
        for (int tc = 0; tc < Pc; ++tc) {
            if (wc[tc] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int xx = min(max(lc + tc, 1), w) - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tr = 0; tr < Pr; ++tr)
                if (wr[tr] != 0.f) v = fma4(wr[tr], ld_rect(in, w, ir, xx, min(max(lr + tr, 1), h) - 1), v);
            a = fma4(wc[tc], v, a);
        }
    } else {
        for (int tr = 0; tr < Pr; ++tr) {
            if (wr[tr] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int yy = min(max(lr + tr, 1), h) - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tc = 0; tc < Pc; ++tc)
                if (wc[tc] != 0.f) v = fma4(wc[tc], ld_rect(in, w, ir, min(max(lc + tc, 1), w) - 1, yy), v);
            a = fma4(wr[tr], v, a);
        }
    }
    return a;
}

template <bool ROWS_FIRST>
__global__ void mb_resize_kernel(PtrTab ins, RectTab irs, int h, int w, int oh, int ow, PtrTab outs, RectTab ors) {
    const float4* __restrict__ in = ins.p[blockIdx.z];
    float4* __restrict__ out = outs.p[blockIdx.z];
    const Rect ir = irs.r[blockIdx.z], orc = ors.r[blockIdx.z];
    // 32 x 4 outputs per 128-thread block: vertically adjacent outputs share most of their input rows (L1 hits)
    const int x = orc.x0 + blockIdx.x * 32 + (threadIdx.x & 31), y = orc.y0 + blockIdx.y * 4 + (threadIdx.x >> 5);
    if (x >= orc.x1 || y >= orc.y1) return;
    int lr, lc;
    float wr[12], wc[12];
    const int Pr = resize_taps(h, oh, y, lr, wr);
    const int Pc = resize_taps(w, ow, x, lc, wc);
    out[(size_t)y * ow + x] = resize_at<ROWS_FIRST>(in, h, w, ir, Pr, lr, wr, Pc, lc, wc);
}

// Blur + downsample in one pass for levels whose next size is exactly half (h == 2*oh, w == 2*ow): a 32 x 8
// tile of the half-resolution output needs a 68 x 20 patch of the blurred image (taps 2o-2 .. 2o+3), which needs
// that patch plus the filter radius of the input.  The blurred full-resolution plane is never stored: it was
// written once and read (twice, measured) by nothing but the downsampler.  Same fma chains in the same order as
// mb_blur_kernel followed by mb_resize_kernel; blurred values outside the blurred footprint are sums of zeros.
constexpr int kFOW = 32, kFOH = 8;
template <int R, bool ROWS_FIRST>
__global__ __launch_bounds__(256) void mb_blur_resize_kernel(PtrTab ins, RectTab irs, int h, int w, Taps tp, int oh,
                                                             int ow, PtrTab outs, RectTab ors) {
    constexpr int BC = 2 * kFOW + 4, BR = 2 * kFOH + 4, IC = BC + 2 * R, IR = BR + 2 * R;
    __shared__ float4 s_a[IR * IC];  // input patch; later the blurred patch [BR][BC]
    __shared__ float4 s_b[BR * IC];  // after the vertical pass
    const float4* __restrict__ in = ins.p[blockIdx.z];
    float4* __restrict__ out = outs.p[blockIdx.z];
    const Rect ir = irs.r[blockIdx.z], orc = ors.r[blockIdx.z];
    const int ox0 = orc.x0 + blockIdx.x * kFOW, oy0 = orc.y0 + blockIdx.y * kFOH, tid = threadIdx.x;
    if (ox0 >= orc.x1 || oy0 >= orc.y1) return;
    const int bx0 = 2 * ox0 - 2, by0 = 2 * oy0 - 2;  // image position of the blurred patch's corner (may be -2)
    for (int e = tid; e < IR * IC; e += 256) {
        const int ly = e / IC, lx = e - ly * IC;
        const int gy = min(max(by0 + ly - R, 0), h - 1), gx = min(max(bx0 + lx - R, 0), w - 1);
        s_a[e] = ld_rect(in, w, ir, gx, gy);
    }
    __syncthreads();
    for (int e = tid; e < BR * IC; e += 256) {  // vertical pass
        const int ly = e / IC, lx = e - ly * IC;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], s_a[(ly + t) * IC + lx], a);
        s_b[e] = a;
    }
    __syncthreads();
    for (int e = tid; e < BR * BC; e += 256) {  // horizontal pass -> blurred patch (reuses s_a)
        const int ly = e / BC, lx = e - ly * BC;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], s_b[ly * IC + lx + t], a);
        s_a[e] = a;
    }
    __syncthreads();
    const int x = ox0 + (tid & 31), y = oy0 + (tid >> 5);
    if (x >= orc.x1 || y >= orc.y1) return;
    int lr, lc;
    float wr[12], wc[12];
    const int Pr = resize_taps(h, oh, y, lr, wr);
    const int Pc = resize_taps(w, ow, x, lc, wc);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ROWS_FIRST) {
        for (int tc = 0; tc < Pc; ++tc) {
            if (wc[tc] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int xx = min(max(lc + tc, 1), w) - 1 - bx0;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tr = 0; tr < Pr; ++tr)
                if (wr[tr] != 0.f) v = fma4(wr[tr], s_a[(min(max(lr + tr, 1), h) - 1 - by0) * BC + xx], v);
            a = fma4(wc[tc], v, a);
        }
    } else {
        for (int tr = 0; tr < Pr; ++tr) {
            if (wr[tr] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int yy = min(max(lr + tr, 1), h) - 1 - by0;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tc = 0; tc < Pc; ++tc)
                if (wc[tc] != 0.f) v = fma4(wc[tc], s_a[yy * BC + min(max(lc + tc, 1), w) - 1 - bx0], v);
            a = fma4(wr[tr], v, a);
        }
    }
    out[(size_t)y * ow + x] = a;
}

// Level l of multiBandBlending.m:136-144 for ALL K layers in one pass:
//   Num_l = sum_k (G_k - imresize(D_k, size_l)) .* w_k      (accumulated in layer order, from zero)
// or, with D == nullptr, the coarsest level (:159): Num_L = sum_k G_k .* w_k.
// A layer whose footprint does not contain the pixel contributes (0 - u) * 0: skipped.
template <bool ROWS_FIRST>
__global__ void mb_lap_all_kernel(PtrTab Gt, RectTab Gr, PtrTab Dt, RectTab Dr, int has_d, int cont, int K, int h, int w,
                                  int dh, int dw, float4* __restrict__ num) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 4 + (threadIdx.x >> 5);  // 32 x 4 per block
    if (x >= w || y >= h) return;
    float acc[3] = {0.f, 0.f, 0.f};
    if (cont) {  // continue the layer-ordered sum of a previous chunk of layers
        const float4 p = num[(size_t)y * w + x];
        acc[0] = p.x;
        acc[1] = p.y;
        acc[2] = p.z;
    }
    if (!has_d) {
        for (int k = 0; k < K; ++k) {
            if (!in_rect(Gr.r[k], x, y)) continue;
            const float4 g = Gt.p[k][(size_t)y * w + x];
            acc[0] = acc[0] + g.x * g.w;
            acc[1] = acc[1] + g.y * g.w;
            acc[2] = acc[2] + g.z * g.w;
        }
    } else {
        int lr = 0, lc = 0, Pr = 0, Pc = 0;
        float wr[12], wc[12];
        bool have_taps = false;
        for (int k = 0; k < K; ++k) {
            if (!in_rect(Gr.r[k], x, y)) continue;
            if (!have_taps) {
                Pr = resize_taps(dh, h, y, lr, wr);
                Pc = resize_taps(dw, w, x, lc, wc);
                have_taps = true;
            }
            const float4 u = resize_at<ROWS_FIRST>(Dt.p[k], dh, dw, Dr.r[k], Pr, lr, wr, Pc, lc, wc);
            const float4 g = Gt.p[k][(size_t)y * w + x];
            acc[0] = acc[0] + (g.x - u.x) * g.w;
            acc[1] = acc[1] + (g.y - u.y) * g.w;
            acc[2] = acc[2] + (g.z - u.z) * g.w;
        }
    }
    num[(size_t)y * w + x] = make_float4(acc[0], acc[1], acc[2], 0.f);
}

// collapse step (:166): F_l = imresize(F_{l+1}, size_l) + Num_l
template <bool ROWS_FIRST>
__global__ void mb_collapse_kernel(const float4* __restrict__ Fc, int ch, int cw, const float4* __restrict__ num, int h,
                                   int w, float4* __restrict__ out) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 4 + (threadIdx.x >> 5);  // 32 x 4 per block
    if (x >= w || y >= h) return;
    int lr, lc;
    float wr[12], wc[12];
    const int Pr = resize_taps(ch, h, y, lr, wr);
    const int Pc = resize_taps(cw, w, x, lc, wc);
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ROWS_FIRST) {
        for (int tc = 0; tc < Pc; ++tc) {
            if (wc[tc] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int xx = min(max(lc + tc, 1), cw) - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tr = 0; tr < Pr; ++tr)
                if (wr[tr] != 0.f) v = fma4(wr[tr], Fc[(size_t)(min(max(lr + tr, 1), ch) - 1) * cw + xx], v);
            u = fma4(wc[tc], v, u);
        }
    } else {
        for (int tr = 0; tr < Pr; ++tr) {
            if (wr[tr] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
            const int yy = min(max(lr + tr, 1), ch) - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int tc = 0; tc < Pc; ++tc)
                if (wc[tc] != 0.f) v = fma4(wc[tc], Fc[(size_t)yy * cw + min(max(lc + tc, 1), cw) - 1], v);
            u = fma4(wr[tr], v, u);
        }
    }
    const float4 n = num[(size_t)y * w + x];
    out[(size_t)y * w + x] = make_float4(u.x + n.x, u.y + n.y, u.z + n.z, 0.f);
}

static PtrTab make_tab(float4* const* p, int count) {
    PtrTab t;
    for (int k = 0; k < kMaxK; ++k) t.p[k] = k < count ? p[k] : nullptr;
    return t;
}
static RectTab make_rtab(const Rect* r, int count) {
    RectTab t;
    for (int k = 0; k < kMaxK; ++k) t.r[k] = k < count ? r[k] : Rect{0, 0, 0, 0};
    return t;
}
// fuseTile's and multiBandBlending's weight normalisations over K layers (either or both), coverage optional.
// rects (optional, K <= kMaxK only): the layers' footprints in the h x w tile.
static void normalize_weights(const std::vector<float4*>& layers, const Rect* rects, int h, int w, int fuse_norm,
                              int mbb_norm, uint8_t* cov) {
    const int K = (int)layers.size();
    const size_t n = (size_t)h * w;
    if (K <= kMaxK) {
        norm_weights_kernel<PtrTab><<<cdiv(n, 256), 256, 0, stream()>>>(
            make_tab(layers.data(), K), make_rtab(rects, rects ? K : 0), rects ? 1 : 0, K, w, n, fuse_norm, mbb_norm, cov);
        check_launch("norm_weights_kernel");
    } else {
        APS_REQUIRE(K <= 64, APS_E_DIM, "more than 64 layers in one tile (%d)", K);
        Ws<float4*> dl(K);
        APS_HIP(hipMemcpyAsync(dl, layers.data(), K * sizeof(float4*), hipMemcpyHostToDevice, stream()));
        norm_weights_kernel<float4* const*><<<cdiv(n, 256), 256, 0, stream()>>>(dl.get(), RectTab{}, 0, K, w, n, fuse_norm,
                                                                               mbb_norm, cov);
        check_launch("norm_weights_kernel");
        APS_HIP(hipStreamSynchronize(stream()));
    }
}

// multiBandBlending on K float4 layers whose weights are ALREADY normalised (normalize_weights(..., mbb=1));
// result float4 in F (unclamped).  Level-major: per level one launch blurs up to 16 layers, one launch
// downsamples them, one pass forms their Laplacians and accumulates them in layer order.  No host sync.
// rects (optional): footprints of the layers; every kernel then works on footprints only (see struct Rect).
static void multiband_device(const std::vector<float4*>& layers, const Rect* rects, int h, int w, int levels, float sigma,
                             float4* F) {
    const int K = (int)layers.size();
    const size_t hw = (size_t)h * w;
    Prof prof("multiband");
    int maxl = (int)std::floor(std::log2((double)std::min(h, w)));  // :98-109
    levels = std::max(1, std::min(levels, maxl));
    std::vector<int> lh(levels), lw(levels);
    lh[0] = h;
    lw[0] = w;
    for (int l = 1; l < levels; ++l) {
        lh[l] = std::max(1, lh[l - 1] / 2);
        lw[l] = std::max(1, lw[l - 1] / 2);
    }
    std::vector<Ws<float4>> store((size_t)std::max(levels - 1, 0) * K), blurred(std::min(K, kMaxK)), num(levels);
    std::vector<std::vector<float4*>> lev(levels, std::vector<float4*>(K));
    lev[0] = layers;
    for (int l = 1; l < levels; ++l)
        for (int k = 0; k < K; ++k) {
            store[(size_t)(l - 1) * K + k].alloc((size_t)lh[l] * lw[l]);
            lev[l][k] = store[(size_t)(l - 1) * K + k];
        }
    std::vector<float4*> bl(std::min(K, kMaxK), nullptr);
    if (levels > 1)
        for (int k = 0; k < (int)bl.size(); ++k) {
            blurred[k].alloc(hw);
            bl[k] = blurred[k];
        }
    for (int l = 0; l < levels; ++l) num[l].alloc((size_t)lh[l] * lw[l]);
    const Taps tp = make_taps(sigma);
    APS_REQUIRE(tp.r >= 1 && tp.r <= 4, APS_E_ARG, "pyrSigma %g needs a %d-tap filter; 3..9 taps are built", (double)sigma, 2 * tp.r + 1);
    // footprints per level: G_l, blurred G_l (grown by the filter radius), G_{l+1} (mapped through the resize)
    std::vector<std::vector<Rect>> gr(levels, std::vector<Rect>(K)), br(levels, std::vector<Rect>(K));
    for (int k = 0; k < K; ++k) gr[0][k] = rects ? clip_rect(rects[k], w, h) : Rect{0, 0, w, h};
    for (int l = 0; l < levels; ++l)
        for (int k = 0; k < K; ++k) {
            const Rect g = gr[l][k];
            const bool empty = g.x1 <= g.x0;
            br[l][k] = empty ? g : clip_rect(Rect{g.x0 - tp.r, g.y0 - tp.r, g.x1 + tp.r, g.y1 + tp.r}, lw[l], lh[l]);
            if (l + 1 < levels) gr[l + 1][k] = empty ? g : map_rect(br[l][k], lh[l], lw[l], lh[l + 1], lw[l + 1]);
        }
    auto span = [](const Rect* r, int count, int& mw, int& mh) {
        mw = mh = 0;
        for (int k = 0; k < count; ++k) {
            mw = std::max(mw, r[k].x1 - r[k].x0);
            mh = std::max(mh, r[k].y1 - r[k].y0);
        }
    };
    for (int l = 0; l < levels; ++l) {
        const int hl = lh[l], wl = lw[l];
        const bool last = l == levels - 1;
        const int nh = last ? 0 : lh[l + 1], nw = last ? 0 : lw[l + 1];
        float4* dst = (last && levels == 1) ? F : num[l].get();
        for (int k0 = 0; k0 < K; k0 += kMaxK) {
            const int kc = std::min(kMaxK, K - k0);
            const PtrTab gt = make_tab(lev[l].data() + k0, kc);
            const RectTab grt = make_rtab(gr[l].data() + k0, kc);
            PtrTab dt = gt;
            RectTab drt = grt;
            if (!last) {
                const PtrTab bt = make_tab(bl.data(), kc);
                const RectTab brt = make_rtab(br[l].data() + k0, kc);
                dt = make_tab(lev[l + 1].data() + k0, kc);
                drt = make_rtab(gr[l + 1].data() + k0, kc);
                int mw, mh;
                const bool fused = hl == 2 * nh && wl == 2 * nw && !std::getenv("APS_RENDER_NO_FUSE");
                if (fused) {
                    span(gr[l + 1].data() + k0, kc, mw, mh);
                    if (mw > 0 && mh > 0) {
                        const dim3 fg(cdiv(mw, kFOW), cdiv(mh, kFOH), kc);
                        const bool rf = rows_first(hl, wl, nh, nw);
#define APS_FUSE_CASE(RR)                                                                                              \
    case RR:                                                                                                           \
        if (rf)                                                                                                        \
            mb_blur_resize_kernel<RR, true><<<fg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, nh, nw, dt, drt);           \
        else                                                                                                           \
            mb_blur_resize_kernel<RR, false><<<fg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, nh, nw, dt, drt);          \
        break;
                        switch (tp.r) {
                            APS_FUSE_CASE(1)
                            APS_FUSE_CASE(2)
                            APS_FUSE_CASE(3)
                            default:
                                APS_FUSE_CASE(4)
                        }
#undef APS_FUSE_CASE
                    }
                }
                span(br[l].data() + k0, kc, mw, mh);
                if (!fused && mw > 0 && mh > 0) {
                    const dim3 bg(cdiv(mw, kBW), cdiv(mh, kBH), kc);
                    switch (tp.r) {
                        case 1: mb_blur_kernel<1><<<bg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, bt, brt); break;
                        case 2: mb_blur_kernel<2><<<bg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, bt, brt); break;
                        case 3: mb_blur_kernel<3><<<bg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, bt, brt); break;
                        default: mb_blur_kernel<4><<<bg, 256, 0, stream()>>>(gt, grt, hl, wl, tp, bt, brt); break;
                    }
                }
                span(gr[l + 1].data() + k0, kc, mw, mh);
                if (!fused && mw > 0 && mh > 0) {
                    const dim3 rg(cdiv(mw, 32), cdiv(mh, 4), kc);
                    if (rows_first(hl, wl, nh, nw))
                        mb_resize_kernel<true><<<rg, 128, 0, stream()>>>(bt, brt, hl, wl, nh, nw, dt, drt);
                    else
                        mb_resize_kernel<false><<<rg, 128, 0, stream()>>>(bt, brt, hl, wl, nh, nw, dt, drt);
                }
            }
            const dim3 lg(cdiv(wl, 32), cdiv(hl, 4));
            if (last || rows_first(nh, nw, hl, wl))
                mb_lap_all_kernel<true><<<lg, 128, 0, stream()>>>(gt, grt, dt, drt, last ? 0 : 1, k0 > 0, kc, hl, wl, nh, nw, dst);
            else
                mb_lap_all_kernel<false><<<lg, 128, 0, stream()>>>(gt, grt, dt, drt, 1, k0 > 0, kc, hl, wl, nh, nw, dst);
            check_launch("multiband level");
        }
    }
    // collapse (:163-167)
    std::vector<Ws<float4>> fl(std::max(levels - 1, 0));
    const float4* cur = levels > 1 ? num[levels - 1].get() : nullptr;
    for (int l = levels - 2; l >= 0; --l) {
        float4* dst = F;
        if (l > 0) {
            fl[l].alloc((size_t)lh[l] * lw[l]);
            dst = fl[l];
        }
        if (rows_first(lh[l + 1], lw[l + 1], lh[l], lw[l]))
            mb_collapse_kernel<true><<<dim3(cdiv(lw[l], 32), cdiv(lh[l], 4)), 128, 0, stream()>>>(cur, lh[l + 1], lw[l + 1], num[l], lh[l], lw[l], dst);
        else
            mb_collapse_kernel<false><<<dim3(cdiv(lw[l], 32), cdiv(lh[l], 4)), 128, 0, stream()>>>(cur, lh[l + 1], lw[l + 1], num[l], lh[l], lw[l], dst);
        check_launch("mb_collapse_kernel");
        cur = dst;
    }
    // no synchronisation: all buffers are stream-ordered workspace of this thread's stream
}

struct ImgJob {
    const uint8_t* src;
    uint32_t* dst;
    float* wx;
    float* wy;
    int h, w, c, layout;
};

// RGBA8 conversion of ALL images in one launch (blockIdx.y = image).  Interleaved RGB with a 4-byte aligned base is
// read as three dwords per four pixels and written as one 16-byte store; everything else goes pixel by pixel.
__global__ __launch_bounds__(256) void to_rgba_batch_kernel(const ImgJob* __restrict__ jobs) {
    const ImgJob j = jobs[blockIdx.y];
    const size_t npx = (size_t)j.h * j.w;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j.layout == APS_IMG_U8_HWC && j.c == 3 && (reinterpret_cast<uintptr_t>(j.src) & 3u) == 0) {
        const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(j.src);
        const size_t nq = npx / 4;
        for (size_t q = t0; q < nq; q += stride) {
            const uint32_t a = s32[3 * q], b = s32[3 * q + 1], c = s32[3 * q + 2];
            uint4 o;
            o.x = (a & 0x00ffffffu) | 0xff000000u;
            o.y = ((a >> 24) | (b << 8)) & 0x00ffffffu;
            o.y |= 0xff000000u;
            o.z = ((b >> 16) | (c << 16)) & 0x00ffffffu;
            o.z |= 0xff000000u;
            o.w = (c >> 8) | 0xff000000u;
            reinterpret_cast<uint4*>(j.dst)[q] = o;
        }
        for (size_t p = nq * 4 + t0; p < npx; p += stride)
            j.dst[p] = (uint32_t)j.src[3 * p] | ((uint32_t)j.src[3 * p + 1] << 8) | ((uint32_t)j.src[3 * p + 2] << 16) | 0xff000000u;
        return;
    }
    for (size_t p = t0; p < npx; p += stride) {
        const int y = (int)(p / (size_t)j.w), x = (int)(p - (size_t)y * j.w);
        uint32_t ch[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kk = j.c == 1 ? 0 : k;
            ch[k] = j.layout == APS_IMG_U8_HWC ? j.src[p * j.c + kk] : j.src[(size_t)kk * npx + (size_t)x * j.h + y];
        }
        j.dst[p] = ch[0] | (ch[1] << 8) | (ch[2] << 16) | 0xff000000u;
    }
}

// warpWeights (renderPanorama.m:1282-1312): wx(1:ceil(w/2)) = linspace(0,1,.), wx(floor(w/2)+1:w) = linspace(1,0,.)
// (the second assignment wins where they overlap), evaluated in f64 and cast; blockIdx.y = 2*image + (0: wx, 1: wy)
__device__ __forceinline__ float tent_value(int n, int k) {
    const int a = (n + 1) / 2, b0 = n / 2, nb = n - n / 2;
    if (k >= b0) {
        const int q = k - b0;
        double v = nb > 1 ? 1.0 + ((double)q * -1.0) / (double)(nb - 1) : 0.0;
        if (q == 0 && nb > 1) v = 1.0;
        if (q == nb - 1) v = 0.0;
        return (float)v;
    }
    double v = a > 1 ? 0.0 + ((double)k * 1.0) / (double)(a - 1) : 1.0;
    if (k == a - 1) v = 1.0;
    return (float)v;
}
__global__ void tent_batch_kernel(const ImgJob* __restrict__ jobs) {
    const ImgJob j = jobs[blockIdx.y >> 1];
    const int n = (blockIdx.y & 1) ? j.h : j.w;
    float* __restrict__ out = (blockIdx.y & 1) ? j.wy : j.wx;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = tent_value(n, k);
}

struct PreparedImages {
    Ws<uint32_t> rgba;  // all images back to back (each 16-byte aligned)
    Ws<float> tent;     // wx, wy of every image back to back
    std::vector<In<uint8_t>> src;
    std::vector<DevImage> host;
    std::vector<ImgJob> jobs;
    Ws<DevImage> dev;
    Ws<ImgJob> djobs;
    std::vector<char> converted;   // image i's pixels and tent tables are in place
    std::vector<std::vector<ImgJob>> sel;  // the job lists of the convert_images() calls (host copies kept until the entry point returns)
};

// Pixels (uint8 -> packed RGBA words) and tent tables of the images marked in `used` (all when null) that have not been converted
// yet.  A rank of a sharded render only meets the views its tiles show - a third of them at 8 ranks on the 64-view scene - and
// converting all 64 was a quarter of its render's kernel time.
static void convert_images(PreparedImages& P, const std::vector<char>* used) {
    const int n = (int)P.jobs.size();
    std::vector<ImgJob> sel;
    int max_len = 1;
    for (int i = 0; i < n; ++i) {
        if (P.converted[i] || (used && !(*used)[i])) continue;
        P.converted[i] = 1;
        sel.push_back(P.jobs[i]);
        max_len = std::max(max_len, std::max(P.jobs[i].h, P.jobs[i].w));
    }
    if (sel.empty()) return;
    const int m = (int)sel.size();
    P.sel.push_back(std::move(sel));
    APS_HIP(hipMemcpyAsync(P.djobs, P.sel.back().data(), m * sizeof(ImgJob), hipMemcpyHostToDevice, stream()));
    to_rgba_batch_kernel<<<dim3(512, m), 256, 0, stream()>>>(P.djobs);
    tent_batch_kernel<<<dim3(cdiv(max_len, 256), 2 * m), 256, 0, stream()>>>(P.djobs);
    check_launch("to_rgba_batch_kernel");
}

// No synchronisation here: the host tables are members of P and outlive the stream work of the calling entry point.
static void prepare_images(const aps_image* images, int n, PreparedImages& P, bool convert_now = true) {
    P.src.resize(n);
    P.converted.assign(n, 0);
    P.host.resize(n);
    P.jobs.resize(n);
    size_t px_total = 0, tent_total = 0;
    int max_len = 1;
    for (int i = 0; i < n; ++i) {
        const aps_image& im = images[i];
        APS_REQUIRE(im.data != nullptr, APS_E_ARG, "image %d: NULL data", i);
        APS_REQUIRE(im.height > 0 && im.width > 0, APS_E_DIM, "image %d: empty", i);
        APS_REQUIRE(im.channels == 1 || im.channels == 3, APS_E_DIM, "image %d: channels must be 1 or 3", i);
        APS_REQUIRE(im.layout == APS_IMG_U8_HWC || im.layout == APS_IMG_U8_MATLAB, APS_E_TYPE,
                    "image %d: unknown layout", i);
        px_total += ((size_t)im.height * im.width + 3) & ~size_t(3);
        tent_total += (size_t)im.height + im.width;
        max_len = std::max(max_len, std::max(im.height, im.width));
    }
    P.rgba.alloc(px_total);
    P.tent.alloc(tent_total);
    size_t po = 0, to = 0;
    for (int i = 0; i < n; ++i) {
        const aps_image& im = images[i];
        const size_t px = (size_t)im.height * im.width;
        P.src[i].bind(im.data, px * im.channels);
        DevImage& d = P.host[i];
        d.rgba = P.rgba.get() + po;
        d.wx = P.tent.get() + to;
        d.wy = P.tent.get() + to + im.width;
        P.jobs[i] = ImgJob{P.src[i].get(), P.rgba.get() + po, P.tent.get() + to, P.tent.get() + to + im.width,
                           im.height, im.width, im.channels, im.layout};
        po += (px + 3) & ~size_t(3);
        to += (size_t)im.height + im.width;
        d.h = im.height;
        d.w = im.width;
        for (int e = 0; e < 9; ++e) d.R[e] = (float)im.R[e];
        d.fx = (float)im.K[0];
        d.fy = (float)im.K[4];
        d.cx = (float)im.K[6];
        d.cy = (float)im.K[7];
        for (int c = 0; c < 3; ++c) d.gain[c] = im.gain[c];
        for (int c = 0; c < 3; ++c) d.g255[c] = (float)((double)im.gain[c] / 255.0);
        {   // tent_value(n, k) = k / (a - 1) on the rising half, (n - 1 - k) / (nb - 1) on the falling one, a = ceil(n / 2),
            // nb = n - floor(n / 2); both halves are straight lines through the table's samples and the table is their minimum
            // capped at one, so between two samples the bilinear interpolation of the table IS this function
            auto slopes = [](int n, float t[2]) {
                const int a = (n + 1) / 2, nb = n - n / 2;
                t[0] = a > 1 ? (float)(1.0 / (double)(a - 1)) : 0.f;
                t[1] = nb > 1 ? (float)(1.0 / (double)(nb - 1)) : 0.f;
            };
            slopes(im.width, d.tx);
            slopes(im.height, d.ty);
        }
        {
            const double ax = std::max(std::fabs(1.0 - im.K[6]), std::fabs((double)im.width - im.K[6])) / std::fabs(im.K[0]);
            const double ay = std::max(std::fabs(1.0 - im.K[7]), std::fabs((double)im.height - im.K[7])) / std::fabs(im.K[4]);
            const double c = 1.0 / std::sqrt(1.0 + ax * ax + ay * ay);
            d.cmin = std::isfinite(c) ? (float)(c * (1.0 - 1e-4) - 1e-6) : -1.0f;
        }
    }
    P.dev.alloc(n);
    P.djobs.alloc(n);
    APS_HIP(hipMemcpyAsync(P.dev, P.host.data(), n * sizeof(DevImage), hipMemcpyHostToDevice, stream()));
    (void)max_len;
    if (convert_now) convert_images(P, nullptr);
}

static DevCanvas make_canvas(const aps_canvas& c) {
    APS_REQUIRE(c.mode >= APS_PROJ_CYLINDRICAL && c.mode <= APS_PROJ_STEREOGRAPHIC, APS_E_ARG, "unknown projection mode %d", c.mode);
    APS_REQUIRE(c.height > 0 && c.width > 0, APS_E_DIM, "empty canvas");
    APS_REQUIRE(c.f_pan > 0, APS_E_ARG, "fPan must be positive");
    DevCanvas d;
    d.mode = c.mode;
    d.H = c.height;
    d.W = c.width;
    d.f = (float)c.f_pan;
    d.o0 = (float)c.origin0;
    d.o1 = (float)c.origin1;
    // Rref entries in the order the ray formula uses them: row c of R_ref' = column c of R_ref
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) d.Rref[3 * col + row] = (float)c.R_ref[row + 3 * col];
    return d;
}

}  // namespace aps

using namespace aps;

extern "C" {

int aps_warp_tile(const aps_image* image, const aps_canvas* canvas, int r0, int c0, int ht, int wt,
                  float angle_power, float* S, uint8_t* M, float* Wang, float* Wf) {
    return guarded([&] {
        APS_REQUIRE(image && canvas && S && M && Wang && Wf, APS_E_ARG, "NULL argument");
        APS_REQUIRE(ht > 0 && wt > 0 && r0 >= 0 && c0 >= 0, APS_E_ARG, "bad tile");
        ctx();
        PreparedImages P;
        prepare_images(image, 1, P);
        const DevCanvas cv = make_canvas(*canvas);
        const size_t n = (size_t)ht * wt;
        Out<float> oS(S, n * 3), oA(Wang, n), oF(Wf, n);
        Out<uint8_t> oM(M, n);
        warp_tile_kernel<<<dim3(cdiv(wt, 32), cdiv(ht, 8)), 256, 0, stream()>>>(cv, P.dev, r0, c0, ht, wt,
                                                                                angle_power, oS, oM, oA, oF);
        check_launch("warp_tile_kernel");
        oS.commit();
        oM.commit();
        oA.commit();
        oF.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_multiband_blend(const float* C, const float* Wt, int k, int h, int w, int levels, float sigma,
                        float* F) {
    return guarded([&] {
        APS_REQUIRE(C && Wt && F, APS_E_ARG, "NULL argument");
        APS_REQUIRE(k >= 1 && h > 0 && w > 0, APS_E_DIM, "need K >= 1 non-empty layers");
        APS_REQUIRE(levels >= 1, APS_E_ARG, "levels must be a positive integer");
        APS_REQUIRE(sigma > 0, APS_E_ARG, "sigma must be positive");
        ctx();
        const size_t hw = (size_t)h * w;
        In<float> dC(C, hw * 3 * k), dW(Wt, hw * k);
        Out<float> oF(F, hw * 3);
        std::vector<Ws<float4>> store(k);
        std::vector<float4*> layers(k);
        for (int i = 0; i < k; ++i) {
            store[i].alloc(hw);
            pack_layer_kernel<<<cdiv(hw, 256), 256, 0, stream()>>>(dC.get() + (size_t)i * hw * 3,
                                                                   dW.get() + (size_t)i * hw, hw, store[i]);
            layers[i] = store[i];
        }
        check_launch("pack_layer_kernel");
        Ws<float4> F4(hw);
        normalize_weights(layers, nullptr, h, w, 0, 1, nullptr);  // multiBandBlending.m:72-85
        multiband_device(layers, nullptr, h, w, levels, sigma, F4);
        unpack_clamp_kernel<<<cdiv(hw, 256), 256, 0, stream()>>>(F4, hw, 1, oF);
        check_launch("unpack_clamp_kernel");
        oF.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_linear_blend(const float* C, const float* Wt, int k, int h, int w, float* F) {
    return guarded([&] {
        APS_REQUIRE(C && Wt && F, APS_E_ARG, "NULL argument");
        APS_REQUIRE(k >= 1 && h > 0 && w > 0, APS_E_DIM, "need K >= 1 non-empty layers");
        ctx();
        const size_t hw = (size_t)h * w;
        In<float> dC(C, hw * 3 * k), dW(Wt, hw * k);
        Out<float> oF(F, hw * 3);
        std::vector<Ws<float4>> store(k);
        std::vector<float4*> layers(k);
        for (int i = 0; i < k; ++i) {
            store[i].alloc(hw);
            pack_layer_kernel<<<cdiv(hw, 256), 256, 0, stream()>>>(dC.get() + (size_t)i * hw * 3,
                                                                   dW.get() + (size_t)i * hw, hw, store[i]);
            layers[i] = store[i];
        }
        Ws<float4*> dl(k);
        APS_HIP(hipMemcpyAsync(dl, layers.data(), k * sizeof(float4*), hipMemcpyHostToDevice, stream()));
        Ws<float4> F4(hw);
        linear_blend_kernel<float4* const*><<<cdiv(hw, 256), 256, 0, stream()>>>(dl.get(), k, hw, F4);
        unpack_clamp_kernel<<<cdiv(hw, 256), 256, 0, stream()>>>(F4, hw, 0, oF);
        check_launch("linear_blend_kernel");
        oF.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

int aps_render(const aps_image* images, int n_img, const aps_canvas* canvas,
               const aps_render_opts* opts, int out_layout, uint8_t* pano, uint8_t* covered) {
    return aps_render_tiles(images, n_img, canvas, opts, out_layout, 0, 1, pano, covered);
}

// the tile loop for the tiles t (row-major index) with first <= t < last and (t - first) % step == 0; every = all tiles of the canvas
static int render_tiles_impl(const aps_image* images, int n_img, const aps_canvas* canvas, const aps_render_opts* opts, int out_layout,
                             int tile_first, int tile_step, int tile_last, bool every, uint8_t* pano, uint8_t* covered);

int aps_render_tiles(const aps_image* images, int n_img, const aps_canvas* canvas,
                     const aps_render_opts* opts, int out_layout, int tile_first, int tile_step,
                     uint8_t* pano, uint8_t* covered) {
    if (!(tile_step >= 1 && tile_first >= 0 && tile_first < tile_step))
        return guarded([&] { fail(APS_E_ARG, "bad tile subset"); });
    return render_tiles_impl(images, n_img, canvas, opts, out_layout, tile_first, tile_step, 0x7fffffff, tile_step == 1, pano, covered);
}

int aps_render_tile_range(const aps_image* images, int n_img, const aps_canvas* canvas, const aps_render_opts* opts, int out_layout,
                          int tile_begin, int tile_end, uint8_t* pano, uint8_t* covered) {
    if (!(tile_begin >= 0 && tile_end >= tile_begin))
        return guarded([&] { fail(APS_E_ARG, "bad tile range"); });
    return render_tiles_impl(images, n_img, canvas, opts, out_layout, tile_begin, 1, tile_end, false, pano, covered);
}

static int render_tiles_impl(const aps_image* images, int n_img, const aps_canvas* canvas, const aps_render_opts* opts, int out_layout,
                             int tile_first, int tile_step, int tile_last, bool every, uint8_t* pano, uint8_t* covered) {
    return guarded([&] {
        APS_REQUIRE(images && canvas && opts && pano, APS_E_ARG, "NULL argument");
        APS_REQUIRE(n_img >= 1, APS_E_ARG, "need at least one image");
        APS_REQUIRE(opts->tile_h > 0 && opts->tile_w > 0, APS_E_ARG,
                    "opts.tile must be given explicitly (never derived from free memory)");
        APS_REQUIRE(opts->blending >= APS_BLEND_NONE && opts->blending <= APS_BLEND_MULTIBAND, APS_E_ARG, "unknown blending");
        APS_REQUIRE(out_layout == APS_IMG_U8_HWC || out_layout == APS_IMG_U8_MATLAB, APS_E_TYPE, "unknown output layout");
        if (opts->blending == APS_BLEND_MULTIBAND) {
            APS_REQUIRE(opts->pyr_levels >= 1, APS_E_ARG, "pyrLevels must be >= 1");
            APS_REQUIRE(opts->pyr_sigma > 0, APS_E_ARG, "pyrSigma must be positive");
        }
        ctx();
        PreparedImages P;
        // all tiles: every image's pixels at once, converted while the host works out the footprints; a share of the tiles (a
        // rank of a sharded render): tables now, pixels once the footprints say which images the share meets
        prepare_images(images, n_img, P, every);
        const std::function<void(const std::vector<char>&)> convert = [&P](const std::vector<char>& used) { convert_images(P, &used); };
        const DevCanvas cv = make_canvas(*canvas);
        const int H = cv.H, W = cv.W;
        const size_t HW = (size_t)H * W;
        Out<uint8_t> oP(pano, HW * 3), oC(covered, HW);
        const int TH = opts->tile_h, TW = opts->tile_w;
        const size_t tmax = (size_t)std::min(TH, H) * std::min(TW, W);
        Ws<float4> F(tmax);
        Ws<uint8_t> cov(tmax);
        std::vector<Ws<float4>> store;
        // a host buffer receives the whole canvas back: start from its current content so that tiles of
        // other ranks are left untouched
        if (!every) {
            if (oP.host) APS_HIP(hipMemcpyAsync(oP.d, oP.host, HW * 3, hipMemcpyHostToDevice, stream()));
            if (oC.host) APS_HIP(hipMemcpyAsync(oC.d, oC.host, HW, hipMemcpyHostToDevice, stream()));
        }
        using Tile = TileRect;
        std::vector<Tile> tiles;
        int tile_index = -1;
        for (int r0 = 0; r0 < H; r0 += TH)
            for (int c0 = 0; c0 < W; c0 += TW) {
                ++tile_index;
                if (tile_index < tile_first || tile_index >= tile_last || (tile_index - tile_first) % tile_step != 0) continue;
                tiles.push_back({r0, c0, std::min(TH, H - r0), std::min(TW, W - c0)});
            }
        const int nt = (int)tiles.size();
        // multiband: all tiles level-major in one launch sequence (render_batch.hip); APS_RENDER_LEGACY=1 keeps the
        // per-tile path below, whose arithmetic the batched kernels reproduce bit for bit
        if (opts->blending == APS_BLEND_MULTIBAND && !std::getenv("APS_RENDER_LEGACY") &&
            render_multiband_batched(P.dev, P.host.data(), n_img, cv, *opts, tiles, out_layout, oP, oC.present() ? oC.get() : nullptr, convert)) {
            oP.commit();
            oC.commit();
            APS_HIP(hipStreamSynchronize(stream()));
            return;
        }
        // 'linear' / 'none': one fused launch over all tiles (render_batch.hip, rw_fuse_kernel); APS_RENDER_LEGACY=1 keeps the
        // per-tile kernels below, whose bytes it reproduces
        if (opts->blending != APS_BLEND_MULTIBAND && !std::getenv("APS_RENDER_LEGACY") &&
            render_fuse_batched(P.dev, P.host.data(), n_img, cv, *opts, tiles, out_layout, oP, oC.present() ? oC.get() : nullptr, convert)) {
            oP.commit();
            oC.commit();
            APS_HIP(hipStreamSynchronize(stream()));
            return;
        }
        convert_images(P, nullptr);  // the per-tile path below samples whatever its cover pass finds
        // phase 1: footprint of every image in every tile (the reference skips images with ~any(Mi), :989;
        // here the bounding box of Mi also bounds all later work on that layer); one read-back
        std::vector<int> hbox((size_t)nt * n_img * 4, 0);
        if (opts->blending != APS_BLEND_NONE && nt > 0) {
            const int nby_max = cdiv(std::min(TH, H), 8);
            std::vector<int> dims((size_t)nt * 4);
            for (int t = 0; t < nt; ++t) {
                int xs = 0;
                while ((cdiv(tiles[t].wt, 32) >> xs) > 64 || (((cdiv(tiles[t].wt, 32) - 1) >> xs) > 63)) ++xs;
                dims[4 * t + 0] = tiles[t].ht;
                dims[4 * t + 1] = tiles[t].wt;
                dims[4 * t + 2] = cdiv(tiles[t].ht, 8);
                dims[4 * t + 3] = xs;
            }
            Ws<unsigned long long> rowmask((size_t)nt * n_img * nby_max);
            Ws<int> bbox(hbox.size()), ddims(dims.size());
            APS_HIP(hipMemsetAsync(rowmask, 0, (size_t)nt * n_img * nby_max * sizeof(unsigned long long), stream()));
            APS_HIP(hipMemcpyAsync(ddims, dims.data(), dims.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
            {
                Prof prof("cover");
                for (int t = 0; t < nt; ++t) {
                    const Tile& tl = tiles[t];
                    cover_kernel<<<dim3(cdiv(tl.wt, 32), cdiv(tl.ht, 8)), 256, 0, stream()>>>(
                        cv, P.dev, n_img, tl.r0, tl.c0, tl.ht, tl.wt, opts->angle_power, nby_max, dims[4 * t + 3],
                        std::getenv("APS_RENDER_NO_CULL") ? 0 : 1, rowmask.get() + (size_t)t * n_img * nby_max);
                }
                footprint_kernel<<<cdiv((size_t)nt * n_img, 256), 256, 0, stream()>>>(rowmask, ddims, n_img, nby_max,
                                                                                     nt * n_img, bbox);
            }
            check_launch("cover_kernel");
            APS_HIP(hipMemcpyAsync(hbox.data(), bbox, hbox.size() * sizeof(int), hipMemcpyDeviceToHost, stream()));
            APS_HIP(hipStreamSynchronize(stream()));  // also keeps `dims` alive until its upload has run
        }
        // phase 2: tiles back to back on the stream, no host round trip in between
        for (int t = 0; t < nt; ++t) {
            const Tile& tl = tiles[t];
            const int r0 = tl.r0, c0 = tl.c0, ht = tl.ht, wt = tl.wt;
            const size_t T = (size_t)ht * wt;
            const dim3 g2(cdiv(wt, 32), cdiv(ht, 8));
            if (opts->blending == APS_BLEND_NONE) {
                APS_HIP(hipMemsetAsync(F, 0, T * sizeof(float4), stream()));
                APS_HIP(hipMemsetAsync(cov, 0, T, stream()));
                for (int i = 0; i < n_img; ++i)
                    none_fuse_kernel<<<g2, 256, 0, stream()>>>(cv, P.dev, i, r0, c0, ht, wt, opts->angle_power,
                                                               opts->none_policy, F, cov);
                check_launch("none_fuse_kernel");
            } else {
                std::vector<int> contrib;
                std::vector<Rect> rects;
                for (int i = 0; i < n_img; ++i) {
                    const int* b = &hbox[((size_t)t * n_img + i) * 4];
                    if (b[2] > b[0] && b[3] > b[1]) {
                        contrib.push_back(i);
                        rects.push_back(Rect{b[0], b[1], b[2], b[3]});
                    }
                }
                const int K = (int)contrib.size();
                // above kMaxK layers the per-pixel kernels take an uploaded pointer table and no footprints
                const bool culled = K <= kMaxK && !std::getenv("APS_RENDER_NO_CULL");
                if (!culled)
                    for (auto& r : rects) r = Rect{0, 0, wt, ht};
                if (K == 0) {
                    APS_HIP(hipMemsetAsync(F, 0, T * sizeof(float4), stream()));
                    APS_HIP(hipMemsetAsync(cov, 0, T, stream()));
                } else {
                    if ((int)store.size() < K) store.resize(K);
                    std::vector<float4*> layers(K);
                    const float wf_floor = opts->blending == APS_BLEND_LINEAR ? 1e-4f : 0.f;
                    {
                        Prof prof("warp_layer");
                        for (int k = 0; k < K; ++k) {
                            if (store[k].n < T) store[k].alloc(tmax);
                            layers[k] = store[k];
                            const Rect& rc = rects[k];
                            warp_layer_kernel<<<dim3(cdiv(rc.x1 - rc.x0, 32), cdiv(rc.y1 - rc.y0, 8)), 256, 0, stream()>>>(
                                cv, P.dev, contrib[k], r0, c0, wt, rc, opts->angle_power, wf_floor, layers[k]);
                        }
                    }
                    check_launch("warp_layer_kernel");
                    if (opts->blending == APS_BLEND_LINEAR) {
                        if (K <= kMaxK) {
                            linear_fuse_kernel<PtrTab><<<cdiv(T, 256), 256, 0, stream()>>>(
                                make_tab(layers.data(), K), make_rtab(rects.data(), K), culled ? 1 : 0, K, wt, T, F, cov);
                            check_launch("linear_fuse_kernel");
                        } else {
                            Ws<float4*> dl(K);
                            APS_HIP(hipMemcpyAsync(dl, layers.data(), K * sizeof(float4*), hipMemcpyHostToDevice, stream()));
                            linear_fuse_kernel<float4* const*><<<cdiv(T, 256), 256, 0, stream()>>>(dl.get(), RectTab{}, 0, K, wt, T, F, cov);
                            check_launch("linear_fuse_kernel");
                            APS_HIP(hipStreamSynchronize(stream()));
                        }
                    } else {
                        // fuseTile's normalisation (:991-1006) and multiBandBlending's own (:72-85), one pass
                        normalize_weights(layers, culled ? rects.data() : nullptr, ht, wt, 1, 1, cov);
                        multiband_device(layers, culled ? rects.data() : nullptr, ht, wt, opts->pyr_levels, opts->pyr_sigma, F);
                    }
                }
            }
            paint_kernel<<<dim3(cdiv(wt, 256), ht), 256, 0, stream()>>>(
                F, cov, r0, c0, ht, wt, H, W, opts->canvas_white, out_layout, oP,
                oC.present() ? oC.get() : nullptr);
            check_launch("paint_kernel");
        }
        oP.commit();
        oC.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// gain-compensation overlap statistics (gainCompensationRKf.m:96-149,239-367)
// ------------------------------------------------------------------------------------------------
namespace aps {
// panoDirsGridTile (:369-430): 1-based canvas coordinates, no normalisation
__device__ __forceinline__ void gain_ray(const DevCanvas& cv, float xp, float yp, float d[3]) {
    if (cv.mode == APS_PROJ_CYLINDRICAL) {
        const float th = cv.o0 + xp / cv.f, hl = cv.o1 + yp / cv.f;
        d[0] = sinf(th);
        d[1] = hl;
        d[2] = cosf(th);
    } else if (cv.mode == APS_PROJ_SPHERICAL) {
        const float th = cv.o0 + xp / cv.f, ph = cv.o1 + yp / cv.f;
        const float cp = cosf(ph), sp = sinf(ph);
        d[0] = cp * sinf(th);
        d[1] = sp;
        d[2] = cp * cosf(th);
    } else {
        float rx, ry, rz;
        if (cv.mode == APS_PROJ_PLANAR) {
            rx = cv.o0 + xp / cv.f;
            ry = cv.o1 + yp / cv.f;
            rz = 1.0f;
        } else {
            const float a = cv.o0 + xp / cv.f, b = cv.o1 + yp / cv.f;
            const float r2 = a * a + b * b, den = 1.0f + r2;
            rx = 2.0f * a / den;
            ry = 2.0f * b / den;
            rz = (1.0f - r2) / den;
        }
        d[0] = (cv.Rref[0] * rx + cv.Rref[1] * ry) + cv.Rref[2] * rz;
        d[1] = (cv.Rref[3] * rx + cv.Rref[4] * ry) + cv.Rref[5] * rz;
        d[2] = (cv.Rref[6] * rx + cv.Rref[7] * ry) + cv.Rref[8] * rz;
    }
}

// projectToImage + sampleLinear(weights) > 0 + sampleLinearRGB on the raw 0..255 values (:432-579)
__device__ __forceinline__ bool gain_sample(const DevImage& im, const float d[3], float col[3]) {
    float cam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c] = fmaf(d[2], im.R[c + 6], fmaf(d[1], im.R[c + 3], d[0] * im.R[c]));
    const bool front = cam[2] > 1e-6f;
    const float u = im.fx * (cam[0] / cam[2]) + im.cx;
    const float v = im.fy * (cam[1] / cam[2]) + im.cy;
    const int w = im.w, h = im.h;
    if (!front || !isfinite(u) || !isfinite(v)) return false;
    if (!((u >= 1.0f) && (u <= (float)w) && (v >= 1.0f) && (v <= (float)h))) return false;
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    x0 = max(1, min(x0, w - 1));
    y0 = max(1, min(y0, h - 1));
    const int x1 = min(x0 + 1, w), y1 = min(y0 + 1, h);
    const float s = u - (float)x0, t = v - (float)y0;
    const float wy0 = im.wy[y0 - 1], wy1 = im.wy[y1 - 1], wx0 = im.wx[x0 - 1], wx1 = im.wx[x1 - 1];
    const float f00 = wy0 * wx0, f10 = wy0 * wx1, f01 = wy1 * wx0, f11 = wy1 * wx1;
    const float wtop = (1.0f - s) * f00 + s * f10, wbot = (1.0f - s) * f01 + s * f11;
    if (!((wtop * (1.0f - t) + wbot * t) > 0.0f)) return false;
    const uint32_t p00 = im.rgba[(size_t)(y0 - 1) * w + (x0 - 1)], p10 = im.rgba[(size_t)(y0 - 1) * w + (x1 - 1)];
    const uint32_t p01 = im.rgba[(size_t)(y1 - 1) * w + (x0 - 1)], p11 = im.rgba[(size_t)(y1 - 1) * w + (x1 - 1)];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v00 = (float)((p00 >> (8 * c)) & 255u), v10 = (float)((p10 >> (8 * c)) & 255u);
        const float v01 = (float)((p01 >> (8 * c)) & 255u), v11 = (float)((p11 >> (8 * c)) & 255u);
        const float top = (1.0f - s) * v00 + s * v10;
        const float bot = (1.0f - s) * v01 + s * v11;
        col[c] = top * (1.0f - t) + bot * t;
    }
    return true;
}

constexpr int kGainSlots = 128;  // per-workgroup pair table
constexpr int kGainMaxCover = 16;

// The pair sums of one workgroup of both gain-statistics kernels: a small LDS hash table keyed by the image pair (the points
// of a 16 x 16 patch share a handful of pairs), flushed with one double atomicAdd per touched entry; a full table sends the
// contribution straight to memory.  Outputs are n x n (x 3) column-major, entry (i, j), i < j.
struct GainPairTable {
    unsigned int key[kGainSlots];
    unsigned int cnt[kGainSlots];
    double sum[kGainSlots][6];
    __device__ __forceinline__ void init() {
        for (int e = threadIdx.x; e < kGainSlots; e += blockDim.x) {
            key[e] = 0u;
            cnt[e] = 0u;
#pragma unroll
            for (int c = 0; c < 6; ++c) sum[e][c] = 0.0;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void add(int n_img, int i, int j, const float* ci, const float* cj, double* __restrict__ Nij,
                                        double* __restrict__ sCi, double* __restrict__ sCj) {
        const unsigned int k = (unsigned int)(i * n_img + j) + 1u;
        unsigned int slot = (k * 2654435761u) >> 25;
        for (int probe = 0; probe < kGainSlots; ++probe) {
            const unsigned int old = atomicCAS(&key[slot], 0u, k);
            if (old == 0u || old == k) {
                atomicAdd(&cnt[slot], 1u);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    atomicAdd(&sum[slot][c], (double)ci[c]);
                    atomicAdd(&sum[slot][3 + c], (double)cj[c]);
                }
                return;
            }
            slot = (slot + 1) & (kGainSlots - 1);
        }
        const size_t nn = (size_t)n_img * n_img, e = (size_t)i + (size_t)n_img * j;  // table full: straight to memory
        atomicAdd(&Nij[e], 1.0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            atomicAdd(&sCi[e + nn * c], (double)ci[c]);
            atomicAdd(&sCj[e + nn * c], (double)cj[c]);
        }
    }
    __device__ __forceinline__ void flush(int n_img, double* __restrict__ Nij, double* __restrict__ sCi, double* __restrict__ sCj) {
        __syncthreads();
        const size_t nn = (size_t)n_img * n_img;
        for (int e = threadIdx.x; e < kGainSlots; e += blockDim.x) {
            const unsigned int k = key[e];
            if (!k) continue;
            const int i = (int)((k - 1u) / (unsigned int)n_img), j = (int)((k - 1u) % (unsigned int)n_img);
            const size_t o = (size_t)i + (size_t)n_img * j;
            atomicAdd(&Nij[o], (double)cnt[e]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(&sCi[o + nn * c], sum[e][c]);
                atomicAdd(&sCj[o + nn * c], sum[e][3 + c]);
            }
        }
    }
};

// One thread per sampled canvas point.  Pair sums go through a small LDS hash table per workgroup (the points of a
// 16 x 16 patch share a handful of pairs), flushed with one double atomicAdd per entry.
__global__ __launch_bounds__(256) void gain_stats_kernel(DevCanvas cv, const DevImage* __restrict__ imgs, int n_img,
                                                         int stride, int ws, int hs, double* __restrict__ Nij,
                                                         double* __restrict__ sCi, double* __restrict__ sCj) {
    __shared__ GainPairTable s_tab;
    s_tab.init();
    auto add_pair = [&](int i, int j, const float* ci, const float* cj) { s_tab.add(n_img, i, j, ci, cj, Nij, sCi, sCj); };
    const int ix = blockIdx.x * 16 + (threadIdx.x & 15), iy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (ix < ws && iy < hs) {
        float d[3];
        gain_ray(cv, (float)(1 + stride * ix), (float)(1 + stride * iy), d);
        const float dn = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
        int cov[kGainMaxCover];
        float col[kGainMaxCover][3];
        int k = 0;
        bool overflow = false;
        for (int i = 0; i < n_img; ++i) {
            const DevImage& im = imgs[i];
            const float cam2 = fmaf(d[2], im.R[8], fmaf(d[1], im.R[5], d[0] * im.R[2]));
            if (!(cam2 >= im.cmin * dn)) continue;  // outside the image's cone: cannot project inside
            float c3[3];
            if (!gain_sample(im, d, c3)) continue;
            if (k < kGainMaxCover) {
                cov[k] = i;
                col[k][0] = c3[0];
                col[k][1] = c3[1];
                col[k][2] = c3[2];
                ++k;
            } else {
                overflow = true;
            }
        }
        if (!overflow) {
            for (int a = 0; a < k; ++a)
                for (int b = a + 1; b < k; ++b) add_pair(cov[a], cov[b], col[a], col[b]);
        } else {  // more than kGainMaxCover images see this point: recompute pair by pair
            for (int i = 0; i < n_img; ++i) {
                float ci[3];
                if (!gain_sample(imgs[i], d, ci)) continue;
                for (int j = i + 1; j < n_img; ++j) {
                    float cj[3];
                    if (gain_sample(imgs[j], d, cj)) add_pair(i, j, ci, cj);
                }
            }
        }
    }
    s_tab.flush(n_img, Nij, sCi, sCj);
}
}  // namespace aps

extern "C" int aps_gain_overlap_stats(const aps_image* images, int n_img, const aps_canvas* canvas, int stride,
                                      double* n_ij, double* sum_ci, double* sum_cj) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(images && canvas && n_ij && sum_ci && sum_cj, APS_E_ARG, "NULL argument");
        APS_REQUIRE(n_img >= 1 && n_img <= 46340, APS_E_ARG, "bad image count");
        APS_REQUIRE(stride >= 1, APS_E_ARG, "overlapStride must be >= 1");
        ctx();
        PreparedImages P;
        prepare_images(images, n_img, P);
        const DevCanvas cv = make_canvas(*canvas);
        const size_t nn = (size_t)n_img * n_img;
        Out<double> oN(n_ij, nn), oI(sum_ci, 3 * nn), oJ(sum_cj, 3 * nn);
        APS_HIP(hipMemsetAsync(oN.get(), 0, nn * sizeof(double), stream()));
        APS_HIP(hipMemsetAsync(oI.get(), 0, 3 * nn * sizeof(double), stream()));
        APS_HIP(hipMemsetAsync(oJ.get(), 0, 3 * nn * sizeof(double), stream()));
        const int ws = (cv.W - 1) / stride + 1, hs = (cv.H - 1) / stride + 1;  // numel(1:stride:W), numel(1:stride:H)
        {
            Prof prof("gain_stats");
            gain_stats_kernel<<<dim3(cdiv(ws, 16), cdiv(hs, 16)), 256, 0, stream()>>>(cv, P.dev, n_img, stride, ws, hs, oN.get(),
                                                                                   oI.get(), oJ.get());
        }
        check_launch("gain_stats_kernel");
        oN.commit();
        oI.commit();
        oJ.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

// ------------------------------------------------------------------------------------------------
// gainCompensationH's overlap statistics (PP/gainCompensation/gainCompensationH.m:45-52,78-149): the planar-scan path has
// every image ALREADY warped to one canvas (Iw{k} H x W x 3 single, Ww{k} H x W single, renderPanorama.m:579-588), so the
// statistics are plain strided sums over those canvases: every ds-th row and column (1:ds:end), a point is valid for
// image k when Ww{k} > 0 and all three channels are finite, and every pair i < j that is valid there adds one count and
// the two colours (accumulated in double, as the reference's sum(..., 'double')).
// ------------------------------------------------------------------------------------------------
namespace aps {
struct WarpedSet {
    const float* const* iw;  // device array of N device pointers
    const float* const* ww;
    int n, h, w, ch, layout;  // layout APS_ROWMAJOR: h x w x ch interleaved rows (C / numpy); APS_COLMAJOR: MATLAB h x w x ch
};

__device__ __forceinline__ size_t warped_at(const WarpedSet& S, int y, int x, int c, int nch) {
    return S.layout == APS_ROWMAJOR ? ((size_t)y * S.w + x) * nch + c : (size_t)y + (size_t)S.h * ((size_t)x + (size_t)S.w * c);
}

// One thread per sampled canvas point; the pair table of gain_stats_kernel (an LDS hash per workgroup, one double
// atomicAdd per touched pair at the end).
__global__ __launch_bounds__(256) void gain_stats_warped_kernel(WarpedSet S, int ds, int ws, int hs, double* __restrict__ Nij,
                                                                double* __restrict__ sCi, double* __restrict__ sCj) {
    __shared__ GainPairTable s_tab;
    s_tab.init();
    const int n_img = S.n;
    auto add_pair = [&](int i, int j, const float* ci, const float* cj) { s_tab.add(n_img, i, j, ci, cj, Nij, sCi, sCj); };
    const int ix = blockIdx.x * 16 + (threadIdx.x & 15), iy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (ix < ws && iy < hs) {
        const int x = ix * ds, y = iy * ds;
        auto sample = [&](int k, float* c3) {
            if (!(S.ww[k][warped_at(S, y, x, 0, 1)] > 0.f)) return false;
            bool fin = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                c3[c] = S.iw[k][warped_at(S, y, x, c < S.ch ? c : S.ch - 1, S.ch)];
                fin = fin && isfinite(c3[c]);
            }
            return fin;
        };
        int cov[kGainMaxCover];
        float col[kGainMaxCover][3];
        int k = 0;
        bool overflow = false;
        for (int i = 0; i < n_img; ++i) {
            float c3[3];
            if (!sample(i, c3)) continue;
            if (k < kGainMaxCover) {
                cov[k] = i;
                col[k][0] = c3[0];
                col[k][1] = c3[1];
                col[k][2] = c3[2];
                ++k;
            } else {
                overflow = true;
            }
        }
        if (!overflow) {
            for (int a = 0; a < k; ++a)
                for (int b = a + 1; b < k; ++b) add_pair(cov[a], cov[b], col[a], col[b]);
        } else {
            for (int i = 0; i < n_img; ++i) {
                float ci[3];
                if (!sample(i, ci)) continue;
                for (int j = i + 1; j < n_img; ++j) {
                    float cj[3];
                    if (sample(j, cj)) add_pair(i, j, ci, cj);
                }
            }
        }
    }
    s_tab.flush(n_img, Nij, sCi, sCj);
}
}  // namespace aps

extern "C" int aps_gain_overlap_stats_warped(const float* const* iw, const float* const* ww, int n_img, int64_t height, int64_t width,
                                             int channels, int layout, int downsample, double* n_ij, double* sum_ci,
                                             double* sum_cj) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(iw && ww && n_ij && sum_ci && sum_cj, APS_E_ARG, "NULL argument");
        APS_REQUIRE(n_img >= 1 && n_img <= 46340, APS_E_ARG, "bad image count");
        APS_REQUIRE(height >= 1 && width >= 1 && height * width < ((int64_t)1 << 31), APS_E_DIM, "bad canvas size");
        APS_REQUIRE(channels == 1 || channels == 3, APS_E_DIM, "channels must be 1 or 3");
        APS_REQUIRE(layout == APS_ROWMAJOR || layout == APS_COLMAJOR, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(downsample >= 1, APS_E_ARG, "overlapDownsample must be >= 1");
        ctx();
        const size_t px = (size_t)height * width;
        std::vector<std::unique_ptr<In<float>>> keep;
        std::vector<const float*> hi(n_img), hw(n_img);
        for (int k = 0; k < n_img; ++k) {
            APS_REQUIRE(iw[k] && ww[k], APS_E_ARG, "NULL canvas %d", k);
            keep.emplace_back(new In<float>(iw[k], px * channels));
            hi[k] = keep.back()->get();
            keep.emplace_back(new In<float>(ww[k], px));
            hw[k] = keep.back()->get();
        }
        Ws<const float*> di(n_img), dw(n_img);
        APS_HIP(hipMemcpyAsync(di, hi.data(), n_img * sizeof(const float*), hipMemcpyHostToDevice, stream()));
        APS_HIP(hipMemcpyAsync(dw, hw.data(), n_img * sizeof(const float*), hipMemcpyHostToDevice, stream()));
        const size_t nn = (size_t)n_img * n_img;
        Out<double> oN(n_ij, nn), oI(sum_ci, 3 * nn), oJ(sum_cj, 3 * nn);
        APS_HIP(hipMemsetAsync(oN.get(), 0, nn * sizeof(double), stream()));
        APS_HIP(hipMemsetAsync(oI.get(), 0, 3 * nn * sizeof(double), stream()));
        APS_HIP(hipMemsetAsync(oJ.get(), 0, 3 * nn * sizeof(double), stream()));
        const int ws = (int)((width - 1) / downsample + 1), hs = (int)((height - 1) / downsample + 1);  // numel(1:ds:end)
        WarpedSet S{di, dw, n_img, (int)height, (int)width, channels, layout};
        {
            Prof prof("gain_stats_warped");
            gain_stats_warped_kernel<<<dim3(cdiv(ws, 16), cdiv(hs, 16)), 256, 0, stream()>>>(S, downsample, ws, hs, oN.get(), oI.get(),
                                                                                          oJ.get());
        }
        check_launch("gain_stats_warped_kernel");
        oN.commit();
        oI.commit();
        oJ.commit();
        APS_HIP(hipStreamSynchronize(stream()));  // the host pointer arrays and the staged inputs must outlive the launch
    });
}

// ------------------------------------------------------------------------------------------------
// imresize on uint8 images (resizeImagesToLimits.m:49-106) -- see oracle/render_oracle.c
// ------------------------------------------------------------------------------------------------
namespace aps {
__device__ __forceinline__ double cubic_kernel(double x) {  // Keys, a = -0.5
    const double a = fabs(x), a2 = a * a, a3 = a2 * a;
    if (a <= 1.0) return (1.5 * a3 - 2.5 * a2) + 1.0;
    if (a <= 2.0) return ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0;
    return 0.0;
}
__device__ __forceinline__ size_t u8_index(int layout, int h, int w, int C, int y, int x, int c) {
    return layout == APS_IMG_U8_HWC ? ((size_t)y * w + x) * C + c : (size_t)y + (size_t)h * ((size_t)x + (size_t)w * c);
}
// one output sample per thread along `dim` (0: rows, 1: columns); all channels
__global__ void imresize_u8_dim_kernel(const uint8_t* __restrict__ in, int h, int w, int C, int layout, int dim,
                                       int out_len, double scale, int bicubic, uint8_t* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;  // position along the other dimension
    const int o = blockIdx.y;                             // output sample along `dim`
    const int in_len = dim == 0 ? h : w, other = dim == 0 ? w : h;
    if (q >= other) return;
    const int oh = dim == 0 ? out_len : h, ow = dim == 0 ? w : out_len;
    const double kw0 = bicubic ? 4.0 : 2.0;
    const double kw = scale < 1.0 ? kw0 / scale : kw0;
    const double u = (double)(o + 1) / scale + 0.5 * (1.0 - 1.0 / scale);
    const int left = (int)floor(u - kw / 2.0);
    int P = (int)ceil(kw) + 2;
    if (P > 64) P = 64;
    double s = 0;
    for (int t = 0; t < P; ++t) {
        const double dx = u - (double)(left + t);
        const double arg = scale < 1.0 ? scale * dx : dx;
        double v = bicubic ? cubic_kernel(arg) : (fabs(arg) < 1.0 ? 1.0 - fabs(arg) : 0.0);
        if (scale < 1.0) v = scale * v;
        s += v;
    }
    for (int c = 0; c < C; ++c) {
        double acc = 0;
        for (int t = 0; t < P; ++t) {
            const double dx = u - (double)(left + t);
            const double arg = scale < 1.0 ? scale * dx : dx;
            double v = bicubic ? cubic_kernel(arg) : (fabs(arg) < 1.0 ? 1.0 - fabs(arg) : 0.0);
            if (scale < 1.0) v = scale * v;
            const double wt = v / s;
            const int idx = min(max(left + t, 1), in_len) - 1;
            const uint8_t px = dim == 0 ? in[u8_index(layout, h, w, C, idx, q, c)] : in[u8_index(layout, h, w, C, q, idx, c)];
            acc = acc + wt * (double)px;
        }
        double r = acc < 0 ? -floor(-acc + 0.5) : floor(acc + 0.5);
        r = r > 0 ? r : 0;
        r = r > 255 ? 255 : r;
        const size_t oi = dim == 0 ? u8_index(layout, oh, ow, C, o, q, c) : u8_index(layout, oh, ow, C, q, o, c);
        out[oi] = (uint8_t)r;
    }
}
}  // namespace aps

extern "C" int aps_imresize_u8(const uint8_t* img, int h, int w, int c, int layout, int oh, int ow, double scale_r,
                               double scale_c, int method, uint8_t* out) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(img && out, APS_E_ARG, "NULL argument");
        APS_REQUIRE(h > 0 && w > 0 && oh > 0 && ow > 0 && c >= 1 && c <= 4, APS_E_DIM, "bad image size");
        APS_REQUIRE(layout == APS_IMG_U8_HWC || layout == APS_IMG_U8_MATLAB, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(method == APS_RESIZE_BICUBIC || method == APS_RESIZE_BILINEAR, APS_E_ARG, "unknown method");
        APS_REQUIRE(scale_r > 0 && scale_c > 0 && std::isfinite(scale_r) && std::isfinite(scale_c), APS_E_ARG, "bad scale");
        APS_REQUIRE(4.0 / std::min(1.0, std::min(scale_r, scale_c)) + 2 <= 64, APS_E_ARG, "scale below 1/15 is not built");
        ctx();
        In<uint8_t> di(img, (size_t)h * w * c);
        Out<uint8_t> dout(out, (size_t)oh * ow * c);
        const int bic = method == APS_RESIZE_BICUBIC;
        Prof prof("imresize_u8");
        if (scale_r <= scale_c) {  // the dimension with the smaller scale first (ties: rows)
            Ws<uint8_t> tmp((size_t)oh * w * c);
            imresize_u8_dim_kernel<<<dim3(cdiv(w, 128), oh), 128, 0, stream()>>>(di, h, w, c, layout, 0, oh, scale_r, bic, tmp);
            imresize_u8_dim_kernel<<<dim3(cdiv(oh, 128), ow), 128, 0, stream()>>>(tmp, oh, w, c, layout, 1, ow, scale_c, bic, dout.get());
            check_launch("imresize_u8_dim_kernel");
            dout.commit();
            APS_HIP(hipStreamSynchronize(stream()));
        } else {
            Ws<uint8_t> tmp((size_t)h * ow * c);
            imresize_u8_dim_kernel<<<dim3(cdiv(h, 128), ow), 128, 0, stream()>>>(di, h, w, c, layout, 1, ow, scale_c, bic, tmp);
            imresize_u8_dim_kernel<<<dim3(cdiv(ow, 128), oh), 128, 0, stream()>>>(tmp, h, ow, c, layout, 0, oh, scale_r, bic, dout.get());
            check_launch("imresize_u8_dim_kernel");
            dout.commit();
            APS_HIP(hipStreamSynchronize(stream()));
        }
    });
}

static void make_hwarp(const double* H, HWarp& hw) {
    double h[9];
    for (int e = 0; e < 9; ++e) h[e] = H[8] != 0 ? H[e] / H[8] : H[e];
#define HH(r, c) h[(r) + 3 * (c)]
    hw.A[0] = HH(1, 1) * HH(2, 2) - HH(1, 2) * HH(2, 1);
    hw.A[3] = HH(0, 2) * HH(2, 1) - HH(0, 1) * HH(2, 2);
    hw.A[6] = HH(0, 1) * HH(1, 2) - HH(0, 2) * HH(1, 1);
    hw.A[1] = HH(1, 2) * HH(2, 0) - HH(1, 0) * HH(2, 2);
    hw.A[4] = HH(0, 0) * HH(2, 2) - HH(0, 2) * HH(2, 0);
    hw.A[7] = HH(0, 2) * HH(1, 0) - HH(0, 0) * HH(1, 2);
    hw.A[2] = HH(1, 0) * HH(2, 1) - HH(1, 1) * HH(2, 0);
    hw.A[5] = HH(0, 1) * HH(2, 0) - HH(0, 0) * HH(2, 1);
    hw.A[8] = HH(0, 0) * HH(1, 1) - HH(0, 1) * HH(1, 0);
    hw.det = (HH(0, 0) * hw.A[0] + HH(0, 1) * hw.A[1]) + HH(0, 2) * hw.A[2];
#undef HH
}

template <class T>
static int image_warp_impl(const T* in, int in_h, int in_w, int c, const double* H, int out_h, int out_w,
                           double x0, double y0, double sx, double sy, T fill, int method, T* out) {
    return guarded([&] {
        APS_REQUIRE(in && H && out, APS_E_ARG, "NULL argument");
        APS_REQUIRE(in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && c >= 1 && c <= 4, APS_E_DIM, "bad dimensions");
        APS_REQUIRE(method == APS_WARP_NEAREST || method == APS_WARP_BILINEAR || method == APS_WARP_BICUBIC, APS_E_ARG,
                    "unknown interpolation method");
        ctx();
        HWarp hw;
        make_hwarp(H, hw);
        In<T> di(in, (size_t)in_h * in_w * c);
        Out<T> oo(out, (size_t)out_h * out_w * c);
        const dim3 grid(cdiv(out_w, 256), out_h);
        if (method == APS_WARP_NEAREST)
            image_warp_h_kernel<T, APS_WARP_NEAREST><<<grid, 256, 0, stream()>>>(di, in_h, in_w, c, hw, out_h, out_w, x0, y0, sx, sy, fill, oo);
        else if (method == APS_WARP_BICUBIC)
            image_warp_h_kernel<T, APS_WARP_BICUBIC><<<grid, 256, 0, stream()>>>(di, in_h, in_w, c, hw, out_h, out_w, x0, y0, sx, sy, fill, oo);
        else
            image_warp_h_kernel<T, APS_WARP_BILINEAR><<<grid, 256, 0, stream()>>>(di, in_h, in_w, c, hw, out_h, out_w, x0, y0, sx, sy, fill, oo);
        check_launch("image_warp_h_kernel");
        oo.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}

extern "C" {

int aps_image_warp_h_u8(const uint8_t* in, int in_h, int in_w, int c, const double* H, int out_h,
                        int out_w, double x0, double y0, double sx, double sy, uint8_t fill,
                        uint8_t* out) {
    return image_warp_impl<uint8_t>(in, in_h, in_w, c, H, out_h, out_w, x0, y0, sx, sy, fill, APS_WARP_BILINEAR, out);
}

int aps_image_warp_u8(const uint8_t* in, int in_h, int in_w, int c, const double* H, int out_h, int out_w, double x0,
                      double y0, double sx, double sy, uint8_t fill, int method, uint8_t* out) {
    return image_warp_impl<uint8_t>(in, in_h, in_w, c, H, out_h, out_w, x0, y0, sx, sy, fill, method, out);
}

int aps_image_warp_f32(const float* in, int in_h, int in_w, int c, const double* H, int out_h, int out_w, double x0,
                       double y0, double sx, double sy, float fill, int method, float* out) {
    return image_warp_impl<float>(in, in_h, in_w, c, H, out_h, out_w, x0, y0, sx, sy, fill, method, out);
}

int aps_image_warp_h_f32(const float* in, int in_h, int in_w, int c, const double* H, int out_h,
                         int out_w, double x0, double y0, double sx, double sy, float fill, float* out) {
    return image_warp_impl<float>(in, in_h, in_w, c, H, out_h, out_w, x0, y0, sx, sy, fill, APS_WARP_BILINEAR, out);
}

}  // extern "C"
