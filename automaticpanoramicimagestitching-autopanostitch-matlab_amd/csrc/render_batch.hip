// render_batch.hip — the multiband render path for ALL tiles of a canvas at once (renderPanorama.m:342-425 tile
// loop, fuseTile :825-1060, sampleOneTile :1063-1146, multiBandBlending.m:45-171), laid out for MI355X:
//
//   * level-major over the whole canvas: every pyramid step is ONE launch covering all (tile, layer) footprints
//     (13 launches per panorama with 5 bands instead of ~22 per tile); work items are 32 x 8 output blocks found
//     through a prefix table, XCD-contiguous so that neighbouring blocks (which share halos) share an L2;
//   * the inverse warp is fused into pyramid level 0: the level-0 layer (16 B per covered pixel) is never written.
//     The blur + downsample pass warps its haloed patch straight into LDS, and the Laplacian/collapse pass of
//     level 0 re-evaluates the same sample (bit-identical: same code on the same inputs) instead of re-reading it;
//   * the fuseTile + multiBandBlending weight normalisations (two sums over the tile's layers, in layer order)
//     are produced by the coverage pass as two floats per canvas pixel and applied inside the warp;
//   * Laplacian accumulation over the layers, the collapse step and (at level 0) the uint8 paint are one pass
//     per level, coarse to fine: Num_l is never stored;
//   * imresize tap positions/weights come from small per-length tables built once per call by the same f64
//     formula the per-tile kernels evaluate per pixel;
//   * pyramid levels are stored compactly over each layer's footprint rectangle (the layer is exactly zero
//     outside it, see struct Rect).
//
// The arithmetic (operation order, roundings) of every value is that of render.hip's per-tile kernels, which
// are the restatement checked against oracle/render_oracle.c: tests/test_render_gpu.py compares both paths
// bit for bit (APS_RENDER_LEGACY=1 selects the per-tile path).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <future>
#include <map>
#include <thread>
#include <vector>

#include <chrono>

#include "render_dev.h"

namespace aps {

namespace {

constexpr int kML = 8;        // pyramid levels supported by the batched path
constexpr int kTD = 6;        // non-zero taps kept per output sample of a shrink (scale in [1/3, 1/2]: <= 6)
constexpr int kTU = 2;        // ... of an enlargement (scale >= 2: <= 2)
constexpr int kNT = 512;                             // threads of the shrink kernels
constexpr int kBW = 32, kBH = 16;                    // their output block (one output pixel per thread)
constexpr int kBC = 2 * kBW + 8, kBR = 2 * kBH + 8;  // blurred patch a block of the shrink can need
constexpr int kUW = 32, kUH = 8;                     // output block of the collapse kernels (256 threads)

struct RwTile {
    int r0, c0, ht, wt;
    int e0, ne;  // entries [e0, e0 + ne), ascending image index
    int nl;      // pyramid levels of this tile: clamp(levels, 1, floor(log2(min(ht, wt))))
    int rf_down, rf_up;  // bit l: imresize(level l -> l+1) / (l+1 -> l) resizes rows first
    long long plane;     // pixel offset of the tile in the canvas-sized planes (norm, cov), pitch wt
    long long f[kML];    // float4 offset of F_l (l >= 1), pitch lw[l]
    int lh[kML], lw[kML];
    int tdr[kML], tdc[kML];  // tap-table offsets (in output samples) of the shrink l -> l+1: rows, columns
    int tur[kML], tuc[kML];  // ... of the enlargement l+1 -> l
};

struct RwEntry {
    int tile, img;
    Rect g[kML];        // footprint of G_l in level-l tile coordinates
    long long off[kML];  // float4 offset of the compact G_l store, pitch g[l].x1 - g[l].x0
};

struct ColTrig {
    float s, c;  // sin/cos of theta (cylindrical, spherical)
};
struct RowTrig {
    float s, c;  // spherical: sin/cos of phi; cylindrical: s = height
};

struct RwArgs {
    const RwTile* tiles;
    const RwEntry* entries;
    int n_tiles, n_entries;
    const DevImage* imgs;
    DevCanvas cv;
    float angle_pow;
    Taps tp;
    const int* td_idx;    // [sample][kTD] clamped 0-based source index
    const float* td_w;    // [sample][kTD]
    const int* tu_idx;    // [sample][kTU]
    const float* tu_w;
    float4* G;            // compact pyramid store
    float4* F;            // collapse planes, levels >= 1
    uint8_t* cov;         // per canvas pixel: any(w > 0)
    int* status;          // sticky error flag raised by a kernel whose assumptions do not hold
    uint8_t* pano;
    uint8_t* covered;
    int H, W, out_layout, white;
    const int* cd_s0;     // fused shrink (rw_down_fused_kernel): first source index per output sample ...
    const float* cd_w;    // ... and the weights of the consecutive source samples from there, [sample][NCP]
    const ColTrig* ct;  // per canvas column: sin/cos of theta (cylindrical, spherical), rw_trig_kernel
    const RowTrig* rt;  // per canvas row: sin/cos of phi (spherical) or the height (cylindrical)
};

// largest s with ptr[s] <= bid (ptr nondecreasing, ptr[0] = 0, n segments); every wave evaluates it for itself
__device__ __forceinline__ int find_segment(const int* __restrict__ ptr, int n, int bid) {
    const int lane = threadIdx.x & 63;
    int cnt = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + lane;
        const bool le = i < n && ptr[i] <= bid;
        cnt += __popcll(__ballot(le));
    }
    return __builtin_amdgcn_readfirstlane(cnt - 1);  // wave-uniform by construction: let the compiler know
}

// The entries of a tile whose level-l footprint meets a block, as bit masks: the lanes of a wave test 64 entries at a
// time, and the wave then walks the set bits in each of its passes instead of re-testing every entry of the tile (the
// uniform per-entry tests were most of rw_warp's scalar work: ~560 SALU instructions and ~50 scalar loads per wave
// against ~700 VALU instructions).  Entries are visited in ascending order, as by the plain loop.  Must be called by
// all 64 lanes of the wave (before any early return).
constexpr int kLM = 2;  // mask words kept; tiles with more entries than that take the plain loop
struct LayerSet {
    unsigned long long m[kLM];
    bool listed;
};
__device__ __forceinline__ bool rects_meet(const Rect& g, int x0, int y0, int x1, int y1) {
    return max(x0, g.x0) < min(x1, g.x1) && max(y0, g.y0) < min(y1, g.y1);
}
__device__ __forceinline__ LayerSet block_layers(const RwArgs& A, const RwTile& T, int l, int x0, int y0, int x1, int y1) {
    LayerSet s;
    s.listed = T.ne <= 64 * kLM;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < kLM; ++c) {
        const int k = c * 64 + lane;
        bool hit = false;
        if (s.listed && k < T.ne) hit = rects_meet(A.entries[T.e0 + k].g[l], x0, y0, x1, y1);
        s.m[c] = __ballot(hit);
    }
    return s;
}
template <class F>
__device__ __forceinline__ void for_block_layers(const LayerSet& s, const RwArgs& A, const RwTile& T, int l, int x0, int y0,
                                                 int x1, int y1, F fn) {
    if (s.listed) {
#pragma unroll
        for (int c = 0; c < kLM; ++c)
            for (unsigned long long m = s.m[c]; m; m &= m - 1) fn(c * 64 + __ffsll((long long)m) - 1);
    } else {
        for (int k = 0; k < T.ne; ++k)
            if (rects_meet(A.entries[T.e0 + k].g[l], x0, y0, x1, y1)) fn(k);
    }
}

// blockIdx -> work item: consecutive work items stay on one XCD (workgroups are dealt round-robin to the 8 XCDs)
__device__ __forceinline__ int xcd_contiguous_id(int n_items) {
    const int per = (n_items + 7) >> 3;
    return (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
}

// ------------------------------------------------------------------------------------------------
// tap tables
// ------------------------------------------------------------------------------------------------
struct TapJob {
    int in_len, out_len, T, off;  // off in output samples
};

// One thread per output sample: the non-zero taps of resize_taps, in order, with their clamped source index.
__global__ void rw_taps_kernel(const TapJob* __restrict__ jobs, int* __restrict__ d_idx, float* __restrict__ d_w,
                               int* __restrict__ u_idx, float* __restrict__ u_w, int* __restrict__ status) {
    const TapJob j = jobs[blockIdx.y];
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= j.out_len) return;
    int left;
    float wts[12];
    const int P = resize_taps(j.in_len, j.out_len, o, left, wts);
    int* ti = j.T == kTD ? d_idx : u_idx;
    float* tw = j.T == kTD ? d_w : u_w;
    const size_t base = (size_t)(j.off + o) * j.T;
    int n = 0, lastidx = min(max(left, 1), j.in_len) - 1;
    for (int t = 0; t < P; ++t) {
        if (wts[t] == 0.f) continue;  // a zero tap adds 0 * v: nothing (finite data)
        lastidx = min(max(left + t, 1), j.in_len) - 1;
        if (n < j.T) {
            ti[base + n] = lastidx;
            tw[base + n] = wts[t];
        }
        ++n;
    }
    if (n > j.T) atomicOr(status, 1);
    for (; n < j.T; ++n) {
        ti[base + n] = lastidx;
        tw[base + n] = 0.f;
    }
}

// The shrink G_{l+1} = imresize(imgaussfilt(G_l)) as ONE separable filter per axis: per output sample the weights of the
// consecutive source samples it depends on, sum over (resize tap t, Gaussian tap k) of w_t g_k at source index
// clamp(clamp(left + t) + k - R) (the blur's replicate padding at the level border and the resize's clamped taps merge
// into the border sample).  Products and sums in f64, stored as f32.  This is the blur and the resize of
// multiBandBlending.m:112-135 in exact arithmetic; it differs from the two-step evaluation by the rounding of the
// intermediate blurred plane only, which is what the render's stated tolerance is for (the per-tile path and
// APS_RENDER_EXACT=1 keep the two-step form bit for bit).
constexpr int kNCPMax = 16;  // kTD + 2 * 4 = 14 weights, padded to whole 16-byte groups
__global__ void rw_ctaps_kernel(const TapJob* __restrict__ jobs, Taps tp, int ncp, int* __restrict__ c_s0,
                                float* __restrict__ c_w, int* __restrict__ status) {
    const TapJob j = jobs[blockIdx.y];
    if (j.T != kTD) return;  // enlargements keep their two-tap tables
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= j.out_len) return;
    int left;
    float wts[12];
    const int P = resize_taps(j.in_len, j.out_len, o, left, wts);
    const int R = tp.r, nc = kTD + 2 * R;
    int lo = j.in_len;
    for (int t = 0; t < P; ++t) {
        if (wts[t] == 0.f) continue;
        const int c = min(max(left + t, 1), j.in_len) - 1;
        lo = min(lo, max(c - R, 0));
    }
    double acc[kNCPMax];
    for (int n = 0; n < kNCPMax; ++n) acc[n] = 0.0;
    bool ok = true;
    for (int t = 0; t < P; ++t) {
        if (wts[t] == 0.f) continue;
        const int c = min(max(left + t, 1), j.in_len) - 1;
        for (int k = 0; k <= 2 * R; ++k) {
            const int i = min(max(c + k - R, 0), j.in_len - 1) - lo;
            if (i < 0 || i >= nc)
                ok = false;
            else
                acc[i] += (double)wts[t] * (double)tp.k[k];
        }
    }
    if (!ok) atomicOr(status, 4);
    c_s0[j.off + o] = lo;
    for (int n = 0; n < ncp; ++n) c_w[(size_t)(j.off + o) * ncp + n] = n < nc ? (float)acc[n] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// sampling with per-block trig tables and a u8 -> [0,1] table (same values as canvas_ray / sample_one)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ ColTrig col_trig(const DevCanvas& cv, float xp) {
    ColTrig t;
    const float th = cv.o0 + xp / cv.f;
    t.s = sinf(th);
    t.c = cosf(th);
    return t;
}
__device__ __forceinline__ RowTrig row_trig(const DevCanvas& cv, float yp) {
    RowTrig t;
    const float ph = cv.o1 + yp / cv.f;
    if (cv.mode == APS_PROJ_SPHERICAL) {
        t.c = cosf(ph);
        t.s = sinf(ph);
    } else {
        t.s = ph;
        t.c = 1.0f;
    }
    return t;
}

// The transcendental parts of every canvas column / row, once per call (W + H entries): the warp's blocks used to
// evaluate sinf / cosf for their 32 columns and 8 rows themselves, behind a barrier - ~60 VALU instructions per wave
// averaged over the kernel, for values that depend on the canvas coordinate alone.  Same functions, same bits.
__global__ void rw_trig_kernel(DevCanvas cv, ColTrig* __restrict__ ct, RowTrig* __restrict__ rt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cv.W) ct[i] = col_trig(cv, (float)i);
    if (i < cv.H) rt[i] = row_trig(cv, (float)i);
}

// canvas_ray with the transcendental parts taken from the tables (cylindrical / spherical) or evaluated in
// place (planar / stereographic have none)
__device__ __forceinline__ void ray_from_tables(const DevCanvas& cv, const ColTrig& ct, const RowTrig& rt, float xp,
                                                float yp, float d[3]) {
    if (cv.mode == APS_PROJ_CYLINDRICAL || cv.mode == APS_PROJ_SPHERICAL) {
        float x, y, z;
        if (cv.mode == APS_PROJ_CYLINDRICAL) {
            x = ct.s;
            y = rt.s;
            z = ct.c;
        } else {
            x = rt.c * ct.s;
            y = rt.s;
            z = rt.c * ct.c;
        }
        float n = sqrtf((x * x + y * y) + z * z);
        if (!(n > 1e-8f)) n = 1e-8f;
        d[0] = x / n;
        d[1] = y / n;
        d[2] = z / n;
    } else {
        canvas_ray(cv, xp, yp, d);
    }
}

// sample_one with (float)b / 255.0f looked up; returns (r, g, b, Wang * Wf) with zeros outside the mask
__device__ __forceinline__ float4 sample_lut(const DevImage& im, const float d[3], float angle_pow,
                                             const float* __restrict__ u8f) {
    float u, v, wa;
    if (!project(im, d, angle_pow, u, v, wa)) return make_float4(0.f, 0.f, 0.f, 0.f);
    const int w = im.w, h = im.h;
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    x0 = max(1, min(x0, w - 1));
    y0 = max(1, min(y0, h - 1));
    const int x1 = min(x0 + 1, w), y1 = min(y0 + 1, h);
    const float s = u - (float)x0, t = v - (float)y0;
    const uint32_t* __restrict__ r0p = im.rgba + (size_t)(y0 - 1) * w;
    const uint32_t* __restrict__ r1p = im.rgba + (size_t)(y1 - 1) * w;
    const uint32_t p00 = r0p[x0 - 1], p10 = r0p[x1 - 1], p01 = r1p[x0 - 1], p11 = r1p[x1 - 1];
    const float wy0 = im.wy[y0 - 1], wy1 = im.wy[y1 - 1], wx0 = im.wx[x0 - 1], wx1 = im.wx[x1 - 1];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = im.gain[c];
        const float v00 = u8f[(p00 >> (8 * c)) & 255u] * g;
        const float v10 = u8f[(p10 >> (8 * c)) & 255u] * g;
        const float v01 = u8f[(p01 >> (8 * c)) & 255u] * g;
        const float v11 = u8f[(p11 >> (8 * c)) & 255u] * g;
        const float top = (1.0f - s) * v00 + s * v10;
        const float bot = (1.0f - s) * v01 + s * v11;
        o[c] = top * (1.0f - t) + bot * t;
    }
    const float f00 = wy0 * wx0, f10 = wy0 * wx1, f01 = wy1 * wx0, f11 = wy1 * wx1;
    const float top = (1.0f - s) * f00 + s * f10;
    const float bot = (1.0f - s) * f01 + s * f11;
    const float wf = top * (1.0f - t) + bot * t;
    return make_float4(o[0], o[1], o[2], wa * wf);
}

// sample_lut in two halves, so that a thread can put the gathers of several samples in flight before it consumes
// any of them: sample_issue projects and loads (from a safe address when the ray misses the image), sample_finish
// interpolates.  Same operations on the same values as sample_lut.
struct SampleLoads {
    uint32_t p00, p10, p01, p11;
    float wy0, wy1, wx0, wx1;
    float s, t, wa;
    bool m;
};
__device__ __forceinline__ void sample_issue(const DevImage& im, const float d[3], float angle_pow, bool active,
                                             SampleLoads& L) {
    float u = 1.0f, v = 1.0f, wa = 0.0f;
    L.m = active && project(im, d, angle_pow, u, v, wa);
    if (!L.m) {
        u = 1.0f;
        v = 1.0f;
    }
    const int w = im.w, h = im.h;
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    x0 = max(1, min(x0, w - 1));
    y0 = max(1, min(y0, h - 1));
    const int x1 = min(x0 + 1, w), y1 = min(y0 + 1, h);
    L.s = u - (float)x0;
    L.t = v - (float)y0;
    L.wa = wa;
    const uint32_t* __restrict__ r0p = im.rgba + (size_t)(y0 - 1) * w;
    const uint32_t* __restrict__ r1p = im.rgba + (size_t)(y1 - 1) * w;
    L.p00 = r0p[x0 - 1];
    L.p10 = r0p[x1 - 1];
    L.p01 = r1p[x0 - 1];
    L.p11 = r1p[x1 - 1];
    L.wy0 = im.wy[y0 - 1];
    L.wy1 = im.wy[y1 - 1];
    L.wx0 = im.wx[x0 - 1];
    L.wx1 = im.wx[x1 - 1];
}
__device__ __forceinline__ float4 sample_finish(const DevImage& im, const SampleLoads& L, const float* __restrict__ u8f) {
    const float s = L.s, t = L.t;
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float g = im.gain[c];
        const float v00 = u8f[(L.p00 >> (8 * c)) & 255u] * g;
        const float v10 = u8f[(L.p10 >> (8 * c)) & 255u] * g;
        const float v01 = u8f[(L.p01 >> (8 * c)) & 255u] * g;
        const float v11 = u8f[(L.p11 >> (8 * c)) & 255u] * g;
        const float top = (1.0f - s) * v00 + s * v10;
        const float bot = (1.0f - s) * v01 + s * v11;
        o[c] = top * (1.0f - t) + bot * t;
    }
    const float f00 = L.wy0 * L.wx0, f10 = L.wy0 * L.wx1, f01 = L.wy1 * L.wx0, f11 = L.wy1 * L.wx1;
    const float top = (1.0f - s) * f00 + s * f10;
    const float bot = (1.0f - s) * f01 + s * f11;
    const float wf = top * (1.0f - t) + bot * t;
    return L.m ? make_float4(o[0], o[1], o[2], L.wa * wf) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- the fast sampler of rw_warp_kernel ---------------------------------------------------------------------------
// Same projection, mask, taps and weights as sample_lut (renderPanorama.m:1089-1131), evaluated for speed instead of
// for bit-identity with the per-tile path: the contract for warped pixels is a stated tolerance against the oracle
// (tests: <= 2 grey levels, >= 99.95 % within one, coverage flips <= 1e-4), and the exact IEEE divisions, the u8 -> float
// table and the 4-tap tent products were two thirds of the kernel's 680 VALU instructions per pixel.
//   * ONE reciprocal of cam_z (v_rcp_f32 + one Newton step, < 1 ulp) serves u and v;
//   * the clamps make x1 = x0 + 1 and y1 = y0 + 1 always, so the taps of a row are two adjacent dwords;
//   * the bilinear form runs on the raw bytes (v_cvt_f32_ubyteN) as three lerps per channel, scaled once by gain / 255;
//   * the tent weight is separable: bilinear(wy (x) wx) = lerp(wy) * lerp(wx).
__device__ __forceinline__ float fast_rcp(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = fmaf(-x, r, 1.0f);
    return fmaf(r, e, r);
}
__device__ __forceinline__ float lerp1(float a, float b, float s) { return fmaf(s, b - a, a); }

// The image, its tent tables and its constants as the warp's layer loop wants them: buffer resources (uniform, 4 SGPRs
// each), so that a tap pair is ONE 8-byte buffer load at a 32-bit offset instead of two flat loads behind 64-bit
// address arithmetic.  Needs w, h >= 2 (then x1 = x0 + 1, y1 = y0 + 1 always); the caller sends smaller images through
// sample_lut.
struct FastImage {
    __amdgpu_buffer_rsrc_t rgba, wx, wy;
    int w, h, row_bytes;
};
__device__ __forceinline__ FastImage fast_image(const DevImage& im) {
    FastImage f;
    f.w = im.w;
    f.h = im.h;
    f.row_bytes = im.w * 4;
    f.rgba = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(im.rgba), 0, im.w * im.h * 4, 0x00020000);
    f.wx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(im.wx), 0, im.w * 4, 0x00020000);
    f.wy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(im.wy), 0, im.h * 4, 0x00020000);
    return f;
}
template <class T2>
__device__ __forceinline__ T2 ld_pair(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    T2 r;  // (copied whole: member access on the builtin's vector type is narrowed by this compiler, see ld_foot)
    static_assert(sizeof v == sizeof r, "b64");
    __builtin_memcpy(&r, &v, sizeof r);
    return r;
}

// exact_ray(de): fills de with the pixel's ray as ray_from_tables computes it (called on the rare path only).
template <bool BAND, bool BUF, class ExactRay>
__device__ __forceinline__ float4 sample_fast(const DevImage& im, const FastImage& fi, const float d[3], float angle_pow,
                                              ExactRay exact_ray) {
    const float cam0 = fmaf(d[2], im.R[6], fmaf(d[1], im.R[3], d[0] * im.R[0]));
    const float cam1 = fmaf(d[2], im.R[7], fmaf(d[1], im.R[4], d[0] * im.R[1]));
    float cam2 = fmaf(d[2], im.R[8], fmaf(d[1], im.R[5], d[0] * im.R[2]));
    const int w = im.w, h = im.h;
    // WHICH pixels an image covers must not depend on the fast arithmetic: a pixel that flips at the rim of the panorama
    // carries a normalised weight of one (it is the only layer there) and shifts the blend of its neighbourhood by a few
    // grey levels.  A ray that lands within `band` of the image border or of the cam_z threshold - a band a few times
    // wider than the fast path's error - is therefore re-projected exactly as sample_lut / project() do it, so that the
    // coverage of the batched path stays that of the per-tile path bit for bit.  (About one wave in a hundred goes there.)
    const float band = 1.6e-5f * (float)max(w, h) + 4e-3f;
    float u = 0.f, v = 0.f;
    bool exact = BAND && fabsf(cam2 - 1e-6f) < 1e-6f;
    if (!exact) {
        if (!(cam2 > 1e-6f)) return make_float4(0.f, 0.f, 0.f, 0.f);  // behind the camera (renderPanorama.m:1112-1116)
        const float rz = fast_rcp(cam2);
        u = fmaf(im.fx * cam0, rz, im.cx);
        v = fmaf(im.fy * cam1, rz, im.cy);
        exact = BAND && fminf(fminf(fabsf(u - 1.0f), fabsf(u - (float)w)), fminf(fabsf(v - 1.0f), fabsf(v - (float)h))) < band;
    }
    if (BAND && exact) {
        float de[3], wa_;
        exact_ray(de);
        if (!project(im, de, 2.0f, u, v, wa_)) return make_float4(0.f, 0.f, 0.f, 0.f);
        cam2 = fmaf(de[2], im.R[8], fmaf(de[1], im.R[5], de[0] * im.R[2]));
    } else if (!((u >= 1.0f) && (u <= (float)w) && (v >= 1.0f) && (v <= (float)h))) {
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float wa = cam2;
    if (angle_pow == 2.0f)
        wa = wa * wa;
    else if (angle_pow != 1.0f)
        wa = __powf(wa, angle_pow);
    const int xm = min((int)u, w - 1) - 1, ym = min((int)v, h - 1) - 1;  // 0-based x0, y0 (u, v >= 1: truncation is floor)
    const float s = u - (float)(xm + 1), t = v - (float)(ym + 1);
    const int off = (ym * w + xm) * 4;
    uint2 pa, pb;  // (x0, y0), (x1, y0) and (x0, y1), (x1, y1)
    if (BUF) {
        pa = ld_pair<uint2>(fi.rgba, off, 0);
        pb = ld_pair<uint2>(fi.rgba, off, fi.row_bytes);
    } else {
        const uint32_t* __restrict__ r0p = im.rgba + ((size_t)ym * w + xm);
        pa = make_uint2(r0p[0], r0p[1]);
        pb = make_uint2(r0p[w], r0p[w + 1]);
    }
    // the tent weight in closed form (DevImage::tx, ty) instead of two more table loads: lerp(wx) * lerp(wy)
    const float px = u - 1.0f, py = v - 1.0f;
    const float wf = fminf(fminf(px * im.tx[0], ((float)(w - 1) - px) * im.tx[1]), 1.0f) *
                     fminf(fminf(py * im.ty[0], ((float)(h - 1) - py) * im.ty[1]), 1.0f);
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float b00 = (float)((pa.x >> (8 * c)) & 255u), b10 = (float)((pa.y >> (8 * c)) & 255u);
        const float b01 = (float)((pb.x >> (8 * c)) & 255u), b11 = (float)((pb.y >> (8 * c)) & 255u);
        o[c] = lerp1(lerp1(b00, b10, s), lerp1(b01, b11, s), t) * im.g255[c];
    }
    return make_float4(o[0], o[1], o[2], wa * wf);
}

// the ray of ray_from_tables with the normalisation through v_rsq_f32 (the spherical / cylindrical rays are unit
// vectors up to rounding already)
__device__ __forceinline__ void ray_fast(const DevCanvas& cv, const ColTrig& ct, const RowTrig& rt, float xp, float yp,
                                         float d[3]) {
    if (cv.mode == APS_PROJ_CYLINDRICAL || cv.mode == APS_PROJ_SPHERICAL) {
        float x, y, z;
        if (cv.mode == APS_PROJ_CYLINDRICAL) {
            x = ct.s;
            y = rt.s;
            z = ct.c;
        } else {
            x = rt.c * ct.s;
            y = rt.s;
            z = rt.c * ct.c;
        }
        const float n2 = fmaf(z, z, fmaf(y, y, x * x));
        const float r = __builtin_amdgcn_rsqf(n2 > 1e-16f ? n2 : 1e-16f);
        d[0] = x * r;
        d[1] = y * r;
        d[2] = z * r;
    } else {
        canvas_ray(cv, xp, yp, d);
    }
}

// geometry + tent weight only (what the normalisation sums need): Wang * Wf, 0 outside the mask
__device__ __forceinline__ float sample_weight(const DevImage& im, const float d[3], float angle_pow) {
    float u, v, wa;
    if (!project(im, d, angle_pow, u, v, wa)) return 0.f;
    const int w = im.w, h = im.h;
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    x0 = max(1, min(x0, w - 1));
    y0 = max(1, min(y0, h - 1));
    const int x1 = min(x0 + 1, w), y1 = min(y0 + 1, h);
    const float s = u - (float)x0, t = v - (float)y0;
    const float wy0 = im.wy[y0 - 1], wy1 = im.wy[y1 - 1], wx0 = im.wx[x0 - 1], wx1 = im.wx[x1 - 1];
    const float f00 = wy0 * wx0, f10 = wy0 * wx1, f01 = wy1 * wx0, f11 = wy1 * wx1;
    const float top = (1.0f - s) * f00 + s * f10;
    const float bot = (1.0f - s) * f01 + s * f11;
    return wa * (top * (1.0f - t) + bot * t);
}

// fuseTile's rescale (renderPanorama.m:1009-1017) followed by multiBandBlending's own (multiBandBlending.m:72-85)
__device__ __forceinline__ float norm_weight(float w, const float2 nrm) {
    const float w1 = w * nrm.x;
    const float wv = w1 > 0.f ? w1 : 0.f;
    return nrm.y > 1e-8f ? wv / nrm.y : 0.f;
}

__device__ __forceinline__ float4 ld_compact(const float4* __restrict__ p, const Rect& r, int x, int y) {
    return in_rect(r, x, y) ? p[(size_t)(y - r.y0) * (r.x1 - r.x0) + (x - r.x0)] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// pass 1a: coverage, all tiles (blockIdx.z = tile)
// ------------------------------------------------------------------------------------------------
// Per 32 x 8 block and image: does any pixel map into the image (rowmask; as cover_kernel in render.hip, with the
// transcendental parts of the rays from per-block tables).  The footprints bound all later work on a layer.
__global__ __launch_bounds__(256) void rw_cover_kernel(RwArgs A, int n_img, int nby_max, const int* __restrict__ xshift,
                                                       unsigned long long* __restrict__ rowmask) {
    __shared__ unsigned long long s_any, s_cand;
    __shared__ float s_dc[3];
    __shared__ int s_cosb;
    __shared__ ColTrig s_ct[32];
    __shared__ RowTrig s_rt[8];
    const RwTile& T = A.tiles[blockIdx.z];
    const int ht = T.ht, wt = T.wt;
    if ((int)(blockIdx.x * 32) >= wt || (int)(blockIdx.y * 8) >= ht) return;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const bool in_tile = x < wt && y < ht;
    if (threadIdx.x < 32) s_ct[threadIdx.x] = col_trig(A.cv, (float)(T.c0 + min((int)(blockIdx.x * 32 + threadIdx.x), wt - 1)));
    if (threadIdx.x >= 64 && threadIdx.x < 72)
        s_rt[threadIdx.x - 64] = row_trig(A.cv, (float)(T.r0 + min((int)(blockIdx.y * 8 + threadIdx.x - 64), ht - 1)));
    if (threadIdx.x == 128) {
        float dc[3];
        canvas_ray(A.cv, (float)(T.c0 + min(blockIdx.x * 32 + 16, (unsigned)wt - 1)),
                   (float)(T.r0 + min(blockIdx.y * 8 + 4, (unsigned)ht - 1)), dc);
        s_dc[0] = dc[0];
        s_dc[1] = dc[1];
        s_dc[2] = dc[2];
        s_cosb = __float_as_int(1.0f);
    }
    __syncthreads();
    float d[3] = {0.f, 0.f, 1.f};
    if (in_tile) ray_from_tables(A.cv, s_ct[threadIdx.x & 31], s_rt[threadIdx.x >> 5], (float)(T.c0 + x), (float)(T.r0 + y), d);
    const float dn = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
    {
        float cb = in_tile ? fmaf(d[2], s_dc[2], fmaf(d[1], s_dc[1], d[0] * s_dc[0])) / fmaxf(dn, 1e-8f) : 1.0f;
        cb = fminf(fmaxf(cb, 0.0f), 1.0f);
        for (int off = 32; off > 0; off >>= 1) cb = fminf(cb, __shfl_xor(cb, off));
        if ((threadIdx.x & 63) == 0) atomicMin(&s_cosb, __float_as_int(cb));
    }
    __syncthreads();
    const float cosb = fmaxf(__int_as_float(s_cosb) - 1e-6f, 0.0f), sinb = sqrtf(fmaxf(0.0f, 1.0f - cosb * cosb));
    const unsigned long long colbit = 1ull << (blockIdx.x >> xshift[blockIdx.z]);
    unsigned long long* rm = rowmask + (size_t)blockIdx.z * n_img * nby_max;
    for (int base = 0; base < n_img; base += 64) {
        const int cnt = min(64, n_img - base);
        if (threadIdx.x < 64) {
            bool cand = false;
            if ((int)threadIdx.x < cnt) {
                const DevImage& im = A.imgs[base + threadIdx.x];
                const float ca = fmaf(s_dc[2], im.R[8], fmaf(s_dc[1], im.R[5], s_dc[0] * im.R[2]));
                const float ci = im.cmin, si = sqrtf(fmaxf(0.0f, 1.0f - ci * ci));
                cand = ci <= 0.0f || cosb <= 0.0f || ci * cosb - si * sinb <= 0.0f || ca >= (ci * cosb - si * sinb) - 1e-5f;
            }
            const unsigned long long b = __ballot(cand);
            if (threadIdx.x == 0) {
                s_cand = b;
                s_any = 0ull;
            }
        }
        __syncthreads();
        unsigned long long mine = 0ull;
        for (unsigned long long todo = s_cand; todo; todo &= todo - 1) {
            const int i = __ffsll((long long)todo) - 1;
            const DevImage& im = A.imgs[base + i];
            const float cam2 = fmaf(d[2], im.R[8], fmaf(d[1], im.R[5], d[0] * im.R[2]));
            if (!__any(in_tile && cam2 >= im.cmin * dn)) continue;
            float u, v, wa;
            const bool m = in_tile && project(im, d, A.angle_pow, u, v, wa);
            if (__any(m)) mine |= 1ull << i;
        }
        if ((threadIdx.x & 63) == 0 && mine) atomicOr(&s_any, mine);
        __syncthreads();
        const unsigned long long all = s_any;
        if ((int)threadIdx.x < cnt && ((all >> threadIdx.x) & 1ull))
            atomicOr(&rm[(size_t)(base + threadIdx.x) * nby_max + blockIdx.y], colbit);
        __syncthreads();
    }
}

// rowmask -> {x0, y0, x1, y1} per (tile, image), pixels of the tile, half-open; x1 <= x0 when nothing is covered
__global__ void rw_footprint_kernel(const unsigned long long* __restrict__ rowmask, const RwTile* __restrict__ tiles,
                                    const int* __restrict__ xshift, int n_img, int nby_max, int total,
                                    int* __restrict__ bbox) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;  // tile * n_img + image
    if (e >= total) return;
    const int t = e / n_img;
    const int ht = tiles[t].ht, wt = tiles[t].wt, nby = (ht + 7) / 8, xs = xshift[t];
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
        x0 = (lo << xs) * 32;
        x1 = min(((hi + 1) << xs) * 32, wt);
    } else {
        y0 = 0;
    }
    bbox[4 * e + 0] = x0;
    bbox[4 * e + 1] = y0;
    bbox[4 * e + 2] = x1;
    bbox[4 * e + 3] = y1;
}

// ------------------------------------------------------------------------------------------------
// pass 1b: the level-0 layers of every tile: sampleOneTile for each (pixel, layer) exactly once
// ------------------------------------------------------------------------------------------------
// One workgroup per 32 x 8 block of a tile, looping over the tile's layers in order: ray once per pixel, then per
// layer project -> bilinear gather -> Wang * Wf (gathers of a layer are issued before the previous layer's sample is
// finished).  The samples wait in LDS until the two normalisation sums of the pixel are known (fuseTile's
// renderPanorama.m:1009-1017, then multiBandBlending.m:72-85, both in layer order) and go to the compact G_0 store
// with their final weight: the level-0 layer is written once and the weights are never re-read for normalising.
constexpr int kWL = 6;  // layers of a block whose samples are parked in LDS; further ones are re-sampled
// FAST (default): sample_fast / ray_fast and reciprocal-multiplies in the two normalisations.  !FAST (APS_WARP_EXACT=1):
// the arithmetic of the per-tile path bit for bit (IEEE divisions, u8 table, 4-tap tent products) - kept so that the
// tests can still pin the REST of the batched pipeline (footprints, compact stores, pyramids, collapse) byte for byte
// against render.hip's per-tile kernels.
// V (experiment switches of the FAST form, APS_WARP_VARIANT, default 6): bit 0 = trig tables from global memory (else
// evaluated per block into LDS), bit 1 = exact re-projection in the border band, bit 2 = buffer loads for the taps.
// The walking form of a block: ray once per pixel, then the tile's layers that meet the block one after the other
// (sample -> first sum, second sum, store).  ct / rt: the pixel's column / row trig entries.
template <bool FAST, int V>
__device__ __forceinline__ void warp_block_walk(const RwArgs& A, const RwTile& T, int x0, int y0, float* s_u8, float4 (*s_g)[256],
                                                const ColTrig& ct, const RowTrig& rt) {
    const int tid = threadIdx.x;
    const int h = T.ht, w = T.wt;
    const int x = x0 + (tid & (kUW - 1)), y = y0 + tid / kUW;
    const bool in_tile = x < w && y < h;
    float d[3] = {0.f, 0.f, 1.f};
    if (in_tile) {
        if (FAST)
            ray_fast(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), d);
        else
            ray_from_tables(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), d);
    }
    // (always_inline: left to its cost model the compiler turned `sample` into a real function - every by-reference
    // capture, the kernel arguments included, then lives in scratch memory: 400 bytes per lane and a 5x slower kernel)
    auto exact_ray = [&](float de[3]) __attribute__((always_inline)) {
        ray_from_tables(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), de);
    };
    auto sample = [&](const DevImage& im) __attribute__((always_inline)) {
        if (FAST && im.w > 3 && im.h > 3)
            return sample_fast<(V & 2) != 0, (V & 4) != 0>(im, (V & 4) ? fast_image(im) : FastImage{}, d, A.angle_pow, exact_ray);
        if (FAST) {  // (an image of fewer than four rows or columns: the exact sampler on the exact ray)
            float de[3];
            exact_ray(de);
            const Sample sm = sample_one(im, de, A.angle_pow);
            return make_float4(sm.s[0], sm.s[1], sm.s[2], sm.wang * sm.wf);
        }
        return sample_lut(im, d, A.angle_pow, s_u8);
    };
    const int bx1 = min(x0 + kUW, w), by1 = min(y0 + kUH, h);
    const LayerSet ls = block_layers(A, T, 0, x0, y0, bx1, by1);
    // pass A: samples and the first sum
    float ssum = 0.f;
    bool any = false;
    int kc = 0;  // layers met so far (block-uniform)
    for_block_layers(ls, A, T, 0, x0, y0, bx1, by1, [&](int k) {
        const RwEntry& E = A.entries[T.e0 + k];
        const Rect g = E.g[0];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in_tile && in_rect(g, x, y)) v = sample(A.imgs[E.img]);
        ssum = ssum + v.w;
        any |= v.w > 0.f;
        if (kc < kWL) s_g[kc][tid] = v;
        ++kc;
    });
    const float inv = ssum > 1e-8f ? (FAST ? fast_rcp(ssum) : 1.0f / ssum) : 0.f;
    // pass B: the second sum over the rescaled weights
    float s2 = 0.f;
    kc = 0;
    for_block_layers(ls, A, T, 0, x0, y0, bx1, by1, [&](int k) {
        float wv;
        if (kc < kWL) {
            wv = s_g[kc][tid].w;
        } else {
            const RwEntry& E = A.entries[T.e0 + k];
            wv = (in_tile && in_rect(E.g[0], x, y)) ? sample(A.imgs[E.img]).w : 0.f;
        }
        const float w1 = wv * inv;
        s2 = s2 + (w1 > 0.f ? w1 : 0.f);
        ++kc;
    });
    const float2 nrm = make_float2(inv, s2);
    const float inv2 = s2 > 1e-8f ? fast_rcp(s2) : 0.f;  // (FAST only)
    // pass C: store with the final weight
    kc = 0;
    for_block_layers(ls, A, T, 0, x0, y0, bx1, by1, [&](int k) {
        const RwEntry& E = A.entries[T.e0 + k];
        const Rect g = E.g[0];
        if (in_tile && in_rect(g, x, y)) {
            float4 v = kc < kWL ? s_g[kc][tid] : sample(A.imgs[E.img]);
            if (FAST) {
                const float w1 = v.w * inv;
                v.w = (w1 > 0.f ? w1 : 0.f) * inv2;
            } else {
                v.w = norm_weight(v.w, nrm);
            }
            A.G[E.off[0] + (size_t)(y - g.y0) * (g.x1 - g.x0) + (x - g.x0)] = v;
        }
        ++kc;
    });
    if (in_tile) A.cov[(size_t)T.plane + (size_t)y * w + x] = any ? 1 : 0;
}

template <bool FAST, int V = 7>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void rw_warp_kernel(RwArgs A, const int* __restrict__ blk_ptr, int n_blocks) {
    __shared__ float s_u8[FAST ? 1 : 256];
    __shared__ float4 s_g[kWL][256];
    __shared__ ColTrig s_ct[(V & 1) ? 1 : kUW];
    __shared__ RowTrig s_rt[(V & 1) ? 1 : kUH];
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int t = find_segment(blk_ptr, A.n_tiles, bid);
    const RwTile& T = A.tiles[t];
    const int tid = threadIdx.x;
    const int h = T.ht, w = T.wt;
    const int local = bid - blk_ptr[t], nbx = (w + kUW - 1) / kUW;
    const int x0 = (local % nbx) * kUW, y0 = (local / nbx) * kUH;
    if (!(V & 1)) {
        if (tid < kUW) s_ct[tid] = col_trig(A.cv, (float)(T.c0 + min(x0 + tid, w - 1)));
        if (tid >= 64 && tid < 64 + kUH) s_rt[tid - 64] = row_trig(A.cv, (float)(T.r0 + min(y0 + tid - 64, h - 1)));
    }
    if (!FAST) s_u8[tid] = (float)tid / 255.0f;
    if (!FAST || !(V & 1)) __syncthreads();
    const ColTrig ct = (V & 1) ? A.ct[T.c0 + min(x0 + (tid & (kUW - 1)), w - 1)] : s_ct[tid & (kUW - 1)];
    const RowTrig rt = (V & 1) ? A.rt[T.r0 + min(y0 + tid / kUW, h - 1)] : s_rt[tid / kUW];
    warp_block_walk<FAST, V>(A, T, x0, y0, s_u8, s_g, ct, rt);
}

// ---- the warp in its staged form (default) ----------------------------------------------------------------------------
// rw_warp_kernel<true> walks a block's layers one after the other and pays, per layer and serially, a scalar load of
// the entry, a dependent scalar load of the image record and a dependent gather: profiles/r03c_pmc_sq_rw_warp_fast_v7.txt
// shows 61 % of its wave cycles waiting and no fewer vector instructions than the exact form.  Here
//   * the layers that meet the block (<= kWF, else the walking form takes the block) are staged ONCE per block: lane j of
//     the first wave copies layer j's constants (rotation, intrinsics, gain, tent slopes, footprint, store offset) into
//     LDS - the two dependent loads of every layer in flight together;
//   * pass A needs no memory at all: projection, mask and weight (the tent in closed form) per layer out of LDS
//     broadcasts, (u, v, w) kept in registers (the layer loops are unrolled over kWF with uniform guards);
//   * pass B is arithmetic on those registers; pass C gathers the four taps of the layers that cover the pixel - the
//     only dependent memory round trip left per layer - interpolates on the raw bytes and stores with the final weight.
// Same values as rw_warp_kernel<true> (sample_fast's arithmetic, the exact re-projection in the border band).
// (Measured and dropped: culling a block's layers against the projected image outline - four corners + centre per
// layer in phase 0, zero fill for the culled ones - instead of the footprint rectangle: the same 2.26e9 vector
// instructions per launch, one more barrier, 5.9 -> 6.6 ms; the rectangles are tight enough on these scenes.)
constexpr int kWF = 8;
// One staged layer = eight 16-byte groups, so that a pass fetches what it needs with a few ds_read_b128 issued together
// (field by field the compiler read a dword, waited, branched - four LDS round trips for the rectangle test alone):
//   q0 R[0..3]   q1 R[4..7]   q2 R[8] fx fy cx   q3 cy tx0 tx1 ty0   q4 ty1 (w) (h) (ok)   q5 footprint x0 y0 x1 y1
//   q6 g255[0..2] -   q7 rgba pointer, float4 offset of the compact G_0 store
struct LayerConst {
    float4 q[8];
};
static_assert(sizeof(LayerConst) == 128, "LayerConst");
__device__ __forceinline__ int f2i(float f) { return __float_as_int(f); }
__device__ __forceinline__ float i2f(int i) { return __int_as_float(i); }

// project() on a staged layer, the same expressions in the same order (render_dev.h)
__device__ __forceinline__ bool project_lc(const float R[9], float fx, float fy, float cx, float cy, int w, int h, const float d[3],
                                           float& u, float& v, float& cz_out) {
    float cam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c] = fmaf(d[2], R[c + 6], fmaf(d[1], R[c + 3], d[0] * R[c]));
    const float epsz = 1e-6f;
    const bool front = cam[2] > epsz;
    const float cz = cam[2] > epsz ? cam[2] : epsz;
    u = fx * (cam[0] / cz) + cx;
    v = fy * (cam[1] / cz) + cy;
    cz_out = cam[2];
    if (!isfinite(u) || !isfinite(v)) {
        u = 1.0f;
        v = 1.0f;
    }
    const bool inside = (u >= 1.0f) && (u <= (float)w) && (v >= 1.0f) && (v <= (float)h);
    return inside && front && cam[2] > 0.0f;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void rw_warp_staged_kernel(
    RwArgs A, const int* __restrict__ blk_ptr, int n_blocks) {
    __shared__ LayerConst s_lc[kWF];
    __shared__ ColTrig s_ct[kUW];
    __shared__ RowTrig s_rt[kUH];
    __shared__ float4 s_g[kWL][256];  // (the walking form's parking area, for the blocks that fall back to it)
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int t = find_segment(blk_ptr, A.n_tiles, bid);
    const RwTile& T = A.tiles[t];
    const int tid = threadIdx.x, lane = tid & 63;
    const int h = T.ht, w = T.wt;
    const int local = bid - blk_ptr[t], nbx = (w + kUW - 1) / kUW;
    const int x0 = (local % nbx) * kUW, y0 = (local / nbx) * kUH;
    const int bx1 = min(x0 + kUW, w), by1 = min(y0 + kUH, h);
    if (tid < kUW) s_ct[tid] = col_trig(A.cv, (float)(T.c0 + min(x0 + tid, w - 1)));
    if (tid >= 64 && tid < 64 + kUH) s_rt[tid - 64] = row_trig(A.cv, (float)(T.r0 + min(y0 + tid - 64, h - 1)));
    const LayerSet ls = block_layers(A, T, 0, x0, y0, bx1, by1);
    const int n0 = __popcll(ls.m[0]);
    const int nvis = n0 + __popcll(ls.m[1]);
    static_assert(kLM == 2, "two mask words");
    const bool staged = ls.listed && nvis <= kWF && (A.angle_pow == 2.0f || A.angle_pow == 1.0f);
    if (staged && tid < 64) {  // lane k of the first wave owns entries k and 64 + k
#pragma unroll
        for (int c = 0; c < kLM; ++c) {
            if ((ls.m[c] >> lane) & 1ull) {
                const int rank = (c ? n0 : 0) + __popcll(ls.m[c] & ((1ull << lane) - 1ull));
                const RwEntry& E = A.entries[T.e0 + c * 64 + lane];
                const DevImage& im = A.imgs[E.img];
                LayerConst L;
                L.q[0] = make_float4(im.R[0], im.R[1], im.R[2], im.R[3]);
                L.q[1] = make_float4(im.R[4], im.R[5], im.R[6], im.R[7]);
                L.q[2] = make_float4(im.R[8], im.fx, im.fy, im.cx);
                L.q[3] = make_float4(im.cy, im.tx[0], im.tx[1], im.ty[0]);
                // (ok: the closed-form tent and the tap pairs need four rows and columns)
                L.q[4] = make_float4(im.ty[1], i2f(im.w), i2f(im.h), i2f((im.w > 3 && im.h > 3) ? 1 : 0));
                L.q[5] = make_float4(i2f(E.g[0].x0), i2f(E.g[0].y0), i2f(E.g[0].x1), i2f(E.g[0].y1));
                L.q[6] = make_float4(im.g255[0], im.g255[1], im.g255[2], 0.f);
                const unsigned long long pa = (unsigned long long)(uintptr_t)im.rgba, po = (unsigned long long)E.off[0];
                L.q[7] = make_float4(i2f((int)(unsigned)pa), i2f((int)(unsigned)(pa >> 32)), i2f((int)(unsigned)po), i2f((int)(unsigned)(po >> 32)));
                s_lc[rank] = L;
            }
        }
    }
    __syncthreads();
    const ColTrig ct = s_ct[tid & (kUW - 1)];
    const RowTrig rt = s_rt[tid / kUW];
    bool small_image = false;
    if (staged)
        for (int j = 0; j < nvis; ++j) small_image |= f2i(s_lc[j].q[4].w) == 0;
    if (!staged || small_image) {
        warp_block_walk<true, 6>(A, T, x0, y0, nullptr, s_g, ct, rt);
        return;
    }
    const int x = x0 + (tid & (kUW - 1)), y = y0 + tid / kUW;
    const bool in_tile = x < w && y < h;
    float d[3] = {0.f, 0.f, 1.f};
    if (in_tile) ray_fast(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), d);
    // pass A: (u, v, w) per staged layer, no global memory; parked in LDS (the parking area of the walking form, which
    // this block does not use).  Rolled loops: unrolled over kWF the kernel was 17 000 instructions, twice the
    // instruction cache.
    float* s_u = reinterpret_cast<float*>(&s_g[0][0]);  // [kWF][256] each
    float* s_v = s_u + kWF * 256;
    float* s_w = s_v + kWF * 256;
    static_assert(3 * kWF * 256 * sizeof(float) <= sizeof(float4) * kWL * 256, "parking area");
    float ssum = 0.f;
    unsigned okm = 0u;  // bit j: layer j covers the pixel (its colour is stored even where its weight is zero)
    for (int j = 0; j < nvis; ++j) {
        const float4 q5 = s_lc[j].q[5];
        float uo = 0.f, vo = 0.f, wo = 0.f;
        if ((int)in_tile & (int)(x >= f2i(q5.x)) & (int)(x < f2i(q5.z)) & (int)(y >= f2i(q5.y)) & (int)(y < f2i(q5.w))) {
            const float4 q0 = s_lc[j].q[0], q1 = s_lc[j].q[1], q2 = s_lc[j].q[2], q3 = s_lc[j].q[3], q4 = s_lc[j].q[4];
            const float R[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
            const float fx = q2.y, fy = q2.z, cx = q2.w, cy = q3.x;
            const int iw = f2i(q4.y), ih = f2i(q4.z);
            const float cam0 = fmaf(d[2], R[6], fmaf(d[1], R[3], d[0] * R[0]));
            const float cam1 = fmaf(d[2], R[7], fmaf(d[1], R[4], d[0] * R[1]));
            float cam2 = fmaf(d[2], R[8], fmaf(d[1], R[5], d[0] * R[2]));
            const float fw = (float)iw, fh = (float)ih;
            const float band = 1.6e-5f * fmaxf(fw, fh) + 4e-3f;  // see sample_fast
            float u = 0.f, v = 0.f;
            bool ok = false, exact = fabsf(cam2 - 1e-6f) < 1e-6f;
            if (!exact && cam2 > 1e-6f) {
                const float rz = fast_rcp(cam2);
                u = fmaf(fx * cam0, rz, cx);
                v = fmaf(fy * cam1, rz, cy);
                exact = fminf(fminf(fabsf(u - 1.0f), fabsf(u - fw)), fminf(fabsf(v - 1.0f), fabsf(v - fh))) < band;
                ok = ((int)(u >= 1.0f) & (int)(u <= fw) & (int)(v >= 1.0f) & (int)(v <= fh)) != 0;
            }
            if (exact) {
                float de[3];
                ray_from_tables(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), de);
                ok = project_lc(R, fx, fy, cx, cy, iw, ih, de, u, v, cam2);
            }
            if (ok) {
                const float wa = A.angle_pow == 2.0f ? cam2 * cam2 : cam2;
                const float px = u - 1.0f, py = v - 1.0f;
                const float wf = fminf(fminf(px * q3.y, (fw - 1.0f - px) * q3.z), 1.0f) *
                                 fminf(fminf(py * q3.w, (fh - 1.0f - py) * q4.x), 1.0f);
                uo = u;
                vo = v;
                wo = wa * wf;
                okm |= 1u << j;
            }
        }
        s_u[j * 256 + tid] = uo;
        s_v[j * 256 + tid] = vo;
        s_w[j * 256 + tid] = wo;
        ssum = ssum + wo;
    }
    const float inv = ssum > 1e-8f ? fast_rcp(ssum) : 0.f;
    float s2 = 0.f;
    bool any = false;
    for (int j = 0; j < nvis; ++j) {
        const float wv = s_w[j * 256 + tid];
        const float w1 = wv * inv;
        s2 = s2 + (w1 > 0.f ? w1 : 0.f);
        any |= wv > 0.f;
    }
    const float inv2 = s2 > 1e-8f ? fast_rcp(s2) : 0.f;
    // pass C: taps, interpolation on the bytes, store with the final weight (zeros where the layer does not cover)
    for (int j = 0; j < nvis; ++j) {
        const float4 q5 = s_lc[j].q[5];
        const int gx0 = f2i(q5.x), gy0 = f2i(q5.y), gx1 = f2i(q5.z), gy1 = f2i(q5.w);
        if ((int)in_tile & (int)(x >= gx0) & (int)(x < gx1) & (int)(y >= gy0) & (int)(y < gy1)) {
            const float4 q7 = s_lc[j].q[7];
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((okm >> j) & 1u) {
                const float4 q4 = s_lc[j].q[4], q6 = s_lc[j].q[6];
                const int iw = f2i(q4.y), ih = f2i(q4.z);
                const float u = s_u[j * 256 + tid], v = s_v[j * 256 + tid];
                const int xm = min((int)u, iw - 1) - 1, ym = min((int)v, ih - 1) - 1;
                const float s = u - (float)(xm + 1), tt = v - (float)(ym + 1);
                const unsigned long long pa = ((unsigned long long)(unsigned)f2i(q7.y) << 32) | (unsigned)f2i(q7.x);
                const __attribute__((address_space(1))) uint32_t* r0p =
                    (const __attribute__((address_space(1))) uint32_t*)pa + (ym * iw + xm);  // (an image has < 2^31 pixels)
                const uint32_t p00 = r0p[0], p10 = r0p[1], p01 = r0p[iw], p11 = r0p[iw + 1];
                const float g3[3] = {q6.x, q6.y, q6.z};
                float c3[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float b00 = (float)((p00 >> (8 * c)) & 255u), b10 = (float)((p10 >> (8 * c)) & 255u);
                    const float b01 = (float)((p01 >> (8 * c)) & 255u), b11 = (float)((p11 >> (8 * c)) & 255u);
                    c3[c] = lerp1(lerp1(b00, b10, s), lerp1(b01, b11, s), tt) * g3[c];
                }
                const float w1 = s_w[j * 256 + tid] * inv;
                o = make_float4(c3[0], c3[1], c3[2], (w1 > 0.f ? w1 : 0.f) * inv2);
            }
            const long long off = (long long)(((unsigned long long)(unsigned)f2i(q7.w) << 32) | (unsigned)f2i(q7.z));
            A.G[off + ((y - gy0) * (gx1 - gx0) + (x - gx0))] = o;  // (a footprint store is < 2^31 bytes: host check)
        }
    }
    if (in_tile) A.cov[(size_t)T.plane + (size_t)y * w + x] = any ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// pass 2: G_{l+1} = imresize(imgaussfilt(G_l)); at l = 0 G_0 is the warp, evaluated straight into LDS
// ------------------------------------------------------------------------------------------------
// Both kernels produce one kBW x kBH block of G_{l+1} per layer from the blurred patch its taps reach (<= kBR x kBC,
// plus the filter radius): column pass, row pass, then the two resize passes, all out of LDS.  Same fma chains as
// mb_blur_resize_kernel (render.hip); the resize runs as two passes through LDS - the first-pass value of an
// (output row, patch column) is one chain whichever output pixel consumes it - instead of once per consumer.
template <int R>
struct DownShape {
    static constexpr int IC = kBC + 2 * R, IR = kBR + 2 * R;
    static constexpr int T1 = kBH * IC > kBR * kBW ? kBH * IC : kBR * kBW;  // first-pass buffer of the resize
};

// `load(ly, lx)` yields the haloed input patch (nir x nic); on return the block's outputs have been written.
template <int R, class Load>
__device__ __forceinline__ void blur_resize_block(Load load, float4* __restrict__ s_a, float4* __restrict__ s_b, const Taps& tp,
                                                  int nbr, int nbc, int bx0, int by0, int ox0, int oy0, int ox1, int oy1,
                                                  const int* __restrict__ tci, const float* __restrict__ tcw,
                                                  const int* __restrict__ tri, const float* __restrict__ trw, bool rows_first,
                                                  float4* __restrict__ out, const Rect& go) {
    constexpr int IC = DownShape<R>::IC;
    const int tid = threadIdx.x, nic = nbc + 2 * R;
    // The column (vertical) Gaussian pass reads its register window straight from the source: the patch itself never
    // goes through LDS (that cost a fill pass, a barrier, a window load and another barrier).  The row pass then runs IN
    // PLACE through a register window, so the patch-sized buffer is the only large LDS allocation and two workgroups
    // fit a CU (one's barriers are covered by the other's work).
    {  // column pass: thread = (row segment, column); rows shared by two segments are fetched twice (L1/L2 hits)
        constexpr int SEG = kNT / IC, RS = (kBR + SEG - 1) / SEG;
        const int col = tid % IC, r0 = (tid / IC) * RS;
        const bool act = tid < SEG * IC && col < nic && r0 < nbr;
        if (act) {
            float4 win[RS + 2 * R];
#pragma unroll
            for (int i = 0; i < RS + 2 * R; ++i)
                win[i] = r0 + i < nbr + 2 * R ? load(r0 + i, col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                if (r0 + i >= nbr) break;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], win[i + t], a);
                s_a[(r0 + i) * IC + col] = a;
            }
        }
    }
    {  // row (horizontal) pass: thread = (row, column segment of CS columns; CS odd keeps the b128 accesses spread)
        constexpr int CS = 7, SEG = (kBC + CS - 1) / CS;
        static_assert(SEG * kBR <= kNT, "row pass needs one thread per (row, segment)");
        const int row = tid / SEG, c0 = (tid % SEG) * CS;
        const bool act = row < nbr && c0 < nbc;
        float4 win[CS + 2 * R];
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < CS + 2 * R; ++i)
                win[i] = c0 + i < nic ? s_a[row * IC + c0 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int i = 0; i < CS; ++i) {
                if (c0 + i >= nbc) break;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t <= 2 * R; ++t) a = fma4(tp.k[t], win[i + t], a);
                s_a[row * IC + c0 + i] = a;
            }
        }
    }
    __syncthreads();
    // zero-weight table slots (always trailing) add 0 * finite to a finished chain: the value is unchanged
    const int now = ox1 - ox0, noh = oy1 - oy0;
    if (rows_first) {
        // first pass over rows: t1[yo][c] = sum_tr rw * B[ri][c]  (s_b, pitch IC); second over columns
        for (int ee = tid; ee < kBH * IC; ee += kNT) {
            const int yo = ee / IC, c = ee - yo * IC;
            if (yo >= noh || c >= nbc) continue;
            const size_t ro = (size_t)(oy0 + yo) * kTD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < kTD; ++t) v = fma4(trw[ro + t], s_a[(tri[ro + t] - by0) * IC + c], v);
            s_b[ee] = v;
        }
        __syncthreads();
        const int xo = tid & (kBW - 1), yo = tid / kBW;
        if (xo < now && yo < noh) {
            const size_t co = (size_t)(ox0 + xo) * kTD;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < kTD; ++t) a = fma4(tcw[co + t], s_b[yo * IC + (tci[co + t] - bx0)], a);
            out[(size_t)(oy0 + yo - go.y0) * (go.x1 - go.x0) + (ox0 + xo - go.x0)] = a;
        }
    } else {
        // first pass over columns: t1[r][xo] = sum_tc cw * B[r][ci]  (s_b, pitch kBW); second over rows
        for (int ee = tid; ee < kBR * kBW; ee += kNT) {
            const int r = ee / kBW, xo = ee - r * kBW;
            if (r >= nbr || xo >= now) continue;
            const size_t co = (size_t)(ox0 + xo) * kTD;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < kTD; ++t) v = fma4(tcw[co + t], s_a[r * IC + (tci[co + t] - bx0)], v);
            s_b[ee] = v;
        }
        __syncthreads();
        const int xo = tid & (kBW - 1), yo = tid / kBW;
        if (xo < now && yo < noh) {
            const size_t ro = (size_t)(oy0 + yo) * kTD;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < kTD; ++t) a = fma4(trw[ro + t], s_b[(tri[ro + t] - by0) * kBW + xo], a);
            out[(size_t)(oy0 + yo - go.y0) * (go.x1 - go.x0) + (ox0 + xo - go.x0)] = a;
        }
    }
}

// One workgroup per kBW x kBH block of one layer's G_{l+1} footprint, input from the compact store of G_l.
template <int R>
__global__ __launch_bounds__(kNT, 4) void rw_down_kernel(RwArgs A, int l, const int* __restrict__ blk_ptr, int n_blocks) {
    constexpr int IC = DownShape<R>::IC;
    extern __shared__ float4 s_dyn[];
    float4* s_a = s_dyn;             // kBR x IC: column-blurred patch, then (in place) the blurred patch
    float4* s_b = s_dyn + kBR * IC;  // DownShape<R>::T1: first pass of the resize
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int e = find_segment(blk_ptr, A.n_entries, bid);
    const RwEntry& E = A.entries[e];
    const RwTile& T = A.tiles[E.tile];
    const int tid = threadIdx.x;
    const int h = T.lh[l], w = T.lw[l];
    const Rect gi = E.g[l], go = E.g[l + 1];
    const int gow = go.x1 - go.x0;
    const int local = bid - blk_ptr[e], nbx = (gow + kBW - 1) / kBW;
    const int ox0 = go.x0 + (local % nbx) * kBW, oy0 = go.y0 + (local / nbx) * kBH;
    const int ox1 = min(ox0 + kBW, go.x1), oy1 = min(oy0 + kBH, go.y1);
    const int* __restrict__ tci = A.td_idx + (size_t)T.tdc[l] * kTD;
    const int* __restrict__ tri = A.td_idx + (size_t)T.tdr[l] * kTD;
    const float* __restrict__ tcw = A.td_w + (size_t)T.tdc[l] * kTD;
    const float* __restrict__ trw = A.td_w + (size_t)T.tdr[l] * kTD;
    const int bx0 = tci[(size_t)ox0 * kTD], bx1 = tci[(size_t)(ox1 - 1) * kTD + kTD - 1];
    const int by0 = tri[(size_t)oy0 * kTD], by1 = tri[(size_t)(oy1 - 1) * kTD + kTD - 1];
    const int nbc = bx1 - bx0 + 1, nbr = by1 - by0 + 1;
    if (nbc > kBC || nbr > kBR) {
        if (tid == 0) atomicOr(A.status, 2);
        return;
    }
    const float4* __restrict__ gin = A.G + E.off[l];
    const int giw = gi.x1 - gi.x0;
    // patch position (ly, lx) -> level-l pixel with replicate padding at the tile border, zero outside the footprint
    auto load = [&](int ly, int lx) {
        const int gy = min(max(by0 + ly - R, 0), h - 1), gx = min(max(bx0 + lx - R, 0), w - 1);
        return in_rect(gi, gx, gy) ? gin[(size_t)(gy - gi.y0) * giw + (gx - gi.x0)] : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    blur_resize_block<R>(load, s_a, s_b, A.tp, nbr, nbc, bx0, by0, ox0, oy0, ox1, oy1, tci, tcw, tri, trw, (T.rf_down >> l) & 1,
                         A.G + E.off[l + 1], go);
}

// ------------------------------------------------------------------------------------------------
// pass 3 (coarse to fine): F_l = imresize(F_{l+1}) + sum_k (G_l,k - imresize(G_{l+1},k)) .* w_l,k ; at the tile's
// coarsest level F = sum_k G_k .* w_k (multiBandBlending.m:136-167); level 0 paints (renderPanorama.m:408-425)
// ------------------------------------------------------------------------------------------------
// An enlargement has at most two non-zero taps per axis (table slots 0, 1; a zero-weight slot adds 0 * finite).
struct UpTaps {
    int r0, r1, c0, c1;
    float wr0, wr1, wc0, wc1;
};
struct UpVals {
    float4 v00, v01, v10, v11;
};
template <class LD>
__device__ __forceinline__ UpVals up_load(const UpTaps& u, LD ld) {
    return UpVals{ld(u.c0, u.r0), ld(u.c0, u.r1), ld(u.c1, u.r0), ld(u.c1, u.r1)};
}
__device__ __forceinline__ float4 up_combine(const UpTaps& u, bool rows_first, const UpVals& v) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rows_first) {  // per column tap: the chain over the row taps, then the chain over the columns
        const float4 a = fma4(u.wr1, v.v01, fma4(u.wr0, v.v00, z));
        const float4 b = fma4(u.wr1, v.v11, fma4(u.wr0, v.v10, z));
        return fma4(u.wc1, b, fma4(u.wc0, a, z));
    }
    const float4 a = fma4(u.wc1, v.v10, fma4(u.wc0, v.v00, z));
    const float4 b = fma4(u.wc1, v.v11, fma4(u.wc0, v.v01, z));
    return fma4(u.wr1, b, fma4(u.wr0, a, z));
}
template <class LD>
__device__ __forceinline__ float4 upsample2(const UpTaps& u, bool rows_first, LD ld) {
    return up_combine(u, rows_first, up_load(u, ld));
}

// A compact footprint store read as a buffer: a lane outside the footprint asks for an offset past the end and the
// hardware range check returns zeros.  No branch and no select around the load, so the loads of a group issue back
// to back (as "cond ? p[i] : 0" the compiler sank every load into its own exec-masked block and, reusing registers,
// put a full vmcnt(0) wait in front of the next one).  Footprints are < 2^31 bytes (checked by the host).
struct FootBuf {
    __amdgpu_buffer_rsrc_t rs;
    Rect r;
    int pw;
};
__device__ __forceinline__ FootBuf foot_buf(const float4* base, const Rect& r) {
    FootBuf b;
    b.r = r;
    b.pw = r.x1 - r.x0;
    b.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(base), 0, (r.y1 - r.y0) * b.pw * 16, 0x00020000);
    return b;
}
__device__ __forceinline__ float4 ld_foot(const FootBuf& b, int x, int y) {
    const unsigned off = in_rect(b.r, x, y) ? (unsigned)((y - b.r.y0) * b.pw + (x - b.r.x0)) * 16u : 0x80000000u;
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(b.rs, (int)off, 0, 0);
    float4 f;  // (member access on the builtin's vector type is narrowed to one dword by this compiler; a copy is not)
    static_assert(sizeof v == sizeof f, "b128");
    __builtin_memcpy(&f, &v, sizeof f);
    return f;
}

#ifndef APS_DFRS
#define APS_DFRS 6  // (measured: 4 rows per thread with 320 threads - 84 instead of 100 VGPRs, 1.45x instead of 1.36x re-reads - 4.35 -> 5.1 ms)
#endif
constexpr int kDFRS = APS_DFRS;                     // output rows per vertical-pass thread (register window 2 (RS - 1) + NC rows)
constexpr int kDFT = kDFRS == 6 ? 256 : 320;        // threads: ceil(16 / RS) segments x <= 82 patch columns
// ------------------------------------------------------------------------------------------------
// pass 2, fused form (default): one kBW x kBH block of G_{l+1} per workgroup straight from the compact store of G_l
// ------------------------------------------------------------------------------------------------
// Vertical pass out of REGISTERS, horizontal pass out of LDS, one barrier: a thread owns one patch column and RS
// consecutive output rows, whose source rows (stride two between output rows: r0 .. r0 + 2 (RS - 1) + NC - 1) it fetches
// once into a register window (range-checked buffer loads: zero outside the layer's footprint) - every source sample is
// requested ~1.4 times instead of five, nothing is staged, and the only LDS buffer is the kBH x VC half-height plane
// (20 KB against the 69 KB of the two-step kernel: eight workgroups fit a CU instead of two).  Rows whose taps do not
// advance by exactly two (one row in ~a thousand when a level's height is odd) take a plain per-tap loop.
template <int R>
__global__ __launch_bounds__(kDFT) void rw_down_fused_kernel(RwArgs A, int l, const int* __restrict__ blk_ptr, int n_blocks) {
    constexpr int NC = kTD + 2 * R, NCP = (NC + 3) & ~3, VC = 2 * kBW + NC + 2, RS = kDFRS, NSEG = (kBH + RS - 1) / RS;
    constexpr int WIN = 2 * (RS - 1) + NC;
    static_assert(NSEG * VC <= kDFT && kBH <= 16 && kBW == 32 && kDFT >= kBW * 8, "thread mapping");
    __shared__ float4 s_v[kBH * VC];
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int e = find_segment(blk_ptr, A.n_entries, bid);
    const RwEntry& E = A.entries[e];
    const RwTile& T = A.tiles[E.tile];
    const int tid = threadIdx.x;
    const Rect gi = E.g[l], go = E.g[l + 1];
    const int gow = go.x1 - go.x0;
    const int local = bid - blk_ptr[e], nbx = (gow + kBW - 1) / kBW;
    const int ox0 = go.x0 + (local % nbx) * kBW, oy0 = go.y0 + (local / nbx) * kBH;
    const int ox1 = min(ox0 + kBW, go.x1), oy1 = min(oy0 + kBH, go.y1);
    const int now = ox1 - ox0, noh = oy1 - oy0;
    const int* __restrict__ cs0 = A.cd_s0 + T.tdc[l];
    const int* __restrict__ rs0 = A.cd_s0 + T.tdr[l];
    const float* __restrict__ cw = A.cd_w + (size_t)T.tdc[l] * NCP;
    const float* __restrict__ rw = A.cd_w + (size_t)T.tdr[l] * NCP;
    const int bx0 = cs0[ox0], nbc = cs0[ox1 - 1] + NC - bx0;
    if (nbc > VC) {
        if (tid == 0) atomicOr(A.status, 8);
        return;
    }
    const FootBuf fb = foot_buf(A.G + E.off[l], gi);
    {  // vertical pass: (column x of the patch, segment of RS output rows)
        const int x = tid % VC, seg = tid / VC, j0 = seg * RS;
        if (seg < NSEG && x < nbc && j0 < noh) {
            const int gx = bx0 + x, nj = min(RS, noh - j0);
            const int r0 = rs0[oy0 + j0];
            bool regular = true;
#pragma unroll
            for (int j = 1; j < RS; ++j) regular &= j >= nj || rs0[oy0 + j0 + min(j, nj - 1)] == r0 + 2 * j;
            if (regular) {
                float4 win[WIN];
#pragma unroll
                for (int i = 0; i < WIN; ++i) win[i] = ld_foot(fb, gx, r0 + i);
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    if (j < nj) {
                        float wr[NCP];
#pragma unroll
                        for (int q = 0; q < NCP / 4; ++q)
                            *reinterpret_cast<float4*>(&wr[4 * q]) = *reinterpret_cast<const float4*>(&rw[(size_t)(oy0 + j0 + j) * NCP + 4 * q]);
                        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int n = 0; n < NC; ++n) a = fma4(wr[n], win[2 * j + n], a);
                        s_v[(j0 + j) * VC + x] = a;
                    }
                }
            } else {
                for (int j = 0; j < nj; ++j) {
                    const int rr = rs0[oy0 + j0 + j];
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int n = 0; n < NC; ++n) a = fma4(rw[(size_t)(oy0 + j0 + j) * NCP + n], ld_foot(fb, gx, rr + n), a);
                    s_v[(j0 + j) * VC + x] = a;
                }
            }
        }
    }
    __syncthreads();
    const int xo = tid & (kBW - 1);
    if (xo < now && tid < 256) {  // horizontal pass: one output column per lane, rows 8 apart (the first 256 threads)
        const int ox = ox0 + xo, c0 = cs0[ox] - bx0;
        float wc[NCP];
#pragma unroll
        for (int q = 0; q < NCP / 4; ++q)
            *reinterpret_cast<float4*>(&wc[4 * q]) = *reinterpret_cast<const float4*>(&cw[(size_t)ox * NCP + 4 * q]);
        float4* __restrict__ out = A.G + E.off[l + 1];
        for (int yo = tid / kBW; yo < noh; yo += 256 / kBW) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int n = 0; n < NC; ++n) a = fma4(wc[n], s_v[yo * VC + c0 + n], a);
            out[(size_t)(oy0 + yo - go.y0) * gow + (ox - go.x0)] = a;
        }
    }
}

// One thread = kUP pixels of a column, kUH rows apart: the loads of a covering layer (its own value + four taps of the
// coarser level, per pixel) are in flight together.  With one pixel per thread the kernel sat at ~1.6 TB/s on
// back-to-back dependent waits (taps, then per covering layer, then F) of waves that had little else to do.
#ifndef APS_KUP
#define APS_KUP 2
#endif
constexpr int kUP = APS_KUP;
// (Measured and dropped in round 3: 8 waves per SIMD through amdgpu_waves_per_eu(8, 8) - 64 VGPRs with ten dwords of
// scratch: 4.2 -> 7.6 ms at level 0.)
// (Round 4, profiles/r04g_rw_up_staged_ab.txt: the coarser level's taps staged in LDS per 32 x 16 block - one barrier, 106
// VGPRs - 4.14 -> 6.24 ms; the resize order settled once per thread instead of the v_cndmask swaps in up_combine: +3 %.)
template <bool LEVEL0>
__global__ __launch_bounds__(256) void rw_up_kernel(RwArgs A, int l, const int* __restrict__ blk_ptr, int n_blocks) {
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int t = find_segment(blk_ptr, A.n_tiles, bid);
    const RwTile& T = A.tiles[t];
    const int tid = threadIdx.x;
    const int h = T.lh[l], w = T.lw[l];
    const bool last = l == T.nl - 1;
    const int local = bid - blk_ptr[t], nbx = (w + kUW - 1) / kUW;
    const int x0 = (local % nbx) * kUW, y0 = (local / nbx) * (kUH * kUP);
    const int x = x0 + (tid & (kUW - 1));
    if (x >= w || y0 + tid / kUW >= h) return;
    int y[kUP];       // rows past the level's end are computed on the last row and not stored
    bool live[kUP];
#pragma unroll
    for (int p = 0; p < kUP; ++p) {
        const int yy = y0 + tid / kUW + p * kUH;
        live[p] = yy < h;
        y[p] = min(yy, h - 1);
    }
    UpTaps u[kUP];
#pragma unroll
    for (int p = 0; p < kUP; ++p) {
        u[p].r0 = u[p].r1 = u[p].c0 = u[p].c1 = 0;
        u[p].wr0 = u[p].wr1 = u[p].wc0 = u[p].wc1 = 0.f;
    }
    if (!last) {
        const size_t co = ((size_t)T.tuc[l] + x) * kTU;
        const int c0 = A.tu_idx[co], c1 = A.tu_idx[co + 1];
        const float wc0 = A.tu_w[co], wc1 = A.tu_w[co + 1];
#pragma unroll
        for (int p = 0; p < kUP; ++p) {
            const size_t ro = ((size_t)T.tur[l] + y[p]) * kTU;
            u[p].r0 = A.tu_idx[ro];
            u[p].r1 = A.tu_idx[ro + 1];
            u[p].wr0 = A.tu_w[ro];
            u[p].wr1 = A.tu_w[ro + 1];
            u[p].c0 = c0;
            u[p].c1 = c1;
            u[p].wc0 = wc0;
            u[p].wc1 = wc1;
        }
    }
    const bool rf = (T.rf_up >> l) & 1;
    float acc[kUP][3];
#pragma unroll
    for (int p = 0; p < kUP; ++p) acc[p][0] = acc[p][1] = acc[p][2] = 0.f;
    // (Measured and dropped: walking a block-level layer mask as rw_warp does - no change; staging the entries' level-l
    // constants in LDS once per block behind a barrier - level 0 4.1 -> 4.8 ms, the other levels 1.45 -> 1.67.)
    for (int k = 0; k < T.ne; ++k) {
        const RwEntry& E = A.entries[T.e0 + k];
        const Rect g = E.g[l];
        bool in[kUP], any = false;
#pragma unroll
        for (int p = 0; p < kUP; ++p) {
            in[p] = in_rect(g, x, y[p]);
            any = any || in[p];
        }
        if (!any) continue;  // a layer contributes (0 - u) * 0 outside its footprint
        const FootBuf bg = foot_buf(A.G + E.off[l], g);
        float4 gv[kUP], uv[kUP];
#pragma unroll
        for (int p = 0; p < kUP; ++p) {
            gv[p] = ld_foot(bg, x, y[p]);
            uv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (!last) {
            const Rect gc = E.g[l + 1];
            if (gc.x1 > gc.x0) {
                const FootBuf bc = foot_buf(A.G + E.off[l + 1], gc);
#pragma unroll
                for (int p = 0; p < kUP; ++p) uv[p] = upsample2(u[p], rf, [&](int cx, int cy) { return ld_foot(bc, cx, cy); });
            }
        }
#pragma unroll
        for (int p = 0; p < kUP; ++p) {
            if (!in[p]) continue;
            if (!last) {
                acc[p][0] = acc[p][0] + (gv[p].x - uv[p].x) * gv[p].w;
                acc[p][1] = acc[p][1] + (gv[p].y - uv[p].y) * gv[p].w;
                acc[p][2] = acc[p][2] + (gv[p].z - uv[p].z) * gv[p].w;
            } else {
                acc[p][0] = acc[p][0] + gv[p].x * gv[p].w;
                acc[p][1] = acc[p][1] + gv[p].y * gv[p].w;
                acc[p][2] = acc[p][2] + gv[p].z * gv[p].w;
            }
        }
    }
    // (Requesting these taps and the coverage bytes ahead of the layer loop was measured: +30 registers, a wave less per
    // SIMD, 4.1 -> 4.7 ms.)
    float f[kUP][3];
#pragma unroll
    for (int p = 0; p < kUP; ++p) f[p][0] = acc[p][0], f[p][1] = acc[p][1], f[p][2] = acc[p][2];
    if (!last && T.ne > 0) {
        const float4* __restrict__ fc = A.F + T.f[l + 1];
        const int cw1 = T.lw[l + 1];
        UpVals fv[kUP];
#pragma unroll
        for (int p = 0; p < kUP; ++p) fv[p] = up_load(u[p], [&](int cx, int cy) { return fc[(size_t)cy * cw1 + cx]; });
#pragma unroll
        for (int p = 0; p < kUP; ++p) {
            const float4 fu = up_combine(u[p], rf, fv[p]);
            f[p][0] = fu.x + acc[p][0];
            f[p][1] = fu.y + acc[p][1];
            f[p][2] = fu.z + acc[p][2];
        }
    }
    bool cov[kUP];
    if (LEVEL0) {
#pragma unroll
        for (int p = 0; p < kUP; ++p) cov[p] = A.cov[(size_t)T.plane + (size_t)y[p] * w + x] != 0;
    }
#pragma unroll
    for (int p = 0; p < kUP; ++p) {
        if (!live[p]) continue;
        if (!LEVEL0) {
            A.F[T.f[l] + (size_t)y[p] * w + x] = make_float4(f[p][0], f[p][1], f[p][2], 0.f);
        } else {
            const bool c = cov[p];
            const int gy = T.r0 + y[p], gx = T.c0 + x;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float tt = roundf(255.0f * f[p][k]);  // MATLAB round: half away from zero
                if (!(tt > 0.f)) tt = 0.f;
                if (tt > 255.f) tt = 255.f;
                const uint8_t b = c ? (uint8_t)tt : (A.white ? 255 : 0);
                if (A.out_layout == APS_IMG_U8_HWC)
                    A.pano[((size_t)gy * A.W + gx) * 3 + k] = b;
                else
                    A.pano[(size_t)k * A.H * A.W + (size_t)gx * A.H + gy] = b;
            }
            if (A.covered) {
                if (A.out_layout == APS_IMG_U8_HWC)
                    A.covered[(size_t)gy * A.W + gx] = c ? 1 : 0;
                else
                    A.covered[(size_t)gx * A.H + gy] = c ? 1 : 0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 'linear' and 'none' blending, every tile in ONE launch (round 4; renderPanorama.m:864-978)
// ------------------------------------------------------------------------------------------------
// The per-tile path (render.hip) materialises one full-tile float4 layer per contributing image (warp_layer_kernel), folds
// the layers (linear_fuse_kernel) or visits every image of the set once per tile (none_fuse_kernel, n_img launches per tile)
// and paints: ~4 500 launches and 54 ms for the 64 x 4K scene.  Neither blend needs the layers afterwards, so here a
// thread owns one canvas pixel, computes its ray once, walks the layers that meet its 32 x 8 block in ascending image order
// (footprint rectangles, block masks as in the warp) and folds each sample straight into the pixel's accumulators - the
// per-tile kernels' expressions on the same values in the same order (sample_one, the 'linear' floor Wf >= 1e-4 and
// best-weight fallback of :931-975, the three 'none' policies of :876-911), then paints.  Nothing is stored but the
// panorama: the bytes equal the per-tile path's (tests/test_render_gpu.py).
template <int MODE>
__global__ __launch_bounds__(256) void rw_fuse_kernel(RwArgs A, const int* __restrict__ blk_ptr, int n_blocks, int none_policy) {
    __shared__ ColTrig s_ct[kUW];
    __shared__ RowTrig s_rt[kUH];
    const int bid = xcd_contiguous_id(n_blocks);
    if (bid >= n_blocks) return;
    const int t = find_segment(blk_ptr, A.n_tiles, bid);
    const RwTile& T = A.tiles[t];
    const int tid = threadIdx.x;
    const int h = T.ht, w = T.wt;
    const int local = bid - blk_ptr[t], nbx = (w + kUW - 1) / kUW;
    const int x0 = (local % nbx) * kUW, y0 = (local / nbx) * kUH;
    if (tid < kUW) s_ct[tid] = col_trig(A.cv, (float)(T.c0 + min(x0 + tid, w - 1)));
    if (tid >= 64 && tid < 64 + kUH) s_rt[tid - 64] = row_trig(A.cv, (float)(T.r0 + min(y0 + tid - 64, h - 1)));
    __syncthreads();
    const ColTrig ct = s_ct[tid & (kUW - 1)];
    const RowTrig rt = s_rt[tid / kUW];
    const int x = x0 + (tid & (kUW - 1)), y = y0 + tid / kUW;
    const bool in_tile = x < w && y < h;
    float d[3] = {0.f, 0.f, 1.f};
    if (in_tile) ray_from_tables(A.cv, ct, rt, (float)(T.c0 + x), (float)(T.r0 + y), d);  // (= canvas_ray, bit for bit)
    const int bx1 = min(x0 + kUW, w), by1 = min(y0 + kUH, h);
    const LayerSet ls = block_layers(A, T, 0, x0, y0, bx1, by1);
    float acc[3] = {0.f, 0.f, 0.f}, ws = 0.f, bestw = 0.f, best[3] = {0.f, 0.f, 0.f};  // 'linear' (linear_fuse_kernel)
    bool anyv = false;
    float f[3] = {0.f, 0.f, 0.f}, fw = 0.f;  // 'none' (none_fuse_kernel): F and, for 'maxangle', the best Wang so far
    bool cov = false;
    for_block_layers(ls, A, T, 0, x0, y0, bx1, by1, [&](int k) __attribute__((always_inline)) {
        const RwEntry& E = A.entries[T.e0 + k];
        if (!(in_tile && in_rect(E.g[0], x, y))) return;  // outside the footprint the sample is (0, 0, 0, mask false)
        const Sample sm = sample_one(A.imgs[E.img], d, A.angle_pow);
        if (MODE == APS_BLEND_LINEAR) {
            float wf = sm.wf;  // warp_layer_kernel with wf_floor = 1e-4 (:933)
            if (!isfinite(wf)) wf = 0.f;
            wf = wf > 1e-4f ? wf : 1e-4f;
            const float gw = sm.m ? sm.wang * wf : 0.0f;
            acc[0] = acc[0] + sm.s[0] * gw;
            acc[1] = acc[1] + sm.s[1] * gw;
            acc[2] = acc[2] + sm.s[2] * gw;
            ws = ws + gw;
            const bool m = gw > 0.f;
            anyv |= m;
            if (m && gw > bestw) {
                bestw = gw;
                best[0] = sm.s[0];
                best[1] = sm.s[1];
                best[2] = sm.s[2];
            }
        } else {
            bool upd;
            if (none_policy == APS_NONE_LAST)
                upd = sm.m;
            else if (none_policy == APS_NONE_FIRST)
                upd = sm.m && !cov;
            else
                upd = sm.m && (sm.wang > fw);
            if (upd) {
                f[0] = sm.s[0];
                f[1] = sm.s[1];
                f[2] = sm.s[2];
                if (none_policy == APS_NONE_MAXANGLE) fw = sm.wang;
                cov = true;
            }
        }
    });
    if (!in_tile) return;
    float v[3];
    bool c;
    if (MODE == APS_BLEND_LINEAR) {
        const bool z = ws > 1e-12f;
        v[0] = z ? acc[0] / ws : (anyv ? best[0] : 0.f);
        v[1] = z ? acc[1] / ws : (anyv ? best[1] : 0.f);
        v[2] = z ? acc[2] / ws : (anyv ? best[2] : 0.f);
        c = ws > 0.f;
    } else {
        v[0] = f[0];
        v[1] = f[1];
        v[2] = f[2];
        c = cov;
    }
    const int gy = T.r0 + y, gx = T.c0 + x;  // paint_kernel (:408-425)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float tt = roundf(255.0f * v[k]);  // MATLAB round: half away from zero
        if (!(tt > 0.f)) tt = 0.f;
        if (tt > 255.f) tt = 255.f;
        const uint8_t b = c ? (uint8_t)tt : (A.white ? 255 : 0);
        if (A.out_layout == APS_IMG_U8_HWC)
            A.pano[((size_t)gy * A.W + gx) * 3 + k] = b;
        else
            A.pano[(size_t)k * A.H * A.W + (size_t)gx * A.H + gy] = b;
    }
    if (A.covered) {
        if (A.out_layout == APS_IMG_U8_HWC)
            A.covered[(size_t)gy * A.W + gx] = c ? 1 : 0;
        else
            A.covered[(size_t)gx * A.H + gy] = c ? 1 : 0;
    }
}

template <int R>
constexpr size_t down_lds_bytes() {
    return (size_t)(kBR * DownShape<R>::IC + DownShape<R>::T1) * sizeof(float4);
}
template <int R>
void launch_down_r(const RwArgs& A, int l, const int* blk_ptr, int n_blocks) {
    static bool once = [] {
        APS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rw_down_kernel<R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)down_lds_bytes<R>()));
        return true;
    }();
    (void)once;
    rw_down_kernel<R><<<8u * (unsigned)((n_blocks + 7) / 8), kNT, down_lds_bytes<R>(), stream()>>>(A, l, blk_ptr, n_blocks);
}
template <int R>
void launch_down_fused_r(const RwArgs& A, int l, const int* blk_ptr, int n_blocks) {
    rw_down_fused_kernel<R><<<8u * (unsigned)((n_blocks + 7) / 8), kDFT, 0, stream()>>>(A, l, blk_ptr, n_blocks);
}
void launch_down(int r, const RwArgs& A, int l, const int* blk_ptr, int n_blocks, bool exact) {
    if (!exact) {
        switch (r) {
            case 1: launch_down_fused_r<1>(A, l, blk_ptr, n_blocks); break;
            case 2: launch_down_fused_r<2>(A, l, blk_ptr, n_blocks); break;
            case 3: launch_down_fused_r<3>(A, l, blk_ptr, n_blocks); break;
            default: launch_down_fused_r<4>(A, l, blk_ptr, n_blocks); break;
        }
        return;
    }
    switch (r) {
        case 1: launch_down_r<1>(A, l, blk_ptr, n_blocks); break;
        case 2: launch_down_r<2>(A, l, blk_ptr, n_blocks); break;
        case 3: launch_down_r<3>(A, l, blk_ptr, n_blocks); break;
        default: launch_down_r<4>(A, l, blk_ptr, n_blocks); break;
    }
}

}  // namespace

namespace {

// Footprints on the host (cylindrical / spherical canvases): the image rectangle [1,w] x [1,h] maps to a region of the
// canvas whose outline is the forward image of the rectangle's border (pixel -> K^-1 -> R' -> angles -> canvas pixels,
// f64 on the f32 camera / canvas values the kernels use).  The outline, sampled in short chords and grown by half a pixel,
// is clipped against each tile (Sutherland-Hodgman) and the clipped polygon's bounding box, padded by 3 pixels (chord sag
// < 0.1, f32 vs f64 rounding < 0.01), is a SUPERSET of the tile pixels that sample the image - which is all a footprint must
// be (kernels evaluate the exact predicate per pixel inside it).  This replaces the coverage kernel and its read-back.
// Not applicable (returns false, the caller runs rw_cover_kernel): planar / stereographic canvases, an outline that
// crosses the theta = +-pi seam, an image that contains a pole.
// The images are independent (image i writes hbox[(t * n_img + i) * 4 ..] only), so the loop runs on a few host threads
// (round 6: 0.65 ms of every render call on one thread, in front of the call's first kernel launch - most of a rank's fixed
// render cost at 8 ranks).  APS_RENDER_HOST_THREADS=1: one thread.
static bool host_footprints_range(const DevImage* himgs, int n_img, int i_begin, int i_end, const DevCanvas& cv, const std::vector<RwTile>& tiles,
                                  std::vector<int>& hbox);
bool host_footprints(const DevImage* himgs, int n_img, const DevCanvas& cv, const std::vector<RwTile>& tiles, std::vector<int>& hbox) {
    if (cv.mode != APS_PROJ_SPHERICAL && cv.mode != APS_PROJ_CYLINDRICAL) return false;
    std::fill(hbox.begin(), hbox.end(), 0);
    const char* e = std::getenv("APS_RENDER_HOST_THREADS");
    int nthr = e ? std::atoi(e) : 6;
    nthr = std::max(1, std::min({nthr, n_img / 8, (int)std::thread::hardware_concurrency()}));
    if (nthr <= 1) return host_footprints_range(himgs, n_img, 0, n_img, cv, tiles, hbox);
    std::vector<std::future<bool>> parts;
    for (int k = 1; k < nthr; ++k)
        parts.push_back(std::async(std::launch::async, [&, k] {
            return host_footprints_range(himgs, n_img, (int)((long long)n_img * k / nthr), (int)((long long)n_img * (k + 1) / nthr), cv, tiles, hbox);
        }));
    bool ok = host_footprints_range(himgs, n_img, 0, n_img / nthr, cv, tiles, hbox);
    for (auto& p : parts) ok = p.get() && ok;
    return ok;
}
static bool host_footprints_range(const DevImage* himgs, int n_img, int i_begin, int i_end, const DevCanvas& cv, const std::vector<RwTile>& tiles,
                                  std::vector<int>& hbox) {
    const double f = (double)cv.f, o0 = (double)cv.o0, o1 = (double)cv.o1;
    const int nt = (int)tiles.size();
    std::vector<double> px, py, qx, qy, rx, ry;
    const double kPi = 3.14159265358979323846;
    for (int i = i_begin; i < i_end; ++i) {
        const DevImage& im = himgs[i];
        if (!(im.fx > 0.f) || !(im.fy > 0.f)) return false;
        const double R[9] = {im.R[0], im.R[1], im.R[2], im.R[3], im.R[4], im.R[5], im.R[6], im.R[7], im.R[8]};
        auto to_canvas = [&](double u, double v, double& x, double& y, double& th) {
            const double c[3] = {(u - im.cx) / im.fx, (v - im.cy) / im.fy, 1.0};
            // cam = R d  (device: cam[c] = d0 R[c] + d1 R[c+3] + d2 R[c+6])  =>  d = R' cam: d[j] = sum_c R[c + 3j] cam[c]
            const double d0 = R[0] * c[0] + R[1] * c[1] + R[2] * c[2];
            const double d1 = R[3] * c[0] + R[4] * c[1] + R[5] * c[2];
            const double d2 = R[6] * c[0] + R[7] * c[1] + R[8] * c[2];
            th = std::atan2(d0, d2);
            const double hz = std::hypot(d0, d2);
            const double b = cv.mode == APS_PROJ_SPHERICAL ? std::atan2(d1, hz) : d1 / hz;
            x = (th - o0) * f;
            y = (b - o1) * f;
        };
        // a pole inside the image: the elevation has an interior extremum, the outline does not bound the region
        for (int sgn = -1; sgn <= 1; sgn += 2) {
            const double cz = R[5] * sgn;  // cam = R (0, sgn, 0)'
            if (cz > 1e-9) {
                const double u = im.fx * (R[3] * sgn / cz) + im.cx, v = im.fy * (R[4] * sgn / cz) + im.cy;
                if (u >= -1 && u <= im.w + 2 && v >= -1 && v <= im.h + 2) return false;
            }
        }
        px.clear();
        py.clear();
        // chord length ~ 1/64 of the shorter side (at least 4 px): the sag stays far below the 3-pixel pad
        const double u0 = 0.5, u1 = im.w + 0.5, v0 = 0.5, v1 = im.h + 0.5, step = std::max(4.0, std::min(im.w, im.h) / 64.0);
        double prev_th = 0;
        bool first = true, ok = true;
        auto add = [&](double u, double v) {
            double x, y, th;
            to_canvas(u, v, x, y, th);
            if (!std::isfinite(x) || !std::isfinite(y)) ok = false;
            if (!first && std::fabs(th - prev_th) > 0.5 * kPi) ok = false;  // seam crossing (or a degenerate camera)
            prev_th = th;
            first = false;
            px.push_back(x);
            py.push_back(y);
        };
        for (double u = u0; u < u1; u += step) add(u, v0);
        for (double v = v0; v < v1; v += step) add(u1, v);
        for (double u = u1; u > u0; u -= step) add(u, v1);
        for (double v = v1; v > v0; v -= step) add(u0, v);
        add(u0, v0);
        if (!ok) return false;
        double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
        for (size_t q = 0; q < px.size(); ++q) {
            bx0 = std::min(bx0, px[q]);
            bx1 = std::max(bx1, px[q]);
            by0 = std::min(by0, py[q]);
            by1 = std::max(by1, py[q]);
        }
        for (int t = 0; t < nt; ++t) {
            const RwTile& T = tiles[t];
            const double tx0 = T.c0 - 0.5, tx1 = T.c0 + T.wt - 0.5, ty0 = T.r0 - 0.5, ty1 = T.r0 + T.ht - 0.5;
            if (bx1 < tx0 || bx0 > tx1 || by1 < ty0 || by0 > ty1) continue;
            // Sutherland-Hodgman against the four sides of the tile
            qx = px;
            qy = py;
            for (int side = 0; side < 4 && !qx.empty(); ++side) {
                rx.clear();
                ry.clear();
                const size_t n = qx.size();
                auto inside = [&](double x, double y) {
                    return side == 0 ? x >= tx0 : side == 1 ? x <= tx1 : side == 2 ? y >= ty0 : y <= ty1;
                };
                for (size_t q = 0; q < n; ++q) {
                    const double ax = qx[q], ay = qy[q], bx = qx[(q + 1) % n], by = qy[(q + 1) % n];
                    const bool ia = inside(ax, ay), ib = inside(bx, by);
                    if (ia != ib) {
                        double tt;
                        if (side < 2) {
                            const double lim = side == 0 ? tx0 : tx1;
                            tt = (lim - ax) / (bx - ax);
                            rx.push_back(lim);
                            ry.push_back(ay + tt * (by - ay));
                        } else {
                            const double lim = side == 2 ? ty0 : ty1;
                            tt = (lim - ay) / (by - ay);
                            rx.push_back(ax + tt * (bx - ax));
                            ry.push_back(lim);
                        }
                    }
                    if (ib) {
                        rx.push_back(bx);
                        ry.push_back(by);
                    }
                }
                qx.swap(rx);
                qy.swap(ry);
            }
            if (qx.empty()) continue;
            double cx0 = 1e300, cx1 = -1e300, cy0 = 1e300, cy1 = -1e300;
            for (size_t q = 0; q < qx.size(); ++q) {
                cx0 = std::min(cx0, qx[q]);
                cx1 = std::max(cx1, qx[q]);
                cy0 = std::min(cy0, qy[q]);
                cy1 = std::max(cy1, qy[q]);
            }
            const double m = 3.0;
            int* b = &hbox[((size_t)t * n_img + i) * 4];
            b[0] = std::max(0, (int)std::floor(cx0 - m) - T.c0);
            b[1] = std::max(0, (int)std::floor(cy0 - m) - T.r0);
            b[2] = std::min(T.wt, (int)std::ceil(cx1 + m) + 1 - T.c0);
            b[3] = std::min(T.ht, (int)std::ceil(cy1 + m) + 1 - T.r0);
            if (b[2] <= b[0] || b[3] <= b[1]) b[0] = b[1] = b[2] = b[3] = 0;
        }
    }
    return true;
}

// pass 1 of both batched renderers: the footprint rectangle of every image in every tile, on the host where the
// projection allows it (no kernel, no read-back), else by the exact coverage kernel.  hbox: [tile][image][x0 y0 x1 y1].
static void tile_footprints(const RwArgs& A, const DevImage* himgs, int n_img, const std::vector<RwTile>& ht_, const std::vector<int>& xshift,
                            const int* d_xshift, const RwTile* d_tiles, int max_ht, int max_wt, std::vector<int>& hbox) {
    const int nt = (int)ht_.size();
    const DevCanvas& cv = A.cv;
    hbox.assign((size_t)nt * n_img * 4, 0);
    const bool check_rects = std::getenv("APS_RENDER_CHECK_RECTS") != nullptr;
    const bool analytic = !std::getenv("APS_RENDER_DEVICE_COVER") && host_footprints(himgs, n_img, cv, ht_, hbox);
    if (!analytic || check_rects) {
        const int nby_max = cdiv(max_ht, 8);
        Ws<unsigned long long> rowmask((size_t)nt * n_img * nby_max);
        Ws<int> d_bbox((size_t)nt * n_img * 4);
        std::vector<int> dbox((size_t)nt * n_img * 4);
        APS_HIP(hipMemsetAsync(rowmask, 0, (size_t)nt * n_img * nby_max * sizeof(unsigned long long), stream()));
        {
            Prof prof("render_cover");
            rw_cover_kernel<<<dim3(cdiv(max_wt, 32), cdiv(max_ht, 8), nt), 256, 0, stream()>>>(A, n_img, nby_max, d_xshift, rowmask);
            rw_footprint_kernel<<<cdiv((size_t)nt * n_img, 256), 256, 0, stream()>>>(rowmask, d_tiles, d_xshift, n_img, nby_max,
                                                                                    nt * n_img, d_bbox);
        }
        check_launch("rw_cover_kernel");
        APS_HIP(hipMemcpyAsync(dbox.data(), d_bbox, dbox.size() * sizeof(int), hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        if (analytic) {  // test hook: every exact footprint must lie inside its analytic rectangle
            for (size_t e = 0; e < dbox.size() / 4; ++e) {
                const int* d = &dbox[4 * e];
                const int* hb = &hbox[4 * e];
                if (!(d[2] > d[0] && d[3] > d[1])) continue;
                // the coverage kernel reports whole 32 x 8 blocks (32 << xshift wide): compare on that grid
                const int t_ = (int)(e / n_img), gx = 32 << xshift[t_];
                const bool inside = hb[2] > hb[0] && hb[0] / gx * gx <= d[0] && hb[1] / 8 * 8 <= d[1] &&
                                    std::min((hb[2] + gx - 1) / gx * gx, ht_[t_].wt) >= d[2] && std::min((hb[3] + 7) / 8 * 8, ht_[t_].ht) >= d[3];
                APS_REQUIRE(inside, APS_E_INTERNAL,
                            "analytic footprint [%d %d %d %d] does not contain the exact one [%d %d %d %d] (tile %zu, image %zu)",
                            hb[0], hb[1], hb[2], hb[3], d[0], d[1], d[2], d[3], e / n_img, e % n_img);
            }
        } else {
            hbox = dbox;
        }
    }

}

}  // namespace

bool render_multiband_batched(const DevImage* dimgs, const DevImage* himgs, int n_img, const DevCanvas& cv, const aps_render_opts& o,
                              const std::vector<TileRect>& tiles, int out_layout, uint8_t* pano, uint8_t* covered,
                              const NeedImages& need_images) {
    const int nt = (int)tiles.size();
    if (nt == 0) return true;
    const bool trace = std::getenv("APS_TRACE") != nullptr;  // host phases of this call on stderr
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t_ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto T0 = t_now();
    const Taps tp = make_taps(o.pyr_sigma);
    APS_REQUIRE(tp.r >= 1 && tp.r <= 4, APS_E_ARG, "pyrSigma %g needs a %d-tap filter; 3..9 taps are built",
                (double)o.pyr_sigma, 2 * tp.r + 1);
    if (o.pyr_levels > kML) return false;
    for (const TileRect& tr : tiles)  // footprint stores are addressed with 32-bit byte offsets
        if ((long long)tr.ht * tr.wt * 16 >= (1ll << 31)) return false;
    // ---- tiles, level sizes, tap tables --------------------------------------------------------------------
    std::vector<RwTile> ht_(nt);
    std::vector<TapJob> jobs;
    std::map<std::pair<int, int>, int> down_off, up_off;
    int n_down = 0, n_up = 0;
    auto tap_table = [&](int in_len, int out_len, bool down) {
        auto& m = down ? down_off : up_off;
        int& total = down ? n_down : n_up;
        auto it = m.find({in_len, out_len});
        if (it != m.end()) return it->second;
        const int off = total;
        m[{in_len, out_len}] = off;
        jobs.push_back(TapJob{in_len, out_len, down ? kTD : kTU, off});
        total += out_len;
        return off;
    };
    long long plane_px = 0, f_total = 0;
    int max_ht = 0, max_wt = 0;
    std::vector<int> xshift(nt);
    for (int t = 0; t < nt; ++t) {
        RwTile& T = ht_[t];
        std::memset(&T, 0, sizeof T);
        T.r0 = tiles[t].r0;
        T.c0 = tiles[t].c0;
        T.ht = tiles[t].ht;
        T.wt = tiles[t].wt;
        max_ht = std::max(max_ht, T.ht);
        max_wt = std::max(max_wt, T.wt);
        const int maxl = (int)std::floor(std::log2((double)std::min(T.ht, T.wt)));  // multiBandBlending.m:98-109
        T.nl = std::max(1, std::min(o.pyr_levels, maxl));
        T.lh[0] = T.ht;
        T.lw[0] = T.wt;
        for (int l = 1; l < T.nl; ++l) {
            T.lh[l] = std::max(1, T.lh[l - 1] / 2);
            T.lw[l] = std::max(1, T.lw[l - 1] / 2);
        }
        for (int l = 0; l + 1 < T.nl; ++l) {
            T.tdr[l] = tap_table(T.lh[l], T.lh[l + 1], true);
            T.tdc[l] = tap_table(T.lw[l], T.lw[l + 1], true);
            T.tur[l] = tap_table(T.lh[l + 1], T.lh[l], false);
            T.tuc[l] = tap_table(T.lw[l + 1], T.lw[l], false);
            if (rows_first(T.lh[l], T.lw[l], T.lh[l + 1], T.lw[l + 1])) T.rf_down |= 1 << l;
            if (rows_first(T.lh[l + 1], T.lw[l + 1], T.lh[l], T.lw[l])) T.rf_up |= 1 << l;
        }
        T.plane = plane_px;
        plane_px += (long long)T.ht * T.wt;
        for (int l = 1; l < T.nl; ++l) {
            T.f[l] = f_total;
            f_total += (long long)T.lh[l] * T.lw[l];
        }
        int xs = 0;
        while ((cdiv(T.wt, 32) >> xs) > 64 || (((cdiv(T.wt, 32) - 1) >> xs) > 63)) ++xs;
        xshift[t] = xs;
    }
    Ws<int> d_status(1), d_xshift(nt);
    APS_HIP(hipMemsetAsync(d_status, 0, sizeof(int), stream()));
    APS_HIP(hipMemcpyAsync(d_xshift, xshift.data(), nt * sizeof(int), hipMemcpyHostToDevice, stream()));
    Ws<int> td_idx((size_t)std::max(n_down, 1) * kTD), tu_idx((size_t)std::max(n_up, 1) * kTU);
    Ws<float> td_w((size_t)std::max(n_down, 1) * kTD), tu_w((size_t)std::max(n_up, 1) * kTU);
    Ws<TapJob> d_jobs(std::max<size_t>(jobs.size(), 1));
    if (!jobs.empty()) {
        APS_HIP(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(TapJob), hipMemcpyHostToDevice, stream()));
        int longest = 1;
        for (const TapJob& j : jobs) longest = std::max(longest, j.out_len);
        rw_taps_kernel<<<dim3(cdiv(longest, 256), (unsigned)jobs.size()), 256, 0, stream()>>>(d_jobs, td_idx, td_w, tu_idx, tu_w,
                                                                                              d_status);
        check_launch("rw_taps_kernel");
    }
    // APS_RENDER_EXACT=1 (or its older name APS_WARP_EXACT): the warp and the shrink keep the per-tile path's arithmetic
    const bool exact = std::getenv("APS_RENDER_EXACT") != nullptr || std::getenv("APS_WARP_EXACT") != nullptr;
    const int ncp = (kTD + 2 * tp.r + 3) & ~3;
    Ws<int> cd_s0((size_t)std::max(n_down, 1));
    Ws<float> cd_w((size_t)std::max(n_down, 1) * ncp);
    if (!jobs.empty() && !exact) {
        int longest = 1;
        for (const TapJob& j : jobs) longest = std::max(longest, j.out_len);
        rw_ctaps_kernel<<<dim3(cdiv(longest, 256), (unsigned)jobs.size()), 256, 0, stream()>>>(d_jobs, tp, ncp, cd_s0, cd_w, d_status);
        check_launch("rw_ctaps_kernel");
    }
    Ws<RwTile> d_tiles(nt);
    APS_HIP(hipMemcpyAsync(d_tiles, ht_.data(), nt * sizeof(RwTile), hipMemcpyHostToDevice, stream()));
    Ws<uint8_t> d_cov((size_t)plane_px);
    APS_HIP(hipMemsetAsync(d_cov, 0, (size_t)plane_px, stream()));  // tiles without layers stay uncovered

    RwArgs A;
    std::memset(&A, 0, sizeof A);
    A.tiles = d_tiles;
    A.n_tiles = nt;
    A.imgs = dimgs;
    A.cv = cv;
    A.angle_pow = o.angle_power;
    A.tp = tp;
    A.td_idx = td_idx;
    A.td_w = td_w;
    A.tu_idx = tu_idx;
    A.tu_w = tu_w;
    A.cd_s0 = cd_s0;
    A.cd_w = cd_w;
    A.cov = d_cov;
    A.status = d_status;
    A.pano = pano;
    A.covered = covered;
    A.H = cv.H;
    A.W = cv.W;
    A.out_layout = out_layout;
    A.white = o.canvas_white;
    Ws<ColTrig> d_ct((size_t)std::max(cv.W, 1));
    Ws<RowTrig> d_rt((size_t)std::max(cv.H, 1));
    rw_trig_kernel<<<cdiv(std::max(cv.W, cv.H), 256), 256, 0, stream()>>>(cv, d_ct, d_rt);
    check_launch("rw_trig_kernel");
    A.ct = d_ct;
    A.rt = d_rt;

    // ---- pass 1: the footprint rectangle of every image in every tile ----------------------------------------------
    const auto T1 = t_now();
    std::vector<int> hbox;
    tile_footprints(A, himgs, n_img, ht_, xshift, d_xshift, d_tiles, max_ht, max_wt, hbox);
    const auto T2 = t_now();

    // ---- entries, compact stores, block tables ----------------------------------------------------------------
    std::vector<RwEntry> ents;
    long long g_total = 0;
    int max_nl = 1;
    for (int t = 0; t < nt; ++t) {
        RwTile& T = ht_[t];
        T.e0 = (int)ents.size();
        max_nl = std::max(max_nl, T.nl);
        for (int i = 0; i < n_img; ++i) {
            const int* b = &hbox[((size_t)t * n_img + i) * 4];
            if (!(b[2] > b[0] && b[3] > b[1])) continue;
            RwEntry E;
            std::memset(&E, 0, sizeof E);
            E.tile = t;
            E.img = i;
            E.g[0] = clip_rect(Rect{b[0], b[1], b[2], b[3]}, T.wt, T.ht);
            for (int l = 0; l + 1 < T.nl; ++l) {
                const Rect g = E.g[l];
                const bool empty = g.x1 <= g.x0;
                const Rect br = empty ? g : clip_rect(Rect{g.x0 - tp.r, g.y0 - tp.r, g.x1 + tp.r, g.y1 + tp.r}, T.lw[l], T.lh[l]);
                E.g[l + 1] = empty ? g : map_rect(br, T.lh[l], T.lw[l], T.lh[l + 1], T.lw[l + 1]);
            }
            for (int l = 0; l < T.nl; ++l) {
                E.off[l] = g_total;
                g_total += (long long)(E.g[l].x1 - E.g[l].x0) * (E.g[l].y1 - E.g[l].y0);
            }
            ents.push_back(E);
        }
        T.ne = (int)ents.size() - T.e0;
    }
    const int ne = (int)ents.size();
    {  // the images these tiles meet: their pixels are needed from here on
        std::vector<char> used(n_img, 0);
        for (const RwEntry& E : ents) used[E.img] = 1;
        need_images(used);
    }
    if (std::getenv("APS_RENDER_DEBUG")) {
        long long rect0 = 0, canvas = 0;
        for (const RwEntry& E : ents) rect0 += (long long)(E.g[0].x1 - E.g[0].x0) * (E.g[0].y1 - E.g[0].y0);
        for (const RwTile& T : ht_) canvas += (long long)T.ht * T.wt;
        std::fprintf(stderr, "[aps render] tiles %d entries %d (%.2f per tile) footprint rect area %.1f MPix canvas %.1f MPix G store %.2f GB\n",
                     nt, ne, (double)ne / nt, rect0 / 1e6, canvas / 1e6, g_total * 16.0 / 1e9);
    }
    // block tables: levels 0 .. max_nl-2 of the shrink over entries, levels 0 .. max_nl-1 of the collapse over tiles
    std::vector<int> blk((size_t)(max_nl) * (ne + 1) + (size_t)max_nl * (nt + 1), 0);
    std::vector<int> down_blocks(max_nl, 0), up_blocks(max_nl, 0);
    // the warp: blocks over the level-0 grid of every tile with layers
    std::vector<int> blk0(nt + 1, 0);
    int warp_blocks = 0;
    {
        long long run = 0;
        for (int t = 0; t < nt; ++t) {
            blk0[t] = (int)run;
            const RwTile& T = ht_[t];
            if (T.ne > 0) run += (long long)cdiv(T.wt, kUW) * cdiv(T.ht, kUH);
        }
        APS_REQUIRE(run < (1ll << 30), APS_E_DIM, "too many canvas blocks in one render call (%lld)", run);
        blk0[nt] = (int)run;
        warp_blocks = (int)run;
    }
    for (int l = 0; l + 1 < max_nl; ++l) {
        int* p = &blk[(size_t)l * (ne + 1)];
        long long run = 0;
        for (int e = 0; e < ne; ++e) {
            p[e] = (int)run;
            const RwEntry& E = ents[e];
            if (l + 1 < ht_[E.tile].nl) {
                const Rect& g = E.g[l + 1];
                run += (long long)cdiv(std::max(g.x1 - g.x0, 0), kBW) * cdiv(std::max(g.y1 - g.y0, 0), kBH);
            }
        }
        APS_REQUIRE(run < (1ll << 30), APS_E_DIM, "too many pyramid blocks in one render call (%lld)", run);
        p[ne] = (int)run;
        down_blocks[l] = (int)run;
    }
    const size_t up_base = (size_t)max_nl * (ne + 1);
    for (int l = 0; l < max_nl; ++l) {
        int* p = &blk[up_base + (size_t)l * (nt + 1)];
        long long run = 0;
        for (int t = 0; t < nt; ++t) {
            p[t] = (int)run;
            const RwTile& T = ht_[t];
            // level 0 always paints; the upper levels exist only for tiles with layers
            if (l < T.nl && (l == 0 || T.ne > 0)) run += (long long)cdiv(T.lw[l], kUW) * cdiv(T.lh[l], kUH * kUP);
        }
        APS_REQUIRE(run < (1ll << 30), APS_E_DIM, "too many canvas blocks in one render call (%lld)", run);
        p[nt] = (int)run;
        up_blocks[l] = (int)run;
    }
    const auto T3 = t_now();
    Ws<RwEntry> d_ents(std::max(ne, 1));
    Ws<int> d_blk(blk.size()), d_blk0(blk0.size());
    APS_HIP(hipMemcpyAsync(d_blk0, blk0.data(), blk0.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
    Ws<float4> d_G((size_t)std::max<long long>(g_total, 1)), d_F((size_t)std::max<long long>(f_total, 1));
    if (ne) APS_HIP(hipMemcpyAsync(d_ents, ents.data(), ne * sizeof(RwEntry), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_tiles, ht_.data(), nt * sizeof(RwTile), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_blk, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
    A.entries = d_ents;
    A.n_entries = ne;
    A.G = d_G;
    A.F = d_F;

    // ---- pass 2: pyramids, fine to coarse ------------------------------------------------------------------------
    if (warp_blocks) {
        Prof prof("render_warp");
        const unsigned wgrid = 8u * (unsigned)((warp_blocks + 7) / 8);
        const char* wv = std::getenv("APS_WARP_VARIANT");  // experiment switch, see rw_warp_kernel
        const int variant = wv ? std::atoi(wv) : 6;  // (measured: per-block LDS trig tables 6.97 ms, canvas-wide global ones 7.39)
        if (exact)
            rw_warp_kernel<false><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else if (!wv)
            rw_warp_staged_kernel<<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else if (variant == 6)
            rw_warp_kernel<true, 6><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else if (variant == 5)
            rw_warp_kernel<true, 5><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else if (variant == 3)
            rw_warp_kernel<true, 3><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else if (variant == 0)
            rw_warp_kernel<true, 0><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        else
            rw_warp_kernel<true, 7><<<wgrid, 256, 0, stream()>>>(A, d_blk0, warp_blocks);
        check_launch("rw_warp_kernel");
    }
    {
        Prof prof("render_pyr_down");
        for (int l = 0; l + 1 < max_nl; ++l) {
            if (down_blocks[l] == 0) continue;
            launch_down(tp.r, A, l, d_blk.get() + (size_t)l * (ne + 1), down_blocks[l], exact);
            check_launch("rw_down_kernel");
        }
    }
    {
        // Laplacian sums + collapse, coarse to fine; level 0 paints
        Prof prof("render_collapse");
        for (int l = max_nl - 1; l >= 0; --l) {
            if (up_blocks[l] == 0) continue;
            const int* bp = d_blk.get() + up_base + (size_t)l * (nt + 1);
            const unsigned grid = 8u * (unsigned)((up_blocks[l] + 7) / 8);
            if (l == 0)
                rw_up_kernel<true><<<grid, 256, 0, stream()>>>(A, l, bp, up_blocks[l]);
            else
                rw_up_kernel<false><<<grid, 256, 0, stream()>>>(A, l, bp, up_blocks[l]);
            check_launch("rw_up_kernel");
        }
    }
    const auto T4 = t_now();
    int status = 0;
    APS_HIP(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, stream()));
    APS_HIP(hipStreamSynchronize(stream()));  // also keeps the host tables alive until their uploads have run
    if (trace)
        std::fprintf(stderr, "[aps render] host: tiles + tap tables %.2f ms, footprints %.2f, entries + block tables %.2f, uploads + launches %.2f, wait for the kernels %.2f\n",
                     t_ms(T0, T1), t_ms(T1, T2), t_ms(T2, T3), t_ms(T3, T4), t_ms(T4, t_now()));
    APS_REQUIRE(status == 0, APS_E_INTERNAL, "batched render: tap table assumption violated (status %d)", status);
    return true;
}

// 'linear' / 'none': all tiles in one launch (rw_fuse_kernel).  Same tile list and footprints as the multiband form.
bool render_fuse_batched(const DevImage* dimgs, const DevImage* himgs, int n_img, const DevCanvas& cv, const aps_render_opts& o,
                         const std::vector<TileRect>& tiles, int out_layout, uint8_t* pano, uint8_t* covered,
                         const NeedImages& need_images) {
    const int nt = (int)tiles.size();
    if (nt == 0) return true;
    if (o.blending != APS_BLEND_LINEAR && o.blending != APS_BLEND_NONE) return false;
    std::vector<RwTile> ht_(nt);
    std::vector<int> xshift(nt);
    int max_ht = 0, max_wt = 0;
    for (int t = 0; t < nt; ++t) {
        RwTile& T = ht_[t];
        std::memset(&T, 0, sizeof T);
        T.r0 = tiles[t].r0;
        T.c0 = tiles[t].c0;
        T.ht = tiles[t].ht;
        T.wt = tiles[t].wt;
        T.nl = 1;
        T.lh[0] = T.ht;
        T.lw[0] = T.wt;
        max_ht = std::max(max_ht, T.ht);
        max_wt = std::max(max_wt, T.wt);
        int xs = 0;
        while ((cdiv(T.wt, 32) >> xs) > 64 || (((cdiv(T.wt, 32) - 1) >> xs) > 63)) ++xs;
        xshift[t] = xs;
    }
    Ws<int> d_xshift(nt);
    Ws<RwTile> d_tiles(nt);
    APS_HIP(hipMemcpyAsync(d_xshift, xshift.data(), nt * sizeof(int), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_tiles, ht_.data(), nt * sizeof(RwTile), hipMemcpyHostToDevice, stream()));
    RwArgs A;
    std::memset(&A, 0, sizeof A);
    A.tiles = d_tiles;
    A.n_tiles = nt;
    A.imgs = dimgs;
    A.cv = cv;
    A.angle_pow = o.angle_power;
    A.pano = pano;
    A.covered = covered;
    A.H = cv.H;
    A.W = cv.W;
    A.out_layout = out_layout;
    A.white = o.canvas_white;
    std::vector<int> hbox;
    tile_footprints(A, himgs, n_img, ht_, xshift, d_xshift, d_tiles, max_ht, max_wt, hbox);
    std::vector<RwEntry> ents;
    for (int t = 0; t < nt; ++t) {
        RwTile& T = ht_[t];
        T.e0 = (int)ents.size();
        for (int i = 0; i < n_img; ++i) {
            const int* b = &hbox[((size_t)t * n_img + i) * 4];
            if (!(b[2] > b[0] && b[3] > b[1])) continue;
            RwEntry E;
            std::memset(&E, 0, sizeof E);
            E.tile = t;
            E.img = i;
            E.g[0] = clip_rect(Rect{b[0], b[1], b[2], b[3]}, T.wt, T.ht);
            ents.push_back(E);
        }
        T.ne = (int)ents.size() - T.e0;
    }
    {
        std::vector<char> used(n_img, 0);
        for (const RwEntry& E : ents) used[E.img] = 1;
        need_images(used);
    }
    // every tile is painted (a tile without layers gets the canvas colour): blocks over the level-0 grid of all tiles
    std::vector<int> blk0(nt + 1, 0);
    long long run = 0;
    for (int t = 0; t < nt; ++t) {
        blk0[t] = (int)run;
        run += (long long)cdiv(ht_[t].wt, kUW) * cdiv(ht_[t].ht, kUH);
    }
    APS_REQUIRE(run < (1ll << 30), APS_E_DIM, "too many canvas blocks in one render call (%lld)", run);
    blk0[nt] = (int)run;
    const int ne = (int)ents.size();
    Ws<RwEntry> d_ents(std::max(ne, 1));
    Ws<int> d_blk0(blk0.size());
    if (ne) APS_HIP(hipMemcpyAsync(d_ents, ents.data(), ne * sizeof(RwEntry), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_tiles, ht_.data(), nt * sizeof(RwTile), hipMemcpyHostToDevice, stream()));
    APS_HIP(hipMemcpyAsync(d_blk0, blk0.data(), blk0.size() * sizeof(int), hipMemcpyHostToDevice, stream()));
    A.entries = d_ents;
    A.n_entries = ne;
    {
        Prof prof("render_fuse");
        const unsigned grid = 8u * (unsigned)((run + 7) / 8);
        if (o.blending == APS_BLEND_LINEAR)
            rw_fuse_kernel<APS_BLEND_LINEAR><<<grid, 256, 0, stream()>>>(A, d_blk0, (int)run, o.none_policy);
        else
            rw_fuse_kernel<APS_BLEND_NONE><<<grid, 256, 0, stream()>>>(A, d_blk0, (int)run, o.none_policy);
    }
    check_launch("rw_fuse_kernel");
    APS_HIP(hipStreamSynchronize(stream()));  // keeps the host tables alive until their uploads have run
    return true;
}

}  // namespace aps
