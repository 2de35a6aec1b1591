// render_dev.h — device-side structures and sampling functions shared by render.hip (per-tile reference path,
// linear/none blending, diagnostics) and render_batch.hip (the batched multiband path).  See render.hip's header
// comment for the reference lines restated here.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#include "aps_internal.h"

namespace aps {

// ------------------------------------------------------------------------------------------------
// device-side descriptors
// ------------------------------------------------------------------------------------------------
struct DevImage {
    const uint32_t* rgba;  // h*w words
    const float* wx;       // tent LUT, w entries
    const float* wy;       // tent LUT, h entries
    int h, w;
    float R[9];  // column-major, single(cam.R)
    float fx, fy, cx, cy;
    float gain[3];
    float g255[3];  // gain / 255: byte -> [0,1] and the gain in one factor (the fast sampler of the batched path)
    float tx[2], ty[2];  // the tent weight as min(p * t[0], (n - 1 - p) * t[1], 1), p = 0-based position (warpWeights in closed form)
    float cmin;  // cos of the largest angle between the optical axis and a ray that hits the pixel rectangle (minus a margin)
};

struct DevCanvas {
    int mode, H, W;
    float f, o0, o1;
    float Rref[9];
};

__device__ __forceinline__ void canvas_ray(const DevCanvas& cv, float xp, float yp, float d[3]) {
    float x, y, z;
    if (cv.mode == APS_PROJ_CYLINDRICAL) {
        const float th = cv.o0 + xp / cv.f, hl = cv.o1 + yp / cv.f;
        x = sinf(th);
        y = hl;
        z = cosf(th);
    } else if (cv.mode == APS_PROJ_SPHERICAL) {
        const float th = cv.o0 + xp / cv.f, ph = cv.o1 + yp / cv.f;
        const float cp = cosf(ph), sp = sinf(ph);
        x = cp * sinf(th);
        y = sp;
        z = cp * cosf(th);
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
        x = (cv.Rref[0] * rx + cv.Rref[1] * ry) + cv.Rref[2] * rz;
        y = (cv.Rref[3] * rx + cv.Rref[4] * ry) + cv.Rref[5] * rz;
        z = (cv.Rref[6] * rx + cv.Rref[7] * ry) + cv.Rref[8] * rz;
    }
    float n = sqrtf((x * x + y * y) + z * z);
    if (!(n > 1e-8f)) n = 1e-8f;
    d[0] = x / n;
    d[1] = y / n;
    d[2] = z / n;
}

struct Sample {
    float s[3];
    float wang, wf;
    bool m;
};

// Geometry only: (u, v, Wang, inside&front).  Used by the coverage prepass and by the sampler.
__device__ __forceinline__ bool project(const DevImage& im, const float d[3], float angle_pow,
                                        float& u, float& v, float& wa) {
    float cam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c] = fmaf(d[2], im.R[c + 6], fmaf(d[1], im.R[c + 3], d[0] * im.R[c]));
    const float epsz = 1e-6f;
    const bool front = cam[2] > epsz;
    const float cz = cam[2] > epsz ? cam[2] : epsz;
    u = im.fx * (cam[0] / cz) + im.cx;
    v = im.fy * (cam[1] / cz) + im.cy;
    wa = cam[2] > 0.0f ? cam[2] : 0.0f;
    if (angle_pow == 2.0f)
        wa = wa * wa;
    else if (angle_pow != 1.0f)
        wa = powf(wa, angle_pow);
    wa = front ? wa : 0.0f;
    if (!isfinite(u) || !isfinite(v)) {
        u = 1.0f;
        v = 1.0f;
    }
    const bool inside = (u >= 1.0f) && (u <= (float)im.w) && (v >= 1.0f) && (v <= (float)im.h);
    return inside && wa > 0.0f;
}

__device__ __forceinline__ Sample sample_one(const DevImage& im, const float d[3], float angle_pow) {
    Sample r;
    float u, v, wa;
    r.m = project(im, d, angle_pow, u, v, wa);
    if (r.m) {
        const int w = im.w, h = im.h;
        int x0 = (int)floorf(u), y0 = (int)floorf(v);
        x0 = max(1, min(x0, w - 1));
        y0 = max(1, min(y0, h - 1));
        const int x1 = min(x0 + 1, w), y1 = min(y0 + 1, h);
        const float s = u - (float)x0, t = v - (float)y0;
        const uint32_t p00 = im.rgba[(size_t)(y0 - 1) * w + (x0 - 1)];
        const uint32_t p10 = im.rgba[(size_t)(y0 - 1) * w + (x1 - 1)];
        const uint32_t p01 = im.rgba[(size_t)(y1 - 1) * w + (x0 - 1)];
        const uint32_t p11 = im.rgba[(size_t)(y1 - 1) * w + (x1 - 1)];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float g = im.gain[c];
            const float v00 = ((float)((p00 >> (8 * c)) & 255u) / 255.0f) * g;
            const float v10 = ((float)((p10 >> (8 * c)) & 255u) / 255.0f) * g;
            const float v01 = ((float)((p01 >> (8 * c)) & 255u) / 255.0f) * g;
            const float v11 = ((float)((p11 >> (8 * c)) & 255u) / 255.0f) * g;
            const float top = (1.0f - s) * v00 + s * v10;
            const float bot = (1.0f - s) * v01 + s * v11;
            r.s[c] = top * (1.0f - t) + bot * t;
        }
        const float wy0 = im.wy[y0 - 1], wy1 = im.wy[y1 - 1], wx0 = im.wx[x0 - 1], wx1 = im.wx[x1 - 1];
        const float f00 = wy0 * wx0, f10 = wy0 * wx1, f01 = wy1 * wx0, f11 = wy1 * wx1;
        const float top = (1.0f - s) * f00 + s * f10;
        const float bot = (1.0f - s) * f01 + s * f11;
        r.wf = top * (1.0f - t) + bot * t;
        r.wang = wa;
    } else {
        r.s[0] = r.s[1] = r.s[2] = 0.0f;
        r.wf = 0.0f;
        r.wang = 0.0f;
    }
    return r;
}

// Footprint of a layer inside its tile, half-open [x0,x1) x [y0,y1).  A layer is EXACTLY zero (colour and
// weight) outside its rect, at every pyramid level (the rect grows by the filter radius per blur and is
// mapped through the resize taps per level), so kernels neither store nor load there: a load outside the
// rect is replaced by 0, which is the value the full-tile computation would have read.  Results are the
// same bits as processing full tiles; the traffic is that of the footprints.
struct Rect {
    int x0, y0, x1, y1;
};
constexpr int kMaxK = 16;  // layers per kernel-argument table
struct RectTab {
    Rect r[kMaxK];
};
__device__ __forceinline__ bool in_rect(const Rect& r, int x, int y) {
    return x >= r.x0 && x < r.x1 && y >= r.y0 && y < r.y1;
}
__device__ __forceinline__ float4 ld_rect(const float4* __restrict__ p, int w, const Rect& r, int x, int y) {
    return in_rect(r, x, y) ? p[(size_t)y * w + x] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// pyramid building blocks on float4 images
// ------------------------------------------------------------------------------------------------
struct Taps {
    float k[17];
    int r;
};

__device__ __forceinline__ float4 fma4(float w, const float4 v, float4 a) {
    a.x = fmaf(w, v.x, a.x);
    a.y = fmaf(w, v.y, a.y);
    a.z = fmaf(w, v.z, a.z);
    a.w = fmaf(w, v.w, a.w);
    return a;
}

// imresize contributions (triangle kernel) for output index x (0-based); all in f64 like MATLAB
__device__ __forceinline__ int resize_taps(int in_len, int out_len, int x, int& left, float wts[12]) {
    const double scale = (double)out_len / (double)in_len;
    const double kw = scale < 1.0 ? 2.0 / scale : 2.0;
    const double u = (double)(x + 1) / scale + 0.5 * (1.0 - 1.0 / scale);
    left = (int)floor(u - kw / 2.0);
    int P = (int)ceil(kw) + 2;
    if (P > 12) P = 12;  // scale >= 0.2 always holds for floor(/2) pyramids (scale in [1/3, 1/2] or >= 2)
    double wd[12], s = 0;
    for (int t = 0; t < P; ++t) {
        const double dx = u - (double)(left + t);
        double a = scale < 1.0 ? scale * dx : dx;
        a = fabs(a);
        double v = a < 1.0 ? 1.0 - a : 0.0;
        if (scale < 1.0) v = scale * v;
        wd[t] = v;
        s += v;
    }
    for (int t = 0; t < P; ++t) wts[t] = (float)(wd[t] / s);
    return P;
}

// ---- host-side helpers shared by both render translation units ------------------------------------
inline Taps make_taps(float sigma) {
    Taps tp;
    const int r = (int)std::ceil(2.0 * (double)sigma);
    APS_REQUIRE(r >= 0 && r <= 8, APS_E_ARG, "MBBsigma %g needs a %d-tap filter (max 17 supported)",
                (double)sigma, 2 * r + 1);
    double t[17], s = 0;
    for (int i = 0; i <= 2 * r; ++i) {
        const double x = (double)(i - r);
        t[i] = std::exp(-(x * x) / (2.0 * (double)sigma * (double)sigma));
        s += t[i];
    }
    for (int i = 0; i <= 2 * r; ++i) tp.k[i] = (float)(t[i] / s);
    tp.r = r;
    return tp;
}

inline bool rows_first(int h, int w, int oh, int ow) { return (double)oh / h <= (double)ow / w; }

inline Rect clip_rect(Rect r, int w, int h) {
    r.x0 = std::max(r.x0, 0);
    r.y0 = std::max(r.y0, 0);
    r.x1 = std::min(r.x1, w);
    r.y1 = std::min(r.y1, h);
    if (r.x1 <= r.x0 || r.y1 <= r.y0) r = Rect{0, 0, 0, 0};
    return r;
}
// Output pixels of imresize (in_len -> out_len, antialiased triangle of half-width max(1, 1/scale)) that can see
// the input interval [a, b): o in [a*s + 0.5*s - 1.5, (b-1)*s + 0.5*s + 0.5] for s < 1; padded by one more pixel.
inline void map_interval(int a, int b, int in_len, int out_len, int& oa, int& ob) {
    if (b <= a) {
        oa = ob = 0;
        return;
    }
    const double s = (double)out_len / in_len;
    const double half = s < 1.0 ? 1.0 : s;  // support in output pixels
    oa = (int)std::floor(a * s - half - 2.0);
    ob = (int)std::ceil(b * s + half + 2.0);
    oa = std::max(oa, 0);
    ob = std::min(ob, out_len);
    if (ob <= oa) oa = ob = 0;
}
inline Rect map_rect(const Rect& r, int h, int w, int oh, int ow) {
    Rect o;
    map_interval(r.x0, r.x1, w, ow, o.x0, o.x1);
    map_interval(r.y0, r.y1, h, oh, o.y0, o.y1);
    if (o.x1 <= o.x0 || o.y1 <= o.y0) o = Rect{0, 0, 0, 0};
    return o;
}


// ---- the batched multiband path (render_batch.hip) ----------------------------------------------------
struct TileRect {
    int r0, c0, ht, wt;
};
// Renders the given tiles of the canvas with multiband blending, all tiles in one launch sequence.  dimgs: the
// prepared device image table, himgs its host copy (the analytic footprints read the cameras there).  pano/covered are DEVICE pointers.  Returns false when the configuration is outside
// what the batched kernels are built for (the caller then takes the per-tile path).
// need_images(used): called once, before the first kernel that samples pixels, with used[i] != 0 for every image that meets one
// of the tiles - the caller converts those images' pixels then (and no others).
using NeedImages = std::function<void(const std::vector<char>& used)>;
bool render_multiband_batched(const DevImage* dimgs, const DevImage* himgs, int n_img, const DevCanvas& cv, const aps_render_opts& o,
                              const std::vector<TileRect>& tiles, int out_layout, uint8_t* pano, uint8_t* covered,
                              const NeedImages& need_images);
bool render_fuse_batched(const DevImage* dimgs, const DevImage* himgs, int n_img, const DevCanvas& cv, const aps_render_opts& o,
                              const std::vector<TileRect>& tiles, int out_layout, uint8_t* pano, uint8_t* covered,
                              const NeedImages& need_images);

}  // namespace aps
