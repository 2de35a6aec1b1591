/*
 * render_oracle.c — CPU restatement of the reference's render path:
 *   PP/renderPanorama/renderPanorama.m:342-425 (tile loop, ray generation, paint, uint8),
 *   :825-1060 (fuseTile), :1063-1146 (sampleOneTile), :1282-1312 (warpWeights), :1393-1456 (sampleBlock),
 *   PP/blending/multiBandBlending.m:45-171, PP/blending/linearBlending.m:46-115,
 *   PP/imageProcessing/imageWarp.m:39-168 (bilinear).
 *
 * TEST INFRASTRUCTURE ONLY (see match_oracle.c).  PARITY UNPINNED: interp2, imgaussfilt and imresize
 * are closed MathWorks toolbox code and the reference has no tests.  Their semantics are fixed here from
 * the public documentation and restated explicitly:
 *   interp2(X,Y,V,u,v,'linear',NaN): x0 = clamp(floor(u),1,w-1), s = u-x0 (same in y),
 *       out = ((1-s)*v00 + s*v10)*(1-t) + ((1-s)*v01 + s*v11)*t in f32; NaN unless 1<=u<=w and 1<=v<=h.
 *   imgaussfilt(A,sigma,'Padding','replicate'): size 2*ceil(2*sigma)+1, taps exp(-x^2/(2 sigma^2))
 *       normalised in double then cast to f32, separable: column (vertical) pass, then row pass.
 *   imresize(A,[oh ow],'bilinear'): triangle kernel, antialiased when shrinking (kernel stretched by
 *       1/scale), u = x/scale + 0.5*(1-1/scale), left = floor(u - width/2), P = ceil(width)+2 taps,
 *       weights normalised in double then cast to f32, indices clamped to [1,n]; the dimension with the
 *       smaller scale factor is resized first (ties: rows first).
 *   All tap sums are k-ascending f32 fma chains acc = fmaf(w_t, x_t, acc) from 0 — the HIP path uses the
 *   same order, so the only device/host differences come from sinf/cosf/powf (ray generation), which is
 *   why rendered pixels are compared with a tolerance (stated in tests/test_render_gpu.py), not bitwise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

typedef struct {
    const uint8_t* data; /* row-major interleaved h x w x c */
    int height, width, channels;
    double K[9]; /* column-major */
    double R[9]; /* column-major */
    float gain[3];
} orc_image;

typedef struct {
    int mode; /* 0 cyl, 1 sph, 2 planar, 3 stereographic */
    int height, width;
    double f_pan, origin0, origin1;
    double R_ref[9];
} orc_canvas;

typedef struct {
    int tile_h, tile_w;
    float angle_power;
    int blending; /* 0 none, 1 linear, 2 multiband */
    int pyr_levels;
    float pyr_sigma;
    int none_policy; /* 0 last, 1 first, 2 maxangle */
    int canvas_white;
} orc_render_opts;

/* ---- warpWeights (:1282-1312): separable tent, linspace in double then stored as single ---------- */
static void tent(int n, float* w) {
    for (int i = 0; i < n; ++i) w[i] = 1.0f;
    const int a = (n + 1) / 2; /* ceil(n/2) */
    for (int k = 0; k < a; ++k) { /* linspace(0,1,a): d1 + k*(d2-d1)/(a-1), endpoints exact */
        double v = a > 1 ? 0.0 + ((double)k * 1.0) / (double)(a - 1) : 1.0; /* linspace(0,1,1) = 1 */
        if (k == a - 1) v = 1.0;
        w[k] = (float)v;
    }
    const int b0 = n / 2, nb = n - n / 2; /* wx(floor(n/2)+1 : n) = linspace(1,0,nb) */
    for (int k = 0; k < nb; ++k) {
        double v = nb > 1 ? 1.0 + ((double)k * -1.0) / (double)(nb - 1) : 0.0; /* linspace(1,0,1) = 0 */
        if (k == 0 && nb > 1) v = 1.0;
        if (k == nb - 1) v = 0.0;
        w[b0 + k] = (float)v;
    }
}

ORC_API void orc_tent(int n, float* w) { tent(n, w); }

/* ---- ray of canvas pixel (xp, yp) 0-based (:349-388) ---------------------------------------------- */
static void ray(const orc_canvas* cv, float xp, float yp, float* d) {
    const float f = (float)cv->f_pan, o0 = (float)cv->origin0, o1 = (float)cv->origin1;
    float x, y, z;
    if (cv->mode == 0) {
        const float th = o0 + xp / f, hl = o1 + yp / f;
        x = sinf(th); y = hl; z = cosf(th);
    } else if (cv->mode == 1) {
        const float th = o0 + xp / f, ph = o1 + yp / f;
        const float cp = cosf(ph), sp = sinf(ph);
        x = cp * sinf(th); y = sp; z = cp * cosf(th);
    } else {
        float rx, ry, rz;
        if (cv->mode == 2) {
            rx = o0 + xp / f; ry = o1 + yp / f; rz = 1.0f;
        } else {
            const float a = o0 + xp / f, b = o1 + yp / f;
            const float r2 = a * a + b * b, den = 1.0f + r2;
            rx = 2.0f * a / den; ry = 2.0f * b / den; rz = (1.0f - r2) / den;
        }
        const float R11 = (float)cv->R_ref[0], R21 = (float)cv->R_ref[1], R31 = (float)cv->R_ref[2];
        const float R12 = (float)cv->R_ref[3], R22 = (float)cv->R_ref[4], R32 = (float)cv->R_ref[5];
        const float R13 = (float)cv->R_ref[6], R23 = (float)cv->R_ref[7], R33 = (float)cv->R_ref[8];
        x = (R11 * rx + R21 * ry) + R31 * rz;
        y = (R12 * rx + R22 * ry) + R32 * rz;
        z = (R13 * rx + R23 * ry) + R33 * rz;
    }
    float n = sqrtf((x * x + y * y) + z * z);
    if (!(n > 1e-8f)) n = 1e-8f; /* max(nrm, 1e-8) */
    d[0] = x / n; d[1] = y / n; d[2] = z / n;
}

/* ---- sampleOneTile for one ray (:1104-1144).  Returns mask; S[3], Wang, Wf zeroed when !mask ------- */
static int sample_one(const orc_image* im, const float* wx, const float* wy, const float* d,
                      float angle_pow, float* S, float* Wang, float* Wf) {
    float R[9], fx = (float)im->K[0], fy = (float)im->K[4], cxp = (float)im->K[6], cyp = (float)im->K[7];
    for (int e = 0; e < 9; ++e) R[e] = (float)im->R[e];
    /* dirc = DWt * R.' : dirc(c) = sum_k d(k) * R(c,k), R(c,k) = R[c + 3k] */
    float cam[3];
    for (int c = 0; c < 3; ++c) cam[c] = fmaf(d[2], R[c + 6], fmaf(d[1], R[c + 3], d[0] * R[c]));
    const float epsz = 1e-6f;
    const int front = cam[2] > epsz;
    const float cz = cam[2] > epsz ? cam[2] : epsz;
    float u = fx * (cam[0] / cz) + cxp;
    float v = fy * (cam[1] / cz) + cyp;
    /* Wang = max(0, DWt*fw).^anglePow .* front, fw = R(3,:)' -> the same dot product as cam[2] */
    float wa = cam[2] > 0.0f ? cam[2] : 0.0f;
    if (angle_pow == 2.0f) wa = wa * wa;
    else if (angle_pow != 1.0f) wa = powf(wa, angle_pow);
    wa = front ? wa : 0.0f;
    if (!isfinite(u) || !isfinite(v)) { u = 1.0f; v = 1.0f; } /* sampleBlock :1433-1438 */
    const int w = im->width, h = im->height;
    const int inside = (u >= 1.0f) && (u <= (float)w) && (v >= 1.0f) && (v <= (float)h);
    int m = inside && wa > 0.0f;
    if (m) {
        int x0 = (int)floorf(u), y0 = (int)floorf(v);
        if (x0 > w - 1) x0 = w - 1;
        if (y0 > h - 1) y0 = h - 1;
        if (x0 < 1) x0 = 1;
        if (y0 < 1) y0 = 1;
        const int x1 = x0 + 1 <= w ? x0 + 1 : w, y1 = y0 + 1 <= h ? y0 + 1 : h; /* w==1 / h==1 */
        const float s = u - (float)x0, t = v - (float)y0;
        const int C = im->channels;
        for (int c = 0; c < 3; ++c) {
            const int cc = C == 1 ? 0 : c; /* gray is replicated to RGB (loadImages.m:62) */
#define PIX(xx, yy) (((float)im->data[((size_t)((yy)-1) * w + ((xx)-1)) * C + cc] / 255.0f) * im->gain[c])
            const float v00 = PIX(x0, y0), v10 = PIX(x1, y0), v01 = PIX(x0, y1), v11 = PIX(x1, y1);
#undef PIX
            const float top = (1.0f - s) * v00 + s * v10;
            const float bot = (1.0f - s) * v01 + s * v11;
            S[c] = top * (1.0f - t) + bot * t;
        }
        /* feather map srcW = wy*wx sampled the same way (:1131) */
        const float f00 = wy[y0 - 1] * wx[x0 - 1], f10 = wy[y0 - 1] * wx[x1 - 1];
        const float f01 = wy[y1 - 1] * wx[x0 - 1], f11 = wy[y1 - 1] * wx[x1 - 1];
        const float top = (1.0f - s) * f00 + s * f10;
        const float bot = (1.0f - s) * f01 + s * f11;
        *Wf = top * (1.0f - t) + bot * t;
        *Wang = wa;
    } else {
        S[0] = S[1] = S[2] = 0.0f;
        *Wf = 0.0f;
        *Wang = 0.0f;
    }
    return m;
}

/* one tile, one image: S ht x wt x 3, M, Wang, Wf (aps_warp_tile's contract) */
ORC_API void orc_warp_tile(const orc_image* im, const orc_canvas* cv, int r0, int c0, int ht, int wt,
                           float angle_pow, float* S, uint8_t* M, float* Wang, float* Wf) {
    float* wx = (float*)malloc(sizeof(float) * (size_t)im->width);
    float* wy = (float*)malloc(sizeof(float) * (size_t)im->height);
    tent(im->width, wx);
    tent(im->height, wy);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ht; ++y)
        for (int x = 0; x < wt; ++x) {
            float d[3];
            ray(cv, (float)(c0 + x), (float)(r0 + y), d);
            const size_t o = (size_t)y * wt + x;
            M[o] = (uint8_t)sample_one(im, wx, wy, d, angle_pow, S + 3 * o, Wang + o, Wf + o);
        }
    free(wx);
    free(wy);
}

/* ---- imgaussfilt / imresize building blocks (planar f32 image h x w, C interleaved channels) ------- */
static void gauss_taps(float sigma, int* r_out, float* k /* >= 2*ceil(2 sigma)+1 */) {
    const int r = (int)ceil(2.0 * (double)sigma);
    double s = 0, t[64];
    for (int i = 0; i <= 2 * r; ++i) {
        const double x = (double)(i - r);
        t[i] = exp(-(x * x) / (2.0 * (double)sigma * (double)sigma));
        s += t[i];
    }
    for (int i = 0; i <= 2 * r; ++i) k[i] = (float)(t[i] / s);
    *r_out = r;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

ORC_API void orc_gaussfilt(const float* in, int h, int w, int C, float sigma, float* out) {
    int r;
    float k[64];
    gauss_taps(sigma, &r, k);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)h * w * C);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int t = 0; t <= 2 * r; ++t)
                    acc = fmaf(k[t], in[((size_t)clampi(y + t - r, 0, h - 1) * w + x) * C + c], acc);
                tmp[((size_t)y * w + x) * C + c] = acc;
            }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int t = 0; t <= 2 * r; ++t)
                    acc = fmaf(k[t], tmp[((size_t)y * w + clampi(x + t - r, 0, w - 1)) * C + c], acc);
                out[((size_t)y * w + x) * C + c] = acc;
            }
    free(tmp);
}

/* contributions() of imresize for the triangle kernel; idx 0-based; returns P */
static int resize_taps(int in_len, int out_len, int x /*0-based*/, int* idx, float* wts) {
    const double scale = (double)out_len / (double)in_len;
    const double kw = scale < 1.0 ? 2.0 / scale : 2.0;
    const double u = (double)(x + 1) / scale + 0.5 * (1.0 - 1.0 / scale);
    const int left = (int)floor(u - kw / 2.0);
    const int P = (int)ceil(kw) + 2;
    double wd[64], s = 0;
    for (int t = 0; t < P; ++t) {
        const double dx = u - (double)(left + t);
        double a = scale < 1.0 ? scale * dx : dx;
        a = fabs(a);
        double v = a < 1.0 ? 1.0 - a : 0.0; /* triangle */
        if (scale < 1.0) v = scale * v;
        wd[t] = v;
        s += v;
    }
    for (int t = 0; t < P; ++t) {
        wts[t] = (float)(wd[t] / s);
        idx[t] = clampi(left + t, 1, in_len) - 1;
    }
    return P;
}

static void resize_dim(const float* in, int h, int w, int C, int dim, int out_len, float* out) {
    const int oh = dim == 0 ? out_len : h, ow = dim == 1 ? out_len : w;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; ++y) {
        int idx[64];
        float wts[64];
        int P = 0;
        if (dim == 0) P = resize_taps(h, out_len, y, idx, wts);
        for (int x = 0; x < ow; ++x) {
            if (dim == 1) P = resize_taps(w, out_len, x, idx, wts);
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int t = 0; t < P; ++t) {
                    const size_t src = dim == 0 ? ((size_t)idx[t] * w + x) : ((size_t)y * w + idx[t]);
                    acc = fmaf(wts[t], in[src * C + c], acc);
                }
                out[((size_t)y * ow + x) * C + c] = acc;
            }
        }
    }
}

ORC_API void orc_imresize(const float* in, int h, int w, int C, int oh, int ow, float* out) {
    const double sr = (double)oh / h, sc = (double)ow / w;
    if (sr <= sc) { /* rows first */
        float* tmp = (float*)malloc(sizeof(float) * (size_t)oh * w * C);
        resize_dim(in, h, w, C, 0, oh, tmp);
        resize_dim(tmp, oh, w, C, 1, ow, out);
        free(tmp);
    } else {
        float* tmp = (float*)malloc(sizeof(float) * (size_t)h * ow * C);
        resize_dim(in, h, w, C, 1, ow, tmp);
        resize_dim(tmp, h, ow, C, 0, oh, out);
        free(tmp);
    }
}

/* ---- multiBandBlending (multiBandBlending.m:45-171) ----------------------------------------------
 * C: K x h x w x 3, Wt: K x h x w, F: h x w x 3 (all f32, row-major). */
ORC_API void orc_multiband_blend(const float* Cc, const float* Wt, int K, int h, int w, int levels,
                                 float sigma, float* F) {
    const size_t hw = (size_t)h * w;
    /* :72-85 weight normalisation */
    float* Wn = (float*)malloc(sizeof(float) * hw * K);
    for (size_t p = 0; p < hw; ++p) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) {
            const float v = Wt[k * hw + p];
            s = s + (v > 0.f ? v : 0.f);
        }
        for (int k = 0; k < K; ++k) {
            const float v = Wt[k * hw + p] > 0.f ? Wt[k * hw + p] : 0.f;
            Wn[k * hw + p] = s > 1e-8f ? v / s : 0.f;
        }
    }
    /* :98-109 level sizes */
    int maxl = (int)floor(log2((double)(h < w ? h : w)));
    if (levels > maxl) levels = maxl;
    if (levels < 1) levels = 1;
    int* lh = (int*)malloc(sizeof(int) * levels);
    int* lw = (int*)malloc(sizeof(int) * levels);
    lh[0] = h; lw[0] = w;
    for (int l = 1; l < levels; ++l) {
        lh[l] = lh[l - 1] / 2 > 1 ? lh[l - 1] / 2 : 1;
        lw[l] = lw[l - 1] / 2 > 1 ? lw[l - 1] / 2 : 1;
    }
    float** Num = (float**)malloc(sizeof(float*) * levels);
    for (int l = 0; l < levels; ++l) Num[l] = (float*)calloc((size_t)lh[l] * lw[l] * 3, sizeof(float));
    /* :119-160 */
    for (int k = 0; k < K; ++k) {
        float* Gc = (float*)malloc(sizeof(float) * hw * 3);
        float* Gw = (float*)malloc(sizeof(float) * hw);
        memcpy(Gc, Cc + (size_t)k * hw * 3, sizeof(float) * hw * 3);
        memcpy(Gw, Wn + (size_t)k * hw, sizeof(float) * hw);
        for (int l = 0; l < levels - 1; ++l) {
            const int hl = lh[l], wl = lw[l], nh = lh[l + 1], nw = lw[l + 1];
            const size_t n = (size_t)hl * wl, nn = (size_t)nh * nw;
            float* blurc = (float*)malloc(sizeof(float) * n * 3);
            float* bw = (float*)malloc(sizeof(float) * n);
            float* Dc = (float*)malloc(sizeof(float) * nn * 3);
            float* Dw = (float*)malloc(sizeof(float) * nn);
            float* Uc = (float*)malloc(sizeof(float) * n * 3);
            orc_gaussfilt(Gc, hl, wl, 3, sigma, blurc);
            orc_imresize(blurc, hl, wl, 3, nh, nw, Dc);
            orc_gaussfilt(Gw, hl, wl, 1, sigma, bw);
            orc_imresize(bw, hl, wl, 1, nh, nw, Dw);
            orc_imresize(Dc, nh, nw, 3, hl, wl, Uc);
            for (size_t p = 0; p < n; ++p)
                for (int c = 0; c < 3; ++c) {
                    const float L = Gc[3 * p + c] - Uc[3 * p + c];
                    Num[l][3 * p + c] = Num[l][3 * p + c] + L * Gw[p];
                }
            free(blurc); free(bw); free(Uc); free(Gc); free(Gw);
            Gc = Dc; Gw = Dw;
        }
        const size_t n = (size_t)lh[levels - 1] * lw[levels - 1];
        for (size_t p = 0; p < n; ++p)
            for (int c = 0; c < 3; ++c)
                Num[levels - 1][3 * p + c] = Num[levels - 1][3 * p + c] + Gc[3 * p + c] * Gw[p];
        free(Gc); free(Gw);
    }
    /* :163-171 collapse + clamp */
    float* cur = (float*)malloc(sizeof(float) * (size_t)lh[levels - 1] * lw[levels - 1] * 3);
    memcpy(cur, Num[levels - 1], sizeof(float) * (size_t)lh[levels - 1] * lw[levels - 1] * 3);
    for (int l = levels - 2; l >= 0; --l) {
        float* up = (float*)malloc(sizeof(float) * (size_t)lh[l] * lw[l] * 3);
        orc_imresize(cur, lh[l + 1], lw[l + 1], 3, lh[l], lw[l], up);
        const size_t n = (size_t)lh[l] * lw[l] * 3;
        for (size_t e = 0; e < n; ++e) up[e] = up[e] + Num[l][e];
        free(cur);
        cur = up;
    }
    for (size_t e = 0; e < hw * 3; ++e) {
        float v = cur[e];
        v = v > 0.f ? v : 0.f; /* max(0,F) maps NaN to 0 like MATLAB's max */
        F[e] = v < 1.f ? v : 1.f;
    }
    free(cur);
    for (int l = 0; l < levels; ++l) free(Num[l]);
    free(Num); free(lh); free(lw); free(Wn);
}

/* ---- linearBlending (linearBlending.m:64-101) ----------------------------------------------------- */
ORC_API void orc_linear_blend(const float* Cc, const float* Wt, int K, int h, int w, float* F) {
    const size_t hw = (size_t)h * w;
    const float tiny = 1.1920928955078125e-07f;
    for (size_t p = 0; p < hw; ++p) {
        float num[3] = {0, 0, 0}, den = 0;
        for (int k = 0; k < K; ++k) {
            const float wv = Wt[k * hw + p];
            for (int c = 0; c < 3; ++c) num[c] = num[c] + Cc[((size_t)k * hw + p) * 3 + c] * wv;
            den = den + wv;
        }
        const float d = den > tiny ? den : tiny;
        for (int c = 0; c < 3; ++c) F[3 * p + c] = num[c] / d;
    }
}

/* ---- fuseTile + tile loop (:342-425, :825-1060) --------------------------------------------------- */
static uint8_t to_u8(float p) { /* uint8(max(0,min(255,round(255*p)))) — MATLAB round: half away */
    float v = 255.0f * p;
    v = roundf(v);
    if (!(v > 0.f)) v = 0.f;
    if (v > 255.f) v = 255.f;
    return (uint8_t)v;
}

ORC_API void orc_render(const orc_image* imgs, int n, const orc_canvas* cv, const orc_render_opts* op,
                        uint8_t* pano, uint8_t* covered) {
    const int H = cv->height, W = cv->width;
    float** wx = (float**)malloc(sizeof(float*) * n);
    float** wy = (float**)malloc(sizeof(float*) * n);
    for (int i = 0; i < n; ++i) {
        wx[i] = (float*)malloc(sizeof(float) * imgs[i].width);
        wy[i] = (float*)malloc(sizeof(float) * imgs[i].height);
        tent(imgs[i].width, wx[i]);
        tent(imgs[i].height, wy[i]);
    }
    for (int r0 = 0; r0 < H; r0 += op->tile_h)
        for (int c0 = 0; c0 < W; c0 += op->tile_w) {
            const int ht = r0 + op->tile_h <= H ? op->tile_h : H - r0;
            const int wt = c0 + op->tile_w <= W ? op->tile_w : W - c0;
            const size_t T = (size_t)ht * wt;
            float* D = (float*)malloc(sizeof(float) * T * 3);
#pragma omp parallel for schedule(static)
            for (int y = 0; y < ht; ++y)
                for (int x = 0; x < wt; ++x) ray(cv, (float)(c0 + x), (float)(r0 + y), D + 3 * ((size_t)y * wt + x));
            float* Ftile = (float*)calloc(T * 3, sizeof(float));
            uint8_t* cov = (uint8_t*)calloc(T, 1);
            float* S = (float*)malloc(sizeof(float) * T * 3);
            float* Wa = (float*)malloc(sizeof(float) * T);
            float* Wf = (float*)malloc(sizeof(float) * T);
            uint8_t* M = (uint8_t*)malloc(T);
            if (op->blending == 0) { /* 'none' (:864-914) */
                float* bestW = (float*)calloc(T, sizeof(float));
                for (int i = 0; i < n; ++i) {
#pragma omp parallel for schedule(static)
                    for (size_t p = 0; p < T; ++p)
                        M[p] = (uint8_t)sample_one(&imgs[i], wx[i], wy[i], D + 3 * p, op->angle_power,
                                                   S + 3 * p, Wa + p, Wf + p);
                    for (size_t p = 0; p < T; ++p) {
                        int upd = 0;
                        if (op->none_policy == 0) upd = M[p];
                        else if (op->none_policy == 1) upd = M[p] && !cov[p];
                        else upd = M[p] && (Wa[p] > bestW[p]);
                        if (upd) {
                            for (int c = 0; c < 3; ++c) Ftile[3 * p + c] = S[3 * p + c];
                            if (op->none_policy == 2) bestW[p] = Wa[p];
                            cov[p] = 1;
                        }
                    }
                }
                free(bestW);
            } else if (op->blending == 1) { /* 'linear' (:916-978) */
                float* acc = (float*)calloc(T * 3, sizeof(float));
                float* wsum = (float*)calloc(T, sizeof(float));
                float* bestW = (float*)calloc(T, sizeof(float));
                float* bestRGB = (float*)calloc(T * 3, sizeof(float));
                uint8_t* anyv = (uint8_t*)calloc(T, 1);
                for (int i = 0; i < n; ++i) {
                    int any = 0;
#pragma omp parallel for schedule(static) reduction(| : any)
                    for (size_t p = 0; p < T; ++p) {
                        M[p] = (uint8_t)sample_one(&imgs[i], wx[i], wy[i], D + 3 * p, op->angle_power,
                                                   S + 3 * p, Wa + p, Wf + p);
                        any |= M[p];
                    }
                    if (!any) continue;
                    for (size_t p = 0; p < T; ++p) {
                        float f = Wf[p];
                        if (!isfinite(f)) f = 0.f;
                        f = f > 1e-4f ? f : 1e-4f; /* max(Wf, 1e-4) also for pixels outside M */
                        float wv = Wa[p] * f;
                        if (!M[p]) wv = 0.f;
                        for (int c = 0; c < 3; ++c) acc[3 * p + c] = acc[3 * p + c] + S[3 * p + c] * wv;
                        wsum[p] = wsum[p] + wv;
                        anyv[p] |= M[p];
                        if (M[p] && wv > bestW[p]) {
                            bestW[p] = wv;
                            for (int c = 0; c < 3; ++c) bestRGB[3 * p + c] = S[3 * p + c];
                        }
                    }
                }
                for (size_t p = 0; p < T; ++p) {
                    const int z = wsum[p] > 1e-12f;
                    for (int c = 0; c < 3; ++c)
                        Ftile[3 * p + c] = z ? acc[3 * p + c] / wsum[p] : (anyv[p] ? bestRGB[3 * p + c] : 0.f);
                    cov[p] = wsum[p] > 0.f;
                }
                free(acc); free(wsum); free(bestW); free(bestRGB); free(anyv);
            } else { /* 'multiband' (:980-1044) */
                float* Ci = NULL;
                float* Wc = NULL;
                int K = 0;
                for (int i = 0; i < n; ++i) {
                    int any = 0;
#pragma omp parallel for schedule(static) reduction(| : any)
                    for (size_t p = 0; p < T; ++p) {
                        M[p] = (uint8_t)sample_one(&imgs[i], wx[i], wy[i], D + 3 * p, op->angle_power,
                                                   S + 3 * p, Wa + p, Wf + p);
                        any |= M[p];
                    }
                    if (!any) continue;
                    Ci = (float*)realloc(Ci, sizeof(float) * T * 3 * (K + 1));
                    Wc = (float*)realloc(Wc, sizeof(float) * T * (K + 1));
                    memcpy(Ci + (size_t)K * T * 3, S, sizeof(float) * T * 3);
                    for (size_t p = 0; p < T; ++p) {
                        const float wv = M[p] ? Wa[p] * Wf[p] : 0.f;
                        Wc[(size_t)K * T + p] = wv;
                        if (wv > 0.f) cov[p] = 1; /* anyValid */
                    }
                    ++K;
                }
                if (K > 0) {
                    for (size_t p = 0; p < T; ++p) { /* :1009-1017 */
                        float s = 0.f;
                        for (int k = 0; k < K; ++k) s = s + Wc[(size_t)k * T + p];
                        const float inv = s > 1e-8f ? 1.0f / s : 0.f;
                        for (int k = 0; k < K; ++k) Wc[(size_t)k * T + p] = Wc[(size_t)k * T + p] * inv;
                    }
                    orc_multiband_blend(Ci, Wc, K, ht, wt, op->pyr_levels, op->pyr_sigma, Ftile);
                }
                free(Ci);
                free(Wc);
            }
            /* write back (:397-400), paint void + uint8 (:408-425) */
            for (int y = 0; y < ht; ++y)
                for (int x = 0; x < wt; ++x) {
                    const size_t p = (size_t)y * wt + x, o = (size_t)(r0 + y) * W + (c0 + x);
                    for (int c = 0; c < 3; ++c)
                        pano[3 * o + c] = cov[p] ? to_u8(Ftile[3 * p + c]) : (op->canvas_white ? 255 : 0);
                    if (covered) covered[o] = cov[p];
                }
            free(D); free(Ftile); free(cov); free(S); free(Wa); free(Wf); free(M);
        }
    for (int i = 0; i < n; ++i) { free(wx[i]); free(wy[i]); }
    free(wx); free(wy);
}

/* ---- imageWarp 'bilinear' (imageWarp.m:39-168), f32 image, row-major interleaved ------------------ */
ORC_API void orc_image_warp_h(const float* in, int in_h, int in_w, int C, const double* Hcm, int out_h,
                              int out_w, double x0, double y0, double sx, double sy, float fill,
                              int round_to_u8, float* out) {
    double H[9];
    for (int e = 0; e < 9; ++e) H[e] = Hcm[e];
    if (H[8] != 0) for (int e = 0; e < 9; ++e) H[e] = Hcm[e] / Hcm[8]; /* :61-64 */
    /* src = H \ [X;Y;1] evaluated with the adjugate (homogeneous scale is divided out below, :99-101) */
#define HH(r, c) H[(r) + 3 * (c)]
    double A[9];
    A[0] = HH(1, 1) * HH(2, 2) - HH(1, 2) * HH(2, 1); A[3] = HH(0, 2) * HH(2, 1) - HH(0, 1) * HH(2, 2); A[6] = HH(0, 1) * HH(1, 2) - HH(0, 2) * HH(1, 1);
    A[1] = HH(1, 2) * HH(2, 0) - HH(1, 0) * HH(2, 2); A[4] = HH(0, 0) * HH(2, 2) - HH(0, 2) * HH(2, 0); A[7] = HH(0, 2) * HH(1, 0) - HH(0, 0) * HH(1, 2);
    A[2] = HH(1, 0) * HH(2, 1) - HH(1, 1) * HH(2, 0); A[5] = HH(0, 1) * HH(2, 0) - HH(0, 0) * HH(2, 1); A[8] = HH(0, 0) * HH(1, 1) - HH(0, 1) * HH(1, 0);
    const double det = (HH(0, 0) * A[0] + HH(0, 1) * A[1]) + HH(0, 2) * A[2];
#undef HH
#pragma omp parallel for schedule(static)
    for (int y = 0; y < out_h; ++y)
        for (int x = 0; x < out_w; ++x) {
            const double X = x0 + (double)x * sx, Y = y0 + (double)y * sy;
            double s0 = ((A[0] * X + A[3] * Y) + A[6]) / det;
            double s1 = ((A[1] * X + A[4] * Y) + A[7]) / det;
            double s2 = ((A[2] * X + A[5] * Y) + A[8]) / det;
            double wv = fabs(s2) > 1e-12 ? fabs(s2) : 1e-12; /* w = sign(w).*max(|w|,1e-12) (:100) */
            wv = s2 < 0 ? -wv : (s2 > 0 ? wv : 0.0);
            const double srcx = s0 / wv, srcy = s1 / wv;
            const double fx1 = floor(srcx), fy1 = floor(srcy);
            const int valid = fx1 >= 1 && fx1 + 1 <= in_w && fy1 >= 1 && fy1 + 1 <= in_h; /* :133 */
            for (int c = 0; c < C; ++c) {
                float o = fill;
                if (valid) {
                    const int x1 = (int)fx1, y1 = (int)fy1;
                    const double wx = srcx - fx1, wy = srcy - fy1;
                    const double w11 = (1 - wx) * (1 - wy), w12 = (1 - wx) * wy, w21 = wx * (1 - wy), w22 = wx * wy;
#define PX(xx, yy) ((double)in[((size_t)((yy)-1) * in_w + ((xx)-1)) * C + c])
                    const double v = ((w11 * PX(x1, y1) + w12 * PX(x1, y1 + 1)) + w21 * PX(x1 + 1, y1)) + w22 * PX(x1 + 1, y1 + 1);
#undef PX
                    if (round_to_u8) { /* cast(interpVals,'like',uint8): round half away, saturate */
                        double rr = round(v);
                        rr = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
                        o = (float)rr;
                    } else {
                        o = (float)v;
                    }
                }
                out[((size_t)y * out_w + x) * C + c] = o;
            }
        }
}

/* ---- imageWarp 'nearest' (:109-123) and 'bicubic' (:170-264); method 0 / 2 (1 = bilinear, above) ---- */
static double orc_warp_cubic(double x) { /* bicubicKernel, imageWarp.m:275-301; |x|^2, |x|^3 as products */
    const double a = fabs(x), a2 = a * a, a3 = a2 * a;
    if (a <= 1.0) return (1.5 * a3 - 2.5 * a2) + 1.0;
    if (a <= 2.0) return ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0;
    return 0.0;
}

ORC_API void orc_image_warp_h_m(const float* in, int in_h, int in_w, int C, const double* Hcm, int out_h,
                                int out_w, double x0, double y0, double sx, double sy, float fill,
                                int round_to_u8, int method, float* out) {
    if (method == 1) {
        orc_image_warp_h(in, in_h, in_w, C, Hcm, out_h, out_w, x0, y0, sx, sy, fill, round_to_u8, out);
        return;
    }
    double H[9];
    for (int e = 0; e < 9; ++e) H[e] = Hcm[e];
    if (H[8] != 0) for (int e = 0; e < 9; ++e) H[e] = Hcm[e] / Hcm[8];
#define HH(r, c) H[(r) + 3 * (c)]
    double A[9];
    A[0] = HH(1, 1) * HH(2, 2) - HH(1, 2) * HH(2, 1); A[3] = HH(0, 2) * HH(2, 1) - HH(0, 1) * HH(2, 2); A[6] = HH(0, 1) * HH(1, 2) - HH(0, 2) * HH(1, 1);
    A[1] = HH(1, 2) * HH(2, 0) - HH(1, 0) * HH(2, 2); A[4] = HH(0, 0) * HH(2, 2) - HH(0, 2) * HH(2, 0); A[7] = HH(0, 2) * HH(1, 0) - HH(0, 0) * HH(1, 2);
    A[2] = HH(1, 0) * HH(2, 1) - HH(1, 1) * HH(2, 0); A[5] = HH(0, 1) * HH(2, 0) - HH(0, 0) * HH(2, 1); A[8] = HH(0, 0) * HH(1, 1) - HH(0, 1) * HH(1, 0);
    const double det = (HH(0, 0) * A[0] + HH(0, 1) * A[1]) + HH(0, 2) * A[2];
#undef HH
#pragma omp parallel for schedule(static)
    for (int y = 0; y < out_h; ++y)
        for (int x = 0; x < out_w; ++x) {
            const double X = x0 + (double)x * sx, Y = y0 + (double)y * sy;
            double s0 = ((A[0] * X + A[3] * Y) + A[6]) / det;
            double s1 = ((A[1] * X + A[4] * Y) + A[7]) / det;
            double s2 = ((A[2] * X + A[5] * Y) + A[8]) / det;
            double wv = fabs(s2) > 1e-12 ? fabs(s2) : 1e-12;
            wv = s2 < 0 ? -wv : (s2 > 0 ? wv : 0.0);
            const double srcx = s0 / wv, srcy = s1 / wv;
            if (method == 0) { /* x = round(srcX) (half away from zero), valid inside the image (:111-113) */
                const double rx = round(srcx), ry = round(srcy);
                const int valid = rx >= 1 && rx <= in_w && ry >= 1 && ry <= in_h;
                for (int c = 0; c < C; ++c)
                    out[((size_t)y * out_w + x) * C + c] = valid ? in[((size_t)((int)ry - 1) * in_w + ((int)rx - 1)) * C + c] : fill;
                continue;
            }
            const double fx = floor(srcx), fy = floor(srcy);
            const int valid = fx >= 2 && fx <= in_w - 2 && fy >= 2 && fy <= in_h - 2; /* :177 */
            double wxk[4] = {0, 0, 0, 0}, wyk[4] = {0, 0, 0, 0};
            if (valid)
                for (int ii = -1; ii <= 2; ++ii) { /* :191-194 */
                    wxk[ii + 1] = orc_warp_cubic((double)ii - (srcx - fx));
                    wyk[ii + 1] = orc_warp_cubic((double)ii - (srcy - fy));
                }
            for (int c = 0; c < C; ++c) {
                float o = fill;
                if (valid) {
                    const int xb = (int)fx, yb = (int)fy;
                    double v = 0.0;
                    for (int jj = 0; jj < 4; ++jj) { /* xInterp(:,jj) over ii ascending from 0 (:236-243), then sum over jj (:246) */
                        double xi = 0.0;
                        for (int ii = 0; ii < 4; ++ii)
                            xi = xi + (double)in[((size_t)(yb + jj - 2) * in_w + (xb + ii - 2)) * C + c] * wxk[ii];
                        v = v + xi * wyk[jj];
                    }
                    if (round_to_u8) {
                        double rr = round(v);
                        rr = rr < 0 ? 0 : (rr > 255 ? 255 : rr);
                        o = (float)rr;
                    } else {
                        v = v < 0 ? 0 : (v > 1.0 ? 1.0 : v); /* maxVal = 1 for float images (:216,:254) */
                        o = (float)v;
                    }
                }
                out[((size_t)y * out_w + x) * C + c] = o;
            }
        }
}

/* ================================================================================================
 * Gain-compensation overlap statistics (PP/gainCompensation/gainCompensationRKf.m:96-149, 239-367,
 * 369-579): for every stride-th canvas point (1-BASED coordinates, :106-107), every image pair (i < j)
 * that both cover it contributes one count and the two bilinear RAW (0..255) colour samples.
 * PARITY UNPINNED (interp2 is toolbox code).  Fixed here and mirrored by the HIP path: rays as in
 * panoDirsGridTile (no normalisation), projection as in projectToImage (cz not clamped), coverage =
 * finite(u,v) & front & inside [1,w]x[1,h] & bilinear(wy*wx) > 0, colours by the same bilinear form as
 * sampleOneTile.  Sums are accumulated in double; the reference sums each tile in single first, and the
 * device adds in an unspecified order, so consumers compare sums with a relative tolerance (counts are exact).
 * Outputs are N x N (x 3) column-major, upper triangle.
 * ================================================================================================ */
static void gain_ray(const orc_canvas* cv, float xp, float yp, float* d) {
    const float f = (float)cv->f_pan, o0 = (float)cv->origin0, o1 = (float)cv->origin1;
    if (cv->mode == 0) {
        const float th = o0 + xp / f, hl = o1 + yp / f;
        d[0] = sinf(th); d[1] = hl; d[2] = cosf(th);
    } else if (cv->mode == 1) {
        const float th = o0 + xp / f, ph = o1 + yp / f;
        const float cp = cosf(ph), sp = sinf(ph);
        d[0] = cp * sinf(th); d[1] = sp; d[2] = cp * cosf(th);
    } else {
        float rx, ry, rz;
        if (cv->mode == 2) {
            rx = o0 + xp / f; ry = o1 + yp / f; rz = 1.0f;
        } else {
            const float a = o0 + xp / f, b = o1 + yp / f;
            const float r2 = a * a + b * b, den = 1.0f + r2;
            rx = 2.0f * a / den; ry = 2.0f * b / den; rz = (1.0f - r2) / den;
        }
        const double* R = cv->R_ref; /* Rt(r,c) = R(c,r): dw(r) = sum_c R(c,r) * dref(c) */
        d[0] = ((float)R[0] * rx + (float)R[1] * ry) + (float)R[2] * rz;
        d[1] = ((float)R[3] * rx + (float)R[4] * ry) + (float)R[5] * rz;
        d[2] = ((float)R[6] * rx + (float)R[7] * ry) + (float)R[8] * rz;
    }
}

static int gain_sample(const orc_image* im, const float* wx, const float* wy, const float* d, float* Cc) {
    float R[9], fx = (float)im->K[0], fy = (float)im->K[4], cxp = (float)im->K[6], cyp = (float)im->K[7];
    for (int e = 0; e < 9; ++e) R[e] = (float)im->R[e];
    float cam[3];
    for (int c = 0; c < 3; ++c) cam[c] = fmaf(d[2], R[c + 6], fmaf(d[1], R[c + 3], d[0] * R[c]));
    const int front = cam[2] > 1e-6f;
    const float u = fx * (cam[0] / cam[2]) + cxp;
    const float v = fy * (cam[1] / cam[2]) + cyp;
    const int w = im->width, h = im->height;
    if (!front || !isfinite(u) || !isfinite(v)) return 0;
    if (!((u >= 1.0f) && (u <= (float)w) && (v >= 1.0f) && (v <= (float)h))) return 0; /* interp2 -> NaN */
    int x0 = (int)floorf(u), y0 = (int)floorf(v);
    if (x0 > w - 1) x0 = w - 1;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 < 1) x0 = 1;
    if (y0 < 1) y0 = 1;
    const int x1 = x0 + 1 <= w ? x0 + 1 : w, y1 = y0 + 1 <= h ? y0 + 1 : h;
    const float s = u - (float)x0, t = v - (float)y0;
    const float f00 = wy[y0 - 1] * wx[x0 - 1], f10 = wy[y0 - 1] * wx[x1 - 1];
    const float f01 = wy[y1 - 1] * wx[x0 - 1], f11 = wy[y1 - 1] * wx[x1 - 1];
    const float wtop = (1.0f - s) * f00 + s * f10, wbot = (1.0f - s) * f01 + s * f11;
    const float wf = wtop * (1.0f - t) + wbot * t;
    if (!(wf > 0.0f)) return 0;
    const int C = im->channels;
    for (int c = 0; c < 3; ++c) {
        const int cc = C == 1 ? 0 : c;
#define RAW(xx, yy) ((float)im->data[((size_t)((yy)-1) * w + ((xx)-1)) * C + cc])
        const float v00 = RAW(x0, y0), v10 = RAW(x1, y0), v01 = RAW(x0, y1), v11 = RAW(x1, y1);
#undef RAW
        const float top = (1.0f - s) * v00 + s * v10;
        const float bot = (1.0f - s) * v01 + s * v11;
        Cc[c] = top * (1.0f - t) + bot * t;
    }
    return 1;
}

ORC_API void orc_gain_overlap_stats(const orc_image* imgs, int n, const orc_canvas* cv, int stride, double* Nij,
                                    double* sumCi, double* sumCj) {
    const size_t nn = (size_t)n * n;
    memset(Nij, 0, nn * sizeof(double));
    memset(sumCi, 0, 3 * nn * sizeof(double));
    memset(sumCj, 0, 3 * nn * sizeof(double));
    float** wx = (float**)malloc(sizeof(float*) * n);
    float** wy = (float**)malloc(sizeof(float*) * n);
    for (int i = 0; i < n; ++i) {
        wx[i] = (float*)malloc(sizeof(float) * imgs[i].width);
        wy[i] = (float*)malloc(sizeof(float) * imgs[i].height);
        tent(imgs[i].width, wx[i]);
        tent(imgs[i].height, wy[i]);
    }
    int* cov = (int*)malloc(sizeof(int) * n);
    float* col = (float*)malloc(sizeof(float) * 3 * n);
    if (stride < 1) stride = 1;
    for (int yp = 1; yp <= cv->height; yp += stride)
        for (int xp = 1; xp <= cv->width; xp += stride) {
            float d[3];
            gain_ray(cv, (float)xp, (float)yp, d);
            int k = 0;
            for (int i = 0; i < n; ++i)
                if (gain_sample(&imgs[i], wx[i], wy[i], d, col + 3 * k)) cov[k++] = i;
            for (int a = 0; a < k; ++a)
                for (int b = a + 1; b < k; ++b) {
                    const size_t e = (size_t)cov[a] + (size_t)n * cov[b];
                    Nij[e] += 1.0;
                    for (int c = 0; c < 3; ++c) {
                        sumCi[e + nn * c] += (double)col[3 * a + c];
                        sumCj[e + nn * c] += (double)col[3 * b + c];
                    }
                }
        }
    for (int i = 0; i < n; ++i) {
        free(wx[i]);
        free(wy[i]);
    }
    free(wx); free(wy); free(cov); free(col);
}

/* ================================================================================================
 * gainCompensationH's overlap statistics (PP/gainCompensation/gainCompensationH.m:45-52, 78-149): the images are
 * already warped to one canvas (Iw{k}: H x W x C float, Ww{k}: H x W float, row-major interleaved here).  Every ds-th
 * row and column (MATLAB 1:ds:end = 0-based 0, ds, 2 ds, ...) is sampled; valid(k) = Ww{k} > 0 & all channels finite
 * (:117); every pair i < j valid at a sample adds one count and its two colours, summed in double (:126-146).
 * Outputs N x N (x 3), column-major, upper triangle - the layout of orc_gain_overlap_stats.
 * ================================================================================================ */
ORC_API void orc_gain_overlap_stats_warped(const float* const* iw, const float* const* ww, int n, int h, int w, int C, int ds,
                                           double* Nij, double* sumCi, double* sumCj) {
    const size_t nn = (size_t)n * n;
    memset(Nij, 0, nn * sizeof(double));
    memset(sumCi, 0, 3 * nn * sizeof(double));
    memset(sumCj, 0, 3 * nn * sizeof(double));
    if (ds < 1) ds = 1;
    int* cov = (int*)malloc(sizeof(int) * n);
    for (int y = 0; y < h; y += ds)
        for (int x = 0; x < w; x += ds) {
            int k = 0;
            for (int i = 0; i < n; ++i) {
                if (!(ww[i][(size_t)y * w + x] > 0.0f)) continue;
                int fin = 1;
                for (int c = 0; c < C; ++c) fin = fin && isfinite(iw[i][((size_t)y * w + x) * C + c]);
                if (fin) cov[k++] = i;
            }
            for (int a = 0; a < k; ++a)
                for (int b = a + 1; b < k; ++b) {
                    const size_t e = (size_t)cov[a] + (size_t)n * cov[b];
                    Nij[e] += 1.0;
                    for (int c = 0; c < 3; ++c) {
                        const int cc = c < C ? c : C - 1;
                        sumCi[e + nn * c] += (double)iw[cov[a]][((size_t)y * w + x) * C + cc];
                        sumCj[e + nn * c] += (double)iw[cov[b]][((size_t)y * w + x) * C + cc];
                    }
                }
        }
    free(cov);
}

/* ================================================================================================
 * imresize(I, s | [oh ow], 'bicubic' | 'bilinear') on uint8 images -- the preprocessing step in front of SIFT
 * (PP/imageProcessing/resizeImagesToLimits.m:49-106).  imresize is toolbox code: PARITY
 * UNPINNED.  Fixed here and mirrored by the HIP path (public documentation of imresize/contributions):
 * antialiasing on shrink (kernel stretched by 1/scale), half-pixel centres u = x/scale + 0.5(1 - 1/scale), P =
 * ceil(width)+2 taps from floor(u - width/2), weights normalised to sum 1, indices clamped (replicate), the dimension
 * with the smaller scale first (ties: rows), the intermediate image rounded and saturated back to uint8 (the mex
 * kernel returns the input class), sums in double in tap order, round half away from zero.
 * ================================================================================================ */
static double cubic_kernel(double x) { /* Keys, a = -0.5 */
    const double a = fabs(x), a2 = a * a, a3 = a2 * a;
    if (a <= 1.0) return (1.5 * a3 - 2.5 * a2) + 1.0;
    if (a <= 2.0) return ((-0.5 * a3 + 2.5 * a2) - 4.0 * a) + 2.0;
    return 0.0;
}
static double tri_kernel(double x) {
    const double a = fabs(x);
    return a < 1.0 ? 1.0 - a : 0.0;
}
/* taps of output sample x (0-based): returns P, first 1-based index in *left, weights (sum 1) */
static int u8_taps(int in_len, int x, double scale, int bicubic, int* left, double* wts /* >= 64 */) {
    (void)in_len;
    const double kw0 = bicubic ? 4.0 : 2.0;
    const double kw = scale < 1.0 ? kw0 / scale : kw0;
    const double u = (double)(x + 1) / scale + 0.5 * (1.0 - 1.0 / scale);
    *left = (int)floor(u - kw / 2.0);
    int P = (int)ceil(kw) + 2;
    if (P > 64) P = 64;
    double s = 0;
    for (int t = 0; t < P; ++t) {
        const double dx = u - (double)(*left + t);
        const double arg = scale < 1.0 ? scale * dx : dx;
        double v = bicubic ? cubic_kernel(arg) : tri_kernel(arg);
        if (scale < 1.0) v = scale * v;
        wts[t] = v;
        s += v;
    }
    for (int t = 0; t < P; ++t) wts[t] = wts[t] / s;
    return P;
}
static uint8_t sat_u8(double v) {
    double r = v < 0 ? -floor(-v + 0.5) : floor(v + 0.5);
    if (!(r > 0)) r = 0;
    if (r > 255) r = 255;
    return (uint8_t)r;
}
static void u8_resize_dim(const uint8_t* in, int h, int w, int C, int dim, int out_len, double scale, int bicubic,
                          uint8_t* out) {
    const int in_len = dim == 0 ? h : w;
    const int oh = dim == 0 ? out_len : h, ow = dim == 0 ? w : out_len;
    for (int o = 0; o < out_len; ++o) {
        int left;
        double wts[64];
        const int P = u8_taps(in_len, o, scale, bicubic, &left, wts);
        const int other = dim == 0 ? w : h;
        for (int q = 0; q < other; ++q)
            for (int c = 0; c < C; ++c) {
                double acc = 0;
                for (int t = 0; t < P; ++t) {
                    int idx = left + t;
                    idx = idx < 1 ? 1 : (idx > in_len ? in_len : idx);
                    const uint8_t v = dim == 0 ? in[((size_t)(idx - 1) * w + q) * C + c] : in[((size_t)q * w + (idx - 1)) * C + c];
                    acc = acc + wts[t] * (double)v;
                }
                if (dim == 0)
                    out[((size_t)o * ow + q) * C + c] = sat_u8(acc);
                else
                    out[((size_t)q * ow + o) * C + c] = sat_u8(acc);
            }
    }
    (void)oh;
}
ORC_API void orc_imresize_u8(const uint8_t* in, int h, int w, int C, int oh, int ow, double scale_r, double scale_c,
                             int bicubic, uint8_t* out) {
    if (scale_r <= scale_c) {
        uint8_t* tmp = (uint8_t*)malloc((size_t)oh * w * C);
        u8_resize_dim(in, h, w, C, 0, oh, scale_r, bicubic, tmp);
        u8_resize_dim(tmp, oh, w, C, 1, ow, scale_c, bicubic, out);
        free(tmp);
    } else {
        uint8_t* tmp = (uint8_t*)malloc((size_t)h * ow * C);
        u8_resize_dim(in, h, w, C, 1, ow, scale_c, bicubic, tmp);
        u8_resize_dim(tmp, h, ow, C, 0, oh, scale_r, bicubic, out);
        free(tmp);
    }
}
