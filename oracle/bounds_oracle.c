/* bounds_oracle.c — TEST INFRASTRUCTURE (CPU oracle), never linked into or called by the product.
 *
 * Plain-C restatement of the canvas geometry of PP/renderPanorama/renderPanorama.m ("PP/" = /root/reference/Procedural
 * Program/): the auto-reference search and canvas sizing (:84-232), cropNonzeroBbox (:1459-1504) and the four bounds
 * functions cylindricalBounds (:1507-1542), sphericalBounds (:1544-1579), planarBounds (:1581-1665),
 * stereographicBounds (:1667-1754).  Written from those lines, independently of the product's numpy mirror
 * (<pkg>/renderPanorama.py), which tests/test_bounds_oracle.py compares against it for every mode.
 *
 * Semantics fixed here where MATLAB leaves them to the toolbox:
 *   K \ x         : K is upper triangular ([f 0 cx; 0 f cy; 0 0 1] in the reference) -> back substitution; a general
 *                   3 x 3 K goes through Cramer's rule.  (MATLAB's mldivide picks the triangular solver for such a K.)
 *   linspace(a,b,n): a + (b-a)*k/(n-1), last point exactly b (MATLAB's documented end-point behaviour).
 *   prctile       : sorted samples sit at 100*(i-0.5)/n percent, linear interpolation in between, clamped outside.
 *   rgb2gray(u8)  : floor(0.298936021293775 R + 0.587043074451121 G + 0.114020904255103 B + 0.5).
 * PARITY UNPINNED: the reference ships no fixtures for these functions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { MODE_CYL = 0, MODE_SPH = 1, MODE_PLANAR = 2, MODE_STEREO = 3 };

/* column-major 3x3: M[r + 3c] */
static void solve_K(const double* K, const double b[3], double x[3]) {
    if (K[1] == 0.0 && K[2] == 0.0 && K[5] == 0.0) { /* upper triangular: back substitution */
        x[2] = b[2] / K[8];
        x[1] = (b[1] - K[7] * x[2]) / K[4];
        x[0] = (b[0] - K[3] * x[1] - K[6] * x[2]) / K[0];
        return;
    }
    const double a = K[0], d = K[3], g = K[6], b_ = K[1], e = K[4], h = K[7], c = K[2], f = K[5], i = K[8];
    const double det = a * (e * i - h * f) - d * (b_ * i - h * c) + g * (b_ * f - e * c);
    x[0] = (b[0] * (e * i - h * f) - d * (b[1] * i - h * b[2]) + g * (b[1] * f - e * b[2])) / det;
    x[1] = (a * (b[1] * i - h * b[2]) - b[0] * (b_ * i - h * c) + g * (b_ * b[2] - b[1] * c)) / det;
    x[2] = (a * (e * b[2] - b[1] * f) - d * (b_ * b[2] - b[1] * c) + b[0] * (b_ * f - e * c)) / det;
}

static double lin(double a, double b, int k, int n) {
    if (n == 1) return b;
    if (k == n - 1) return b;
    return a + (b - a) * (double)k / (double)(n - 1);
}

/* world rays of the sample grid of one camera: 48 x 32 interior (column-major order of meshgrid's U(:)) and, when
 * border != 0, 4 x border edge samples.  Returns the count; rays[3*q + {0,1,2}]. */
static int grid_rays(const double* K, const double* R, double H, double W, int border, double* rays) {
    const int nx = 48, ny = 32;
    int q = 0;
    const int total = nx * ny + 4 * border;
    for (int s = 0; s < total; ++s) {
        double u, v;
        if (s < nx * ny) {
            const int ix = s / ny, iy = s % ny; /* U(:) walks down the rows of each column */
            u = lin(1.0, W, ix, nx);
            v = lin(1.0, H, iy, ny);
        } else {
            const int e = s - nx * ny, side = e / border, k = e % border;
            const double xb = lin(1.0, W, k, border), yb = lin(1.0, H, k, border);
            if (side == 0) { u = xb; v = 1.0; }
            else if (side == 1) { u = xb; v = H; }
            else if (side == 2) { u = 1.0; v = yb; }
            else { u = W; v = yb; }
        }
        const double b[3] = {u, v, 1.0};
        double c[3];
        solve_K(K, b, c);
        /* rayW = R' * rayC */
        for (int r = 0; r < 3; ++r) rays[3 * q + r] = R[3 * r + 0] * c[0] + R[3 * r + 1] * c[1] + R[3 * r + 2] * c[2];
        ++q;
    }
    return q;
}

static int cmp_d(const void* a, const void* b) {
    const double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
static double prctile_sorted(const double* x, int n, double p) {
    double pos = p / 100.0 * (double)n + 0.5;
    if (pos < 1.0) pos = 1.0;
    if (pos > (double)n) pos = (double)n;
    const int lo = (int)floor(pos);
    const int hi = lo + 1 < n ? lo + 1 : n;
    return x[lo - 1] + (pos - (double)lo) * (x[hi - 1] - x[lo - 1]);
}

/* out = {aMin, aMax, bMin, bMax}.  K, R: n x (3x3 column-major); sizes: n x {H, W}. */
ORC_API void orc_bounds(int mode, int n, const double* K, const double* R, const double* sizes, const double* Rref,
                        double pct_lo, double pct_hi, double abs_cap, double* out) {
    double amin = INFINITY, amax = -INFINITY, bmin = INFINITY, bmax = -INFINITY;
    const int border = (mode == MODE_PLANAR || mode == MODE_STEREO) ? 512 : 0;
    const int cap = 48 * 32 + 4 * 512;
    double* rays = (double*)malloc(sizeof(double) * 3 * (size_t)cap);
    double* av = (double*)malloc(sizeof(double) * (size_t)cap);
    double* bv = (double*)malloc(sizeof(double) * (size_t)cap);
    for (int i = 0; i < n; ++i) {
        const int m = grid_rays(K + 9 * i, R + 9 * i, sizes[2 * i], sizes[2 * i + 1], border, rays);
        if (mode == MODE_CYL || mode == MODE_SPH) {
            for (int q = 0; q < m; ++q) {
                const double x = rays[3 * q], y = rays[3 * q + 1], z = rays[3 * q + 2];
                const double th = atan2(x, z);
                const double b = mode == MODE_CYL ? y / hypot(x, z) : atan2(y, hypot(x, z));
                if (th < amin) amin = th;
                if (th > amax) amax = th;
                if (b < bmin) bmin = b;
                if (b > bmax) bmax = b;
            }
            continue;
        }
        int cnt = 0;
        for (int q = 0; q < m; ++q) {
            double r[3];
            for (int k = 0; k < 3; ++k) /* rayR = Rref * rayW */
                r[k] = Rref[k] * rays[3 * q] + Rref[k + 3] * rays[3 * q + 1] + Rref[k + 6] * rays[3 * q + 2];
            double a, b;
            if (mode == MODE_PLANAR) {
                if (!(r[2] > 1e-4)) continue;
                a = r[0] / r[2];
                b = r[1] / r[2];
            } else {
                const double nr = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
                const double xr = r[0] / nr, yr = r[1] / nr, zr = r[2] / nr;
                const double den = 1.0 + zr;
                if (!(den > 1e-6)) continue;
                a = xr / den;
                b = yr / den;
            }
            if (isfinite(abs_cap) && abs_cap > 0) {
                a = fmax(-abs_cap, fmin(abs_cap, a));
                b = fmax(-abs_cap, fmin(abs_cap, b));
            }
            av[cnt] = a;
            bv[cnt] = b;
            ++cnt;
        }
        if (cnt == 0) continue;
        qsort(av, (size_t)cnt, sizeof(double), cmp_d);
        qsort(bv, (size_t)cnt, sizeof(double), cmp_d);
        const double alo = prctile_sorted(av, cnt, pct_lo), ahi = prctile_sorted(av, cnt, pct_hi);
        const double blo = prctile_sorted(bv, cnt, pct_lo), bhi = prctile_sorted(bv, cnt, pct_hi);
        if (alo < amin) amin = alo;
        if (ahi > amax) amax = ahi;
        if (blo < bmin) bmin = blo;
        if (bhi > bmax) bmax = bhi;
    }
    if (mode == MODE_PLANAR || mode == MODE_STEREO) { /* safety fallbacks (:1657-1663, :1746-1752) */
        if (!isfinite(amin) || !isfinite(amax) || amin >= amax) { amin = -1; amax = 1; }
        if (!isfinite(bmin) || !isfinite(bmax) || bmin >= bmax) { bmin = -1; bmax = 1; }
    }
    out[0] = amin; out[1] = amax; out[2] = bmin; out[3] = bmax;
    free(rays); free(av); free(bv);
}

typedef struct orc_geo_opts {
    double f_pan, res_scale, margin, max_megapixel, pct_lo, pct_hi, uv_abs_cap, pixel_pad;
    int auto_ref;
} orc_geo_opts;

static double dmax(double a, double b) { return a > b ? a : b; }

/* renderPanorama.m:84-232.  ref_idx 0-based in/out.  out = {W, H, origin0, origin1, resScale}. */
ORC_API void orc_canvas_geometry(int mode, int n, const double* K, const double* R, const double* sizes, int* ref_idx,
                                 const orc_geo_opts* o, double* out) {
    const double f = o->f_pan;
    double rs = o->res_scale;
    if ((mode == MODE_PLANAR || mode == MODE_STEREO) && o->auto_ref) {
        double best = INFINITY;
        int best_idx = *ref_idx;
        for (int ii = 0; ii < n; ++ii) {
            double b[4], area;
            orc_bounds(mode, n, K, R, sizes, R + 9 * ii, o->pct_lo, o->pct_hi, o->uv_abs_cap, b);
            if (mode == MODE_STEREO) {
                double ext = dmax(dmax(fabs(b[0]), fabs(b[1])), dmax(fabs(b[2]), fabs(b[3])));
                ext = ext * (1 + 2 * o->margin) + o->pixel_pad / f;
                const double Wi = dmax(1, ceil(2 * f * ext * rs));
                area = Wi * Wi;
            } else {
                const double du = b[1] - b[0], dv = b[3] - b[2];
                const double u0 = b[0] - o->margin * du - o->pixel_pad / f, u1 = b[1] + o->margin * du + o->pixel_pad / f;
                const double v0 = b[2] - o->margin * dv - o->pixel_pad / f, v1 = b[3] + o->margin * dv + o->pixel_pad / f;
                area = dmax(1, ceil(f * (u1 - u0) * rs)) * dmax(1, ceil(f * (v1 - v0) * rs));
            }
            if (area < best) { best = area; best_idx = ii; }
        }
        *ref_idx = best_idx;
    }
    double b[4];
    orc_bounds(mode, n, K, R, sizes, R + 9 * (*ref_idx), o->pct_lo, o->pct_hi, o->uv_abs_cap, b);
    double a0 = b[0], a1 = b[1], b0 = b[2], b1 = b[3];
    if (mode == MODE_STEREO) { /* centred square (:196-201) */
        const double ext = dmax(dmax(fabs(a0), fabs(a1)), dmax(fabs(b0), fabs(b1)));
        a0 = -ext; a1 = ext; b0 = -ext; b1 = ext;
    }
    const double da = a1 - a0, db = b1 - b0;
    a0 = a0 - o->margin * da; a1 = a1 + o->margin * da;
    b0 = b0 - o->margin * db; b1 = b1 + o->margin * db;
    if (mode == MODE_PLANAR || mode == MODE_STEREO) {
        a0 = a0 - o->pixel_pad / f; a1 = a1 + o->pixel_pad / f;
        b0 = b0 - o->pixel_pad / f; b1 = b1 + o->pixel_pad / f;
    }
    double W = dmax(1, ceil(f * (a1 - a0) * rs)), H = dmax(1, ceil(f * (b1 - b0) * rs));
    if (mode == MODE_PLANAR || mode == MODE_STEREO) {
        const double max_pixel = round(o->max_megapixel * 1e6);
        if (H * W > max_pixel) {
            const double s = sqrt(max_pixel / (H * W));
            rs = rs * s;
            W = dmax(1, ceil(f * (a1 - a0) * rs));
            H = dmax(1, ceil(f * (b1 - b0) * rs));
        }
    }
    out[0] = W; out[1] = H; out[2] = a0; out[3] = b0; out[4] = rs;
}

/* cropNonzeroBbox (:1459-1504): rect = {r1, r2, c1, c2} 1-based inclusive; returns didCrop. img: h x w x 3 row-major. */
ORC_API int orc_crop_nonzero_bbox(const uint8_t* img, int64_t h, int64_t w, int white, int64_t* rect) {
    int64_t rmin = h, rmax = -1, cmin = w, cmax = -1;
    for (int64_t y = 0; y < h; ++y)
        for (int64_t x = 0; x < w; ++x) {
            const uint8_t* p = img + 3 * (y * w + x);
            double g = 0.298936021293775 * (double)p[0];
            g = g + 0.587043074451121 * (double)p[1];
            g = g + 0.114020904255103 * (double)p[2];
            const double G = floor(g + 0.5);
            const int fg = white ? (G < 255.0) : (G > 0.0);
            if (fg) {
                if (y < rmin) rmin = y;
                if (y > rmax) rmax = y;
                if (x < cmin) cmin = x;
                if (x > cmax) cmax = x;
            }
        }
    if (rmax < 0) {
        rect[0] = 1; rect[1] = h; rect[2] = 1; rect[3] = w;
        return 0;
    }
    const int64_t pad = 6;
    rect[0] = rmin + 1 - pad < 1 ? 1 : rmin + 1 - pad;
    rect[1] = rmax + 1 + pad > h ? h : rmax + 1 + pad;
    rect[2] = cmin + 1 - pad < 1 ? 1 : cmin + 1 - pad;
    rect[3] = cmax + 1 + pad > w ? w : cmax + 1 + pad;
    return 1;
}
