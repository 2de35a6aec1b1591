/*
 * sift_oracle.c — CPU restatement of the SIFT stage behind
 *   PP/featureMatching/getFeaturePoints.m:26-40,71-74
 *   (rgb2gray -> detectSIFTFeatures(gray,'NumLayersInOctave',L,ContrastThreshold,EdgeThreshold,Sigma)
 *    -> extractFeatures -> double(validPts.Location)).
 *
 * TEST INFRASTRUCTURE ONLY (see match_oracle.c).  PARITY UNPINNED: detectSIFTFeatures/extractFeatures
 * are closed MathWorks toolbox code (documented as built on OpenCV's SIFT) and the reference has no
 * tests.  The algorithm restated is the published one — Lowe 2004 in OpenCV cv::SIFT's parameterisation
 * (features2d/src/sift.dispatch.cpp + sift.simd.hpp, OpenCV 4.x; not present in /root/reference):
 *   base image: gray -> f32, 2x bilinear upsample, blur to sigma (assumed camera blur 0.5);
 *   octaves = round(log2(min(w,h) of the base)) - 2 + 1; L+3 Gaussians and L+2 DoGs per octave;
 *   extrema over 26 neighbours inside a 5-pixel border, <= 5 Newton steps, contrast test
 *   |D|*L < ContrastThreshold, edge test tr^2*e >= (e+1)^2*det; orientation histogram (36 bins,
 *   radius 4.5*scale, peaks >= 0.8*max, parabolic refinement); 4x4x8 descriptor with trilinear binning,
 *   0.2 clipping, scaling to 512 and saturation to 0..255; finally unit L2 normalisation (the toolbox
 *   returns single descriptors; either convention passes through matchFeaturesScratch.m:105-110).
 * Deliberate, documented deviations that make the result independent of evaluation order (so that the
 * HIP path can reproduce it bit for bit):
 *   - Gaussian taps are applied as k-ascending f32 fma chains (row pass, then column pass);
 *   - orientation and descriptor histograms accumulate in 2^-20 fixed point (int64);
 *   - exp, 2^x, atan2 (OpenCV's fastAtan2 polynomial) and sin/cos of the keypoint angle are the
 *     polynomials written below, not libm;
 *   - keypoints are emitted in ascending (octave, layer, row, col, orientation bin) order and exact
 *     duplicates (same refined cell) are dropped, in place of OpenCV's removeDuplicatedSorted.
 * MATLAB's rgb2gray for uint8: round(0.298936021293775 R + 0.587043074451121 G + 0.114020904255103 B).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#define SIFT_IMG_BORDER 5
#define SIFT_MAX_INTERP_STEPS 5
#define SIFT_ORI_HIST_BINS 36
#define SIFT_ORI_SIG_FCTR 1.5f
#define SIFT_ORI_RADIUS 4.5f
#define SIFT_ORI_PEAK_RATIO 0.8f
#define SIFT_DESCR_WIDTH 4
#define SIFT_DESCR_HIST_BINS 8
#define SIFT_DESCR_SCL_FCTR 3.0f
#define SIFT_DESCR_MAG_THR 0.2f
#define SIFT_INT_DESCR_FCTR 512.0f
#define FIX_SCALE 1048576.0f /* 2^20 */
#define FLT_EPS 1.1920928955078125e-07f

typedef struct {
    double sigma;
    int n_layers;
    double contrast_threshold;
    double edge_threshold;
    int max_features;
} orc_sift_params;

typedef struct {
    int w, h;
    float* d;
} Img;

static Img img_new(int w, int h) {
    Img m;
    m.w = w;
    m.h = h;
    m.d = (float*)malloc(sizeof(float) * (size_t)w * h);
    return m;
}

/* ---- self-contained elementary functions (same formulas in csrc/sift.hip) ----------------------- */
static inline float poly_exp2(float f) { /* 2^f, f in [-0.5, 0.5] */
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
static inline float my_exp2(float t) {
    if (t < -125.0f) return 0.0f;
    if (t > 125.0f) t = 125.0f;
    const float n = rintf(t);
    const float p = poly_exp2(t - n);
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)((int)n + 127) << 23;
    return p * s.f;
}
static inline float my_exp(float x) { return my_exp2(x * 1.4426950408889634f); }

static inline float fast_atan2_deg(float y, float x) { /* OpenCV hal fastAtan2, degrees in [0,360) */
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f;
    const float p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.0f - a;
    if (y < 0) a = 360.0f - a;
    return a;
}

static inline void sincos_deg(float a, float* s, float* c) { /* a in degrees */
    const float k = rintf(a / 90.0f);
    const float r = a - 90.0f * k; /* [-45, 45] */
    const float x = r * 0.017453292519943295f, x2 = x * x;
    float sp = 2.7557319223985893e-06f;
    sp = fmaf(sp, x2, -1.9841269841269841e-04f);
    sp = fmaf(sp, x2, 8.3333333333333332e-03f);
    sp = fmaf(sp, x2, -1.6666666666666666e-01f);
    sp = fmaf(sp * x2, x, x); /* x + x^3 * (...) */
    float cp = -2.7557319223985888e-07f;
    cp = fmaf(cp, x2, 2.4801587301587302e-05f);
    cp = fmaf(cp, x2, -1.3888888888888889e-03f);
    cp = fmaf(cp, x2, 4.1666666666666664e-02f);
    cp = fmaf(cp, x2, -0.5f);
    cp = fmaf(cp, x2, 1.0f);
    const int q = ((int)k % 4 + 4) % 4;
    if (q == 0) { *s = sp; *c = cp; }
    else if (q == 1) { *s = cp; *c = -sp; }
    else if (q == 2) { *s = -sp; *c = -cp; }
    else { *s = -cp; *c = sp; }
}

static inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

/* ---- Gaussian blur as OpenCV's GaussianBlur on CV_32F (kernel size and taps), fma chains ---------- */
static int gauss_kernel(double sigma, float* k /* >= 64 */) {
    int n = (int)lrint(sigma * 8.0 + 1.0) | 1; /* cvRound(sigma*4*2+1)|1 */
    if (n > 63) n = 63;
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) {
        const double x = i - (n - 1) * 0.5;
        k[i] = (float)exp(s2 * x * x);
        sum += k[i];
    }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
    return n;
}

static void gauss_blur(const Img* in, double sigma, Img* out) {
    float k[64];
    const int n = gauss_kernel(sigma, k), r = n / 2, w = in->w, h = in->h;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for schedule(static) if ((size_t)w * h > 16384)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float acc = 0.f;
            for (int t = 0; t < n; ++t) acc = fmaf(k[t], in->d[(size_t)y * w + reflect101(x + t - r, w)], acc);
            tmp[(size_t)y * w + x] = acc;
        }
#pragma omp parallel for schedule(static) if ((size_t)w * h > 16384)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            float acc = 0.f;
            for (int t = 0; t < n; ++t) acc = fmaf(k[t], tmp[(size_t)reflect101(y + t - r, h) * w + x], acc);
            out->d[(size_t)y * w + x] = acc;
        }
    free(tmp);
}

/* ---- keypoint record ------------------------------------------------------------------------------ */
typedef struct {
    int o, layer, r, c; /* refined cell */
    float xc, xr, xi, contr;
    int bin;     /* orientation peak bin (sort key) */
    float angle; /* OpenCV kpt.angle, degrees */
} Kp;

static int kp_cmp(const void* a, const void* b) {
    const Kp *p = (const Kp*)a, *q = (const Kp*)b;
    if (p->o != q->o) return p->o - q->o;
    if (p->layer != q->layer) return p->layer - q->layer;
    if (p->r != q->r) return p->r - q->r;
    if (p->c != q->c) return p->c - q->c;
    return p->bin - q->bin;
}

#define AT(im, rr, cc) ((im)->d[(size_t)(rr) * (im)->w + (cc)])

/* adjustLocalExtrema.  dog: the L+2 DoG images of this octave.  Returns 1 and fills kp on success. */
static int adjust_extremum(const Img* dog, int nl, int o, int layer, int r, int c, float contr_thr,
                           float edge_thr, Kp* kp) {
    const float img_scale = 1.0f / 255.0f, deriv_scale = img_scale * 0.5f, second_scale = img_scale,
                cross_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
    for (; i < SIFT_MAX_INTERP_STEPS; ++i) {
        const Img *im = &dog[layer], *pv = &dog[layer - 1], *nx = &dog[layer + 1];
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
        /* X = H \ dD by LU with partial pivoting on the 3x3 (Matx33f::solve(DECOMP_LU)) */
        float A[3][4] = {{dxx, dxy, dxs, dD0}, {dxy, dyy, dys, dD1}, {dxs, dys, dss, dD2}};
        int singular = 0;
        for (int col = 0; col < 3; ++col) {
            int piv = col;
            for (int row = col + 1; row < 3; ++row)
                if (fabsf(A[row][col]) > fabsf(A[piv][col])) piv = row;
            if (fabsf(A[piv][col]) < FLT_EPS) { singular = 1; break; }
            if (piv != col)
                for (int e = 0; e < 4; ++e) { const float t = A[piv][e]; A[piv][e] = A[col][e]; A[col][e] = t; }
            const float d = -1.0f / A[col][col];
            for (int row = col + 1; row < 3; ++row) {
                const float alpha = A[row][col] * d;
                for (int e = col + 1; e < 4; ++e) A[row][e] = fmaf(alpha, A[col][e], A[row][e]);
            }
        }
        float X[3] = {0, 0, 0};
        if (!singular) {
            for (int row = 2; row >= 0; --row) {
                float s = A[row][3];
                for (int e = row + 1; e < 3; ++e) s = s - A[row][e] * X[e];
                X[row] = s / A[row][row];
            }
        }
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > 7.158278826666667e8f || fabsf(xr) > 7.158278826666667e8f || fabsf(xc) > 7.158278826666667e8f) return 0;
        c += (int)lrintf(xc);
        r += (int)lrintf(xr);
        layer += (int)lrintf(xi);
        if (layer < 1 || layer > nl || c < SIFT_IMG_BORDER || c >= dog[0].w - SIFT_IMG_BORDER ||
            r < SIFT_IMG_BORDER || r >= dog[0].h - SIFT_IMG_BORDER)
            return 0;
    }
    if (i >= SIFT_MAX_INTERP_STEPS) return 0;
    {
        const Img *im = &dog[layer], *pv = &dog[layer - 1], *nx = &dog[layer + 1];
        const float dD0 = (AT(im, r, c + 1) - AT(im, r, c - 1)) * deriv_scale;
        const float dD1 = (AT(im, r + 1, c) - AT(im, r - 1, c)) * deriv_scale;
        const float dD2 = (AT(nx, r, c) - AT(pv, r, c)) * deriv_scale;
        const float t = (dD0 * xc + dD1 * xr) + dD2 * xi;
        const float contr = AT(im, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * (float)nl < contr_thr) return 0;
        const float v2 = AT(im, r, c) * 2.0f;
        const float dxx = (AT(im, r, c + 1) + AT(im, r, c - 1) - v2) * second_scale;
        const float dyy = (AT(im, r + 1, c) + AT(im, r - 1, c) - v2) * second_scale;
        const float dxy = (AT(im, r + 1, c + 1) - AT(im, r + 1, c - 1) - AT(im, r - 1, c + 1) + AT(im, r - 1, c - 1)) * cross_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0 || tr * tr * edge_thr >= (edge_thr + 1) * (edge_thr + 1) * det) return 0;
        kp->o = o; kp->layer = layer; kp->r = r; kp->c = c;
        kp->xc = xc; kp->xr = xr; kp->xi = xi; kp->contr = fabsf(contr);
    }
    return 1;
}

static float kp_scale_in_octave(float sigma, int layer, float xi, int nl) {
    return sigma * my_exp2(((float)layer + xi) / (float)nl); /* = kpt.size*0.5/(1<<octv) */
}

/* calcOrientationHist + peak search; appends oriented keypoints to out[]; returns how many */
static int orientations(const Img* g, const Kp* kp, float scl_octv, Kp* out) {
    const int n = SIFT_ORI_HIST_BINS;
    const int radius = (int)lrintf(SIFT_ORI_RADIUS * scl_octv);
    const float sig = SIFT_ORI_SIG_FCTR * scl_octv;
    const float expf_scale = -1.0f / (2.0f * sig * sig);
    int64_t acc[SIFT_ORI_HIST_BINS];
    for (int b = 0; b < n; ++b) acc[b] = 0;
    for (int i = -radius; i <= radius; ++i) {
        const int y = kp->r + i;
        if (y <= 0 || y >= g->h - 1) continue;
        for (int j = -radius; j <= radius; ++j) {
            const int x = kp->c + j;
            if (x <= 0 || x >= g->w - 1) continue;
            const float dx = AT(g, y, x + 1) - AT(g, y, x - 1);
            const float dy = AT(g, y - 1, x) - AT(g, y + 1, x);
            const float wgt = my_exp((float)(i * i + j * j) * expf_scale);
            const float ori = fast_atan2_deg(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy);
            int bin = (int)lrintf(((float)n / 360.0f) * ori);
            if (bin >= n) bin -= n;
            if (bin < 0) bin += n;
            acc[bin] += (int64_t)llrintf((wgt * mag) * FIX_SCALE);
        }
    }
    float th[SIFT_ORI_HIST_BINS + 4], hist[SIFT_ORI_HIST_BINS];
    for (int b = 0; b < n; ++b) th[b + 2] = (float)acc[b] * (1.0f / FIX_SCALE);
    th[0] = th[n]; th[1] = th[n + 1]; th[n + 2] = th[2]; th[n + 3] = th[3];
    float omax = 0;
    for (int b = 0; b < n; ++b) {
        hist[b] = (th[b] + th[b + 4]) * (1.0f / 16.0f) + (th[b + 1] + th[b + 3]) * (4.0f / 16.0f) + th[b + 2] * (6.0f / 16.0f);
        if (b == 0 || hist[b] > omax) omax = hist[b];
    }
    const float thr = omax * SIFT_ORI_PEAK_RATIO;
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
        const int l = j > 0 ? j - 1 : n - 1, r2 = j < n - 1 ? j + 1 : 0;
        if (hist[j] > hist[l] && hist[j] > hist[r2] && hist[j] >= thr) {
            float bin = (float)j + 0.5f * (hist[l] - hist[r2]) / (hist[l] - 2.0f * hist[j] + hist[r2]);
            bin = bin < 0 ? (float)n + bin : (bin >= (float)n ? bin - (float)n : bin);
            float angle = 360.0f - (360.0f / (float)n) * bin;
            if (fabsf(angle - 360.0f) < FLT_EPS) angle = 0.0f;
            out[cnt] = *kp;
            out[cnt].bin = j;
            out[cnt].angle = angle;
            ++cnt;
        }
    }
    return cnt;
}

/* calcSIFTDescriptor + unit normalisation; dst[128] */
static void descriptor(const Img* g, float ptx, float pty, float ori, float scl, float* dst) {
    const int d = SIFT_DESCR_WIDTH, n = SIFT_DESCR_HIST_BINS;
    const int px = (int)lrintf(ptx), py = (int)lrintf(pty);
    float sin_t, cos_t;
    sincos_deg(ori, &sin_t, &cos_t);
    const float bins_per_deg = (float)n / 360.0f;
    const float exp_scale = -1.0f / ((float)(d * d) * 0.5f);
    const float hist_width = SIFT_DESCR_SCL_FCTR * scl;
    int radius = (int)lrintf(hist_width * 1.4142135623730951f * (float)(d + 1) * 0.5f);
    const int diag = (int)sqrt((double)g->w * g->w + (double)g->h * g->h);
    if (radius > diag) radius = diag;
    cos_t = cos_t / hist_width;
    sin_t = sin_t / hist_width;
    int64_t hist[(SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_WIDTH + 2) * (SIFT_DESCR_HIST_BINS + 2)];
    memset(hist, 0, sizeof hist);
    for (int i = -radius; i <= radius; ++i)
        for (int j = -radius; j <= radius; ++j) {
            const float c_rot = (float)j * cos_t - (float)i * sin_t;
            const float r_rot = (float)j * sin_t + (float)i * cos_t;
            float rbin = r_rot + (float)(d / 2) - 0.5f;
            float cbin = c_rot + (float)(d / 2) - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1 && rbin < d && cbin > -1 && cbin < d && r > 0 && r < g->h - 1 && c > 0 && c < g->w - 1)) continue;
            const float dx = AT(g, r, c + 1) - AT(g, r, c - 1);
            const float dy = AT(g, r - 1, c) - AT(g, r + 1, c);
            const float wgt = my_exp((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            const float o_deg = fast_atan2_deg(dy, dx);
            const float mag = sqrtf(dx * dx + dy * dy) * wgt;
            float obin = (o_deg - ori) * bins_per_deg;
            const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
            int o0 = (int)floorf(obin);
            rbin -= (float)r0; cbin -= (float)c0; obin -= (float)o0;
            if (o0 < 0) o0 += n;
            if (o0 >= n) o0 -= n;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
            const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
            const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
            const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
            const int idx = ((r0 + 1) * (d + 2) + c0 + 1) * (n + 2) + o0;
#define ADD(off, v) hist[idx + (off)] += (int64_t)llrintf((v) * FIX_SCALE)
            ADD(0, v000); ADD(1, v001);
            ADD(n + 2, v010); ADD(n + 3, v011);
            ADD((d + 2) * (n + 2), v100); ADD((d + 2) * (n + 2) + 1, v101);
            ADD((d + 3) * (n + 2), v110); ADD((d + 3) * (n + 2) + 1, v111);
#undef ADD
        }
    float raw[128];
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const int idx = ((i + 1) * (d + 2) + (j + 1)) * (n + 2);
            hist[idx] += hist[idx + n];
            hist[idx + 1] += hist[idx + n + 1];
            for (int k = 0; k < n; ++k) raw[(i * d + j) * n + k] = (float)hist[idx + k] * (1.0f / FIX_SCALE);
        }
    float nrm2 = 0;
    for (int k = 0; k < 128; ++k) nrm2 = fmaf(raw[k], raw[k], nrm2);
    const float thr = sqrtf(nrm2) * SIFT_DESCR_MAG_THR;
    nrm2 = 0;
    for (int k = 0; k < 128; ++k) {
        const float v = raw[k] < thr ? raw[k] : thr;
        raw[k] = v;
        nrm2 = fmaf(v, v, nrm2);
    }
    const float sn = sqrtf(nrm2);
    const float scale = SIFT_INT_DESCR_FCTR / (sn > FLT_EPS ? sn : FLT_EPS);
    float q[128], qq = 0;
    for (int k = 0; k < 128; ++k) { /* saturate_cast<uchar>(x): round half to even, clamp */
        float v = rintf(raw[k] * scale);
        v = v < 0 ? 0 : (v > 255 ? 255 : v);
        q[k] = v;
        qq = fmaf(v, v, qq);
    }
    const float inv = sqrtf(qq);
    for (int k = 0; k < 128; ++k) dst[k] = inv > 0 ? q[k] / inv : 0.0f;
}

/* rgb2gray (uint8) -> f32 gray on the 0..255 scale */
static void to_gray(const uint8_t* img, int h, int w, int C, float* g) {
    for (size_t p = 0; p < (size_t)h * w; ++p) {
        if (C == 1) { g[p] = (float)img[p]; continue; }
        const double v = 0.298936021293775 * img[3 * p] + 0.587043074451121 * img[3 * p + 1] + 0.114020904255103 * img[3 * p + 2];
        g[p] = (float)floor(v + 0.5);
    }
}

/* cv::resize(..., 2x, INTER_LINEAR) on f32 */
static void upsample2(const float* in, int h, int w, Img* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < 2 * h; ++y) {
        float fy = ((float)y + 0.5f) * 0.5f - 0.5f;
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        if (sy < 0) { sy = 0; fy = 0; }
        if (sy >= h - 1) { sy = h - 1; fy = 0; }
        const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
        for (int x = 0; x < 2 * w; ++x) {
            float fx = ((float)x + 0.5f) * 0.5f - 0.5f;
            int sx = (int)floorf(fx);
            fx -= (float)sx;
            if (sx < 0) { sx = 0; fx = 0; }
            if (sx >= w - 1) { sx = w - 1; fx = 0; }
            const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
            const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
            const float h0 = in[(size_t)sy * w + sx] * a0 + in[(size_t)sy * w + sx1] * a1;
            const float h1 = in[(size_t)sy1 * w + sx] * a0 + in[(size_t)sy1 * w + sx1] * a1;
            out->d[(size_t)y * out->w + x] = h0 * b0 + h1 * b1;
        }
    }
}

/* Exposed for unit tests of the building blocks */
ORC_API float orc_sift_exp(float x) { return my_exp(x); }
ORC_API float orc_sift_atan2(float y, float x) { return fast_atan2_deg(y, x); }
ORC_API void orc_sift_sincos(float a, float* s, float* c) { sincos_deg(a, s, c); }
ORC_API void orc_sift_blur(const float* in, int h, int w, double sigma, float* out) {
    Img a = {w, h, (float*)in}, b = {w, h, out};
    gauss_blur(&a, sigma, &b);
}

/* Pyramid accessor for tests: returns the number of octaves; fills sizes if non-NULL */
ORC_API int orc_sift_num_octaves(int H, int W) {
    const int mn = (2 * W < 2 * H ? 2 * W : 2 * H);
    return (int)lrint(log((double)mn) / log(2.0) - 2.0) + 1;
}

/*
 * The whole stage.  img: uint8 H x W x C row-major interleaved.  Outputs (row-major): desc count x 128,
 * loc count x 2 [x y] 1-based, aux count x 4 [size, angle, response, octave + 256*layer].
 * Returns the number of features found (may exceed cap; only the first cap are written).
 */
ORC_API int64_t orc_sift(const uint8_t* img, int H, int W, int C, const orc_sift_params* prm, float* desc,
                         double* loc, float* aux, int64_t cap) {
    const int nl = prm->n_layers;
    const float sigma = (float)prm->sigma;
    float* gray = (float*)malloc(sizeof(float) * (size_t)H * W);
    to_gray(img, H, W, C, gray);
    Img up = img_new(2 * W, 2 * H), base = img_new(2 * W, 2 * H);
    upsample2(gray, H, W, &up);
    free(gray);
    {
        double sd = prm->sigma * prm->sigma - 4.0 * 0.5 * 0.5;
        if (sd < 0.01) sd = 0.01;
        gauss_blur(&up, sqrt(sd), &base);
    }
    free(up.d);
    const int n_oct = orc_sift_num_octaves(H, W);
    if (n_oct <= 0) { free(base.d); return 0; }
    double sig[16];
    sig[0] = prm->sigma;
    const double kf = pow(2.0, 1.0 / nl);
    for (int i = 1; i < nl + 3; ++i) {
        const double sp = pow(kf, (double)(i - 1)) * prm->sigma, st = sp * kf;
        sig[i] = sqrt(st * st - sp * sp);
    }
    Img* G = (Img*)calloc((size_t)n_oct * (nl + 3), sizeof(Img));
    Img* D = (Img*)calloc((size_t)n_oct * (nl + 2), sizeof(Img));
    for (int o = 0; o < n_oct; ++o) {
        for (int i = 0; i < nl + 3; ++i) {
            Img* dst = &G[o * (nl + 3) + i];
            if (o == 0 && i == 0) {
                *dst = base;
            } else if (i == 0) {
                const Img* src = &G[(o - 1) * (nl + 3) + nl];
                const int w2 = src->w / 2, h2 = src->h / 2;
                *dst = img_new(w2 > 0 ? w2 : 1, h2 > 0 ? h2 : 1);
                for (int y = 0; y < dst->h; ++y)
                    for (int x = 0; x < dst->w; ++x) /* INTER_NEAREST at exactly 1/2 */
                        dst->d[(size_t)y * dst->w + x] = src->d[(size_t)(2 * y < src->h ? 2 * y : src->h - 1) * src->w + (2 * x < src->w ? 2 * x : src->w - 1)];
            } else {
                const Img* src = &G[o * (nl + 3) + i - 1];
                *dst = img_new(src->w, src->h);
                gauss_blur(src, sig[i], dst);
            }
        }
        for (int i = 0; i < nl + 2; ++i) {
            const Img *a = &G[o * (nl + 3) + i], *b = &G[o * (nl + 3) + i + 1];
            Img* dd = &D[o * (nl + 2) + i];
            *dd = img_new(a->w, a->h);
            for (size_t p = 0; p < (size_t)a->w * a->h; ++p) dd->d[p] = b->d[p] - a->d[p];
        }
    }
    /* ---- extrema + refinement ---- */
    const int thr_i = (int)floor(0.5 * prm->contrast_threshold / nl * 255.0);
    const float thr = (float)thr_i;
    size_t kcap = 1 << 16, kn = 0;
    Kp* kps = (Kp*)malloc(sizeof(Kp) * kcap);
    for (int o = 0; o < n_oct; ++o) {
        const Img* dog = &D[o * (nl + 2)];
        const int w = dog[0].w, h = dog[0].h;
        for (int layer = 1; layer <= nl; ++layer) {
            const Img *im = &dog[layer], *pv = &dog[layer - 1], *nx = &dog[layer + 1];
#pragma omp parallel for schedule(dynamic, 16) if ((size_t)w * h > 16384)
            for (int r = SIFT_IMG_BORDER; r < h - SIFT_IMG_BORDER; ++r)
                for (int c = SIFT_IMG_BORDER; c < w - SIFT_IMG_BORDER; ++c) {
                    const float val = AT(im, r, c);
                    if (!(fabsf(val) > thr)) continue;
                    int is_max = val > 0, is_min = val < 0;
                    for (int dr = -1; dr <= 1 && (is_max || is_min); ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            const float a = AT(pv, r + dr, c + dc), b = AT(nx, r + dr, c + dc), m = AT(im, r + dr, c + dc);
                            if (!(val >= a && val >= b && val >= m)) is_max = 0;
                            if (!(val <= a && val <= b && val <= m)) is_min = 0;
                        }
                    if (!(is_max || is_min)) continue;
                    Kp kp;
                    if (!adjust_extremum(dog, nl, o, layer, r, c, (float)prm->contrast_threshold, (float)prm->edge_threshold, &kp)) continue;
                    kp.bin = 0;
                    kp.angle = 0;
#pragma omp critical(kp_append) /* order is fixed by the sort below */
                    {
                        if (kn == kcap) { kcap *= 2; kps = (Kp*)realloc(kps, sizeof(Kp) * kcap); }
                        kps[kn++] = kp;
                    }
                }
        }
    }
    /* canonical order + drop exact duplicates (two extrema converging to the same refined cell) */
    qsort(kps, kn, sizeof(Kp), kp_cmp);
    size_t ku = 0;
    for (size_t i = 0; i < kn; ++i)
        if (ku == 0 || kp_cmp(&kps[ku - 1], &kps[i]) != 0) kps[ku++] = kps[i];
    kn = ku;
    /* ---- orientations ---- */
    size_t ocap = kn * 2 + 16, on = 0;
    Kp* oks = (Kp*)malloc(sizeof(Kp) * ocap);
    for (size_t i = 0; i < kn; ++i) {
        const Kp* kp = &kps[i];
        const float scl = kp_scale_in_octave(sigma, kp->layer, kp->xi, nl);
        Kp tmp[SIFT_ORI_HIST_BINS];
        const int c = orientations(&G[kp->o * (nl + 3) + kp->layer], kp, scl, tmp);
        if (on + c > ocap) { ocap = (on + c) * 2; oks = (Kp*)realloc(oks, sizeof(Kp) * ocap); }
        for (int e = 0; e < c; ++e) oks[on++] = tmp[e];
    }
    /* ---- descriptors ---- */
    const int64_t total = (int64_t)on;
    const int64_t nw = total < cap ? total : cap;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < nw; ++i) {
        const Kp* kp = &oks[i];
        const float scl = kp_scale_in_octave(sigma, kp->layer, kp->xi, nl);
        float ang = 360.0f - kp->angle;
        if (fabsf(ang - 360.0f) < FLT_EPS) ang = 0.0f;
        descriptor(&G[kp->o * (nl + 3) + kp->layer], (float)kp->c + kp->xc, (float)kp->r + kp->xr, ang, scl, desc + 128 * i);
        /* kpt.pt = (c+xc, r+xr) * 2^o in base coordinates, * 0.5 back to the input image, +1 for MATLAB */
        const float s = ldexpf(1.0f, kp->o) * 0.5f;
        loc[2 * i + 0] = (double)(((float)kp->c + kp->xc) * s) + 1.0;
        loc[2 * i + 1] = (double)(((float)kp->r + kp->xr) * s) + 1.0;
        if (aux) {
            aux[4 * i + 0] = scl * s * 2.0f; /* kpt.size after the 0.5 rescale */
            aux[4 * i + 1] = kp->angle;
            aux[4 * i + 2] = kp->contr;
            aux[4 * i + 3] = (float)(kp->o + 256 * kp->layer);
        }
    }
    for (int i = 0; i < n_oct * (nl + 3); ++i) free(G[i].d);
    for (int i = 0; i < n_oct * (nl + 2); ++i) free(D[i].d);
    free(G); free(D); free(kps); free(oks);
    return total;
}
