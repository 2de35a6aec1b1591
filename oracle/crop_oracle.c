// crop_oracle.c — CPU restatement of the panorama crop.
//
// TEST INFRASTRUCTURE ONLY: called from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
// checker for the HIP path; never linked into or called by the product library.
//
// Follows PP/imageProcessing/panoramaCropper.m:
//   :73-84   gray = rgb2gray(stitchedImage); BW = imbinarize(gray, blackRange/255)   ("white": whiteRange, complemented)
//   :89      BW2 = imfill(BW, 'holes')
//   :92-97   pixels outside BW2 count as canvas (max channel < 0)
//   :99-151  the line-by-line "largest rectangle under a histogram" scan with its pointer-jumping left/right arrays
//   :153-165 cropH = hh + 1, cropW = rr - ll + 1, offsetx = ll, offsety = nl - hh + 1; the crop
//            stitchedImage(offsety:offsety+cropH, offsetx:offsetx+cropW, :) falls back to the input when out of range
//
// PARITY UNPINNED for the three toolbox calls (no MATLAB in this image, the reference ships no vectors); their
// semantics are fixed here once and mirrored by csrc/crop.hip:
//   rgb2gray (uint8)  : floor(0.298936021293775 R + 0.587043074451121 G + 0.114020904255103 B + 0.5) in f64, R then G then B
//   imbinarize(g, t)  : (double)g > (t * 255.0) with t = range / 255.0, both in f64
//   imfill(.,'holes') : background pixels not 4-connected to the image border through background become foreground
// The scan itself is restated literally, quirks included: `right` is only computed for k = w-1 .. 1 and never reaches
// column w (right(w) stays 0), the strict `maxarea < val` keeps the FIRST maximum in (line, k) order, and the crop
// takes cropH + 1 rows and cropW + 1 columns.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline uint8_t gray_of(const uint8_t* p) {
    double g = 0.298936021293775 * (double)p[0];
    g = g + 0.587043074451121 * (double)p[1];
    g = g + 0.114020904255103 * (double)p[2];
    g = g + 0.5;
    int v = (int)g;  // g >= 0: truncation == floor
    return (uint8_t)(v > 255 ? 255 : v);
}

// inside[h*w] = BW2 of panoramaCropper.m:89 (1 = content or enclosed hole)
ORC_API void orc_crop_inside(const uint8_t* rgb, int64_t h, int64_t w, int canvas_white, double range, uint8_t* inside) {
    const double t = (range / 255.0) * 255.0;
    const int64_t n = h * w;
    uint8_t* bw = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        const int fg = (double)gray_of(rgb + 3 * i) > t;
        bw[i] = (uint8_t)(canvas_white ? !fg : fg);
    }
    // flood the background from the border, 4-connected
    int64_t* stack = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t sp = 0;
    memset(inside, 1, (size_t)n);
#define ORC_PUSH(r, c)                                   \
    do {                                                 \
        const int64_t q_ = (r) * w + (c);                \
        if (!bw[q_] && inside[q_]) {                     \
            inside[q_] = 0;                              \
            stack[sp++] = q_;                            \
        }                                                \
    } while (0)
    for (int64_t c = 0; c < w; ++c) {
        ORC_PUSH(0, c);
        ORC_PUSH(h - 1, c);
    }
    for (int64_t r = 0; r < h; ++r) {
        ORC_PUSH(r, 0);
        ORC_PUSH(r, w - 1);
    }
    while (sp > 0) {
        const int64_t q = stack[--sp];
        const int64_t r = q / w, c = q - r * w;
        if (r > 0) ORC_PUSH(r - 1, c);
        if (r + 1 < h) ORC_PUSH(r + 1, c);
        if (c > 0) ORC_PUSH(r, c - 1);
        if (c + 1 < w) ORC_PUSH(r, c + 1);
    }
#undef ORC_PUSH
    free(stack);
    free(bw);
}

// out[0..3] = offsetx, offsety, cropW, cropH (1-based, panoramaCropper.m:153-157); out[4] = 1 iff the crop
// stitchedImage(offsety:offsety+cropH, offsetx:offsetx+cropW, :) is inside the image (else the reference warns and
// returns the input); out[5..8] = ll, rr, hh, nl (diagnostics).
ORC_API void orc_crop_rect(const uint8_t* rgb, int64_t h, int64_t w, int canvas_white, double range, int64_t* out) {
    const int64_t n = h * w;
    uint8_t* inside = (uint8_t*)malloc((size_t)(n > 0 ? n : 1));
    orc_crop_inside(rgb, h, w, canvas_white, range, inside);
    // 1-based arrays as in the reference
    int64_t* height = (int64_t*)calloc((size_t)w + 2, sizeof(int64_t));
    int64_t* left = (int64_t*)calloc((size_t)w + 2, sizeof(int64_t));
    int64_t* right = (int64_t*)calloc((size_t)w + 2, sizeof(int64_t));
    int64_t maxarea = 0, ll = 0, rr = 0, hh = 0, nl = 0;
    for (int64_t line = 1; line <= h; ++line) {
        for (int64_t k = 1; k <= w; ++k) {
            if (!inside[(line - 1) * w + (k - 1)])
                height[k] = 0;
            else
                height[k] = height[k] + 1;
        }
        for (int64_t k = 1; k <= w; ++k) {
            left[k] = k;
            while (left[k] > 1 && height[k] <= height[left[k] - 1]) left[k] = left[left[k] - 1];
        }
        for (int64_t k = w - 1; k >= 1; --k) {
            right[k] = k;
            while (right[k] < w - 1 && height[k] <= height[right[k] + 1]) right[k] = right[right[k] + 1];
        }
        for (int64_t k = 1; k <= w; ++k) {
            const int64_t val = (right[k] - left[k] + 1) * height[k];
            if (maxarea < val) {
                maxarea = val;
                ll = left[k];
                rr = right[k];
                hh = height[k];
                nl = line;
            }
        }
    }
    const int64_t cropH = hh + 1, cropW = rr - ll + 1, offsetx = ll, offsety = nl - hh + 1;
    out[0] = offsetx;
    out[1] = offsety;
    out[2] = cropW;
    out[3] = cropH;
    out[4] = offsetx >= 1 && offsety >= 1 && offsety + cropH <= h && offsetx + cropW <= w;
    out[5] = ll;
    out[6] = rr;
    out[7] = hh;
    out[8] = nl;
    free(height);
    free(left);
    free(right);
    free(inside);
}
