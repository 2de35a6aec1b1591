// crop.hip — the panorama crop rectangle on gfx950.
//
// Restates PP/imageProcessing/panoramaCropper.m:73-165: rgb2gray + imbinarize, imfill(.,'holes'), and the
// line-by-line "largest rectangle under a histogram" scan whose serial MATLAB loops visit every pixel of a
// >= 100 MPix canvas several times.  Semantics of the three toolbox calls and every quirk of the scan are fixed in
// oracle/crop_oracle.c (parity unpinned for the toolbox calls); results are integers and must be identical.
//
// Device formulation (all byte/bit work, HBM- and latency-bound; nothing here is a GEMM):
//   mask    : one wave per 64 pixels -> one 64-bit word of the background mask (ballot), 1 bit per pixel from here on
//   imfill  : "reachable background" grows from the border by alternating row passes (Kogge-Stone fill inside a word,
//             carry between words, one thread per row) and column passes (bitwise carry down/up a word column, split
//             into row chunks with a carry scan between them) until a full round changes nothing; at the fixed point
//             this is the 4-connected flood fill
//   heights : run length of inside pixels ending at each line (u16), per column, chunked the same way
//   scan    : one workgroup per line with the line's heights in LDS plus block minima over 32 and 1024 elements;
//             each element finds its nearest strictly smaller neighbour on both sides through those minima, which is
//             what the reference's pointer-jumping left/right arrays compute; (area, -k) max per line, then
//             (area, -line) max over lines = the reference's first maximum in (line, k) order
#include <algorithm>
#include <climits>
#include <cstdint>

#include "aps_internal.h"

namespace aps {

typedef unsigned long long u64;

__device__ __forceinline__ uint8_t crop_gray(uint8_t r, uint8_t g, uint8_t b) {
    double v = 0.298936021293775 * (double)r;
    v = v + 0.587043074451121 * (double)g;
    v = v + 0.114020904255103 * (double)b;
    v = v + 0.5;
    const int q = (int)v;
    return (uint8_t)(q > 255 ? 255 : q);
}

// bg[r][wd]: bit j set iff pixel (r, 64 wd + j) exists and is background (panoramaCropper.m:73-84)
__global__ __launch_bounds__(256) void crop_mask_kernel(const uint8_t* __restrict__ img, int64_t h, int64_t w, int layout,
                                                        int white, double t, int64_t W64, u64* __restrict__ bg) {
    const int lane = threadIdx.x & 63;
    const int64_t wd = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t r = blockIdx.y;
    if (wd >= W64) return;
    const int64_t col = wd * 64 + lane;
    bool isbg = false;
    if (col < w) {
        uint8_t p[3];
        if (layout == APS_IMG_U8_HWC) {
            const uint8_t* q = img + (r * w + col) * 3;
            p[0] = q[0];
            p[1] = q[1];
            p[2] = q[2];
        } else {  // MATLAB h x w x 3, column-major planes
            const int64_t plane = h * w, o = col * h + r;
            p[0] = img[o];
            p[1] = img[plane + o];
            p[2] = img[2 * plane + o];
        }
        const bool fg = (double)crop_gray(p[0], p[1], p[2]) > t;
        isbg = white ? fg : !fg;  // "white": BW = complement
    }
    const u64 m = __ballot(isbg);
    if (lane == 0) bg[r * W64 + wd] = m;
}

// seeds: background pixels on the image border
__global__ void crop_seed_kernel(const u64* __restrict__ bg, int64_t h, int64_t w, int64_t W64, u64* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= h * W64) return;
    const int64_t r = i / W64, wd = i - r * W64;
    u64 m = 0;
    if (r == 0 || r == h - 1) m = ~0ull;
    if (wd == 0) m |= 1ull;
    if (wd == (w - 1) >> 6) m |= 1ull << ((w - 1) & 63);
    out[i] = bg[i] & m;
}

// bits reachable from `seed` through `prop` moving towards higher (UP) or lower bit positions
template <bool UP>
__device__ __forceinline__ u64 fill_dir(u64 seed, u64 prop) {
    u64 g = seed & prop, p = prop;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        g |= p & (UP ? g << s : g >> s);
        p &= UP ? p << s : p >> s;
    }
    return g;
}

// one thread per row: left-to-right then right-to-left through the words of the row
__global__ void crop_fill_rows_kernel(const u64* __restrict__ bg, int64_t h, int64_t W64, u64* __restrict__ out,
                                      int* __restrict__ changed) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= h) return;
    const u64* b = bg + r * W64;
    u64* o = out + r * W64;
    bool ch = false;
    u64 carry = 0;
    for (int64_t wd = 0; wd < W64; ++wd) {
        const u64 old = o[wd];
        const u64 nw = old | fill_dir<true>(old | carry, b[wd]);
        carry = nw >> 63;
        if (nw != old) {
            o[wd] = nw;
            ch = true;
        }
    }
    carry = 0;
    for (int64_t wd = W64 - 1; wd >= 0; --wd) {
        const u64 old = o[wd];
        const u64 nw = old | fill_dir<false>(old | (carry << 63), b[wd]);
        carry = nw & 1ull;
        if (nw != old) {
            o[wd] = nw;
            ch = true;
        }
    }
    if (ch) *changed = 1;
}

// Column passes, 64 pixel columns at once (bitwise), in three steps so that the sequential depth is the chunk length
// and the chunk count instead of the image height: (A) per chunk of kChunk rows and word column, what reaches the
// chunk's far end from seeds inside it (G) and which bits pass straight through (P = AND of the background);
// (B) per word column a serial carry scan over the chunks, carry' = G | (P & carry); (C) per chunk the pass itself
// with its carry-in.  DOWN = top-to-bottom.
constexpr int kChunk = 64;

template <bool DOWN>
__global__ void crop_col_gp_kernel(const u64* __restrict__ bg, const u64* __restrict__ out, int64_t h, int64_t W64,
                                   int64_t n_chunks, u64* __restrict__ G, u64* __restrict__ P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks * W64) return;
    const int64_t ch = i / W64, wd = i - ch * W64;
    const int64_t r_lo = ch * kChunk, r_hi = min(h, r_lo + kChunk);
    u64 g = 0, p = ~0ull;
    if (DOWN) {
        for (int64_t r = r_lo; r < r_hi; ++r) {
            const u64 b = bg[r * W64 + wd];
            g = out[r * W64 + wd] | (g & b);
            p &= b;
        }
    } else {
        for (int64_t r = r_hi - 1; r >= r_lo; --r) {
            const u64 b = bg[r * W64 + wd];
            g = out[r * W64 + wd] | (g & b);
            p &= b;
        }
    }
    G[i] = g;
    P[i] = p;
}

// carry INTO every chunk (overwrites G with it)
template <bool DOWN>
__global__ void crop_col_scan_kernel(u64* __restrict__ G, const u64* __restrict__ P, int64_t W64, int64_t n_chunks) {
    const int64_t wd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wd >= W64) return;
    u64 carry = 0;
    if (DOWN) {
        for (int64_t ch = 0; ch < n_chunks; ++ch) {
            const u64 g = G[ch * W64 + wd], p = P[ch * W64 + wd];
            G[ch * W64 + wd] = carry;
            carry = g | (p & carry);
        }
    } else {
        for (int64_t ch = n_chunks - 1; ch >= 0; --ch) {
            const u64 g = G[ch * W64 + wd], p = P[ch * W64 + wd];
            G[ch * W64 + wd] = carry;
            carry = g | (p & carry);
        }
    }
}

template <bool DOWN>
__global__ void crop_col_apply_kernel(const u64* __restrict__ bg, int64_t h, int64_t W64, int64_t n_chunks,
                                      const u64* __restrict__ Cin, u64* __restrict__ out, int* __restrict__ changed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks * W64) return;
    const int64_t ch = i / W64, wd = i - ch * W64;
    const int64_t r_lo = ch * kChunk, r_hi = min(h, r_lo + kChunk);
    u64 prev = Cin[i];
    bool chg = false;
    if (DOWN) {
        for (int64_t r = r_lo; r < r_hi; ++r) {
            const u64 old = out[r * W64 + wd];
            const u64 nw = old | (prev & bg[r * W64 + wd]);
            if (nw != old) {
                out[r * W64 + wd] = nw;
                chg = true;
            }
            prev = nw;
        }
    } else {
        for (int64_t r = r_hi - 1; r >= r_lo; --r) {
            const u64 old = out[r * W64 + wd];
            const u64 nw = old | (prev & bg[r * W64 + wd]);
            if (nw != old) {
                out[r * W64 + wd] = nw;
                chg = true;
            }
            prev = nw;
        }
    }
    if (chg) *changed = 1;
}

// height(line, k) of panoramaCropper.m:111-121: run length of inside pixels (BW2) ending at this line.  Same three
// steps as the column passes: per chunk of rows the run length at its end (and whether the whole chunk is inside),
// a serial scan over the chunks per column, then the chunk itself with its carried-in run length.
__global__ void crop_height_gp_kernel(const u64* __restrict__ out, int64_t h, int64_t w, int64_t W64, int64_t n_chunks,
                                      uint32_t* __restrict__ carry) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks * w) return;
    const int64_t ch = i / w, col = i - ch * w;
    const int64_t wd = col >> 6;
    const int bit = (int)(col & 63);
    const int64_t r_lo = ch * kChunk, r_hi = min(h, r_lo + kChunk);
    uint32_t cnt = 0;
    bool all_in = true;
    for (int64_t r = r_lo; r < r_hi; ++r) {
        const bool outside = (out[r * W64 + wd] >> bit) & 1ull;
        cnt = outside ? 0u : cnt + 1u;
        all_in = all_in && !outside;
    }
    carry[i] = cnt | (all_in ? 0x80000000u : 0u);
}

// run length carried INTO every chunk
__global__ void crop_height_scan_kernel(uint32_t* __restrict__ carry, int64_t w, int64_t n_chunks) {
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= w) return;
    uint32_t run = 0;
    for (int64_t ch = 0; ch < n_chunks; ++ch) {
        const uint32_t v = carry[ch * w + col];
        carry[ch * w + col] = run;
        run = (v & 0x80000000u) ? run + (v & 0x7fffffffu) : (v & 0x7fffffffu);
    }
}

__global__ void crop_heights_kernel(const u64* __restrict__ out, int64_t h, int64_t w, int64_t W64, int64_t n_chunks,
                                    const uint32_t* __restrict__ carry, uint16_t* __restrict__ H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_chunks * w) return;
    const int64_t ch = i / w, col = i - ch * w;
    const int64_t wd = col >> 6;
    const int bit = (int)(col & 63);
    const int64_t r_lo = ch * kChunk, r_hi = min(h, r_lo + kChunk);
    uint32_t cnt = carry[i];
    for (int64_t r = r_lo; r < r_hi; ++r) {
        const bool outside = (out[r * W64 + wd] >> bit) & 1ull;
        cnt = outside ? 0u : cnt + 1u;
        H[r * w + col] = (uint16_t)cnt;
    }
}

struct LineBest {
    u64 key;           // (area << 16) | (0xFFFF - k0): the maximum is the reference's first maximum of the line
    int32_t ll, rr, hh;  // 1-based left, right and the height of that element
    int32_t pad;
};

// panoramaCropper.m:123-149 for one line
__global__ __launch_bounds__(256) void crop_scan_kernel(const uint16_t* __restrict__ H, int64_t w, LineBest* __restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_h[];
    const int n = (int)w;
    const int n1 = (n + 31) >> 5, n2 = (n + 1023) >> 10;
    uint16_t* m1 = s_h + ((n + 7) & ~7);
    uint16_t* m2 = m1 + ((n1 + 7) & ~7);
    __shared__ u64 s_key[4];
    __shared__ u64 s_win;
    const int64_t line = blockIdx.x;
    const uint16_t* row = H + line * w;
    for (int k = threadIdx.x; k < n; k += 256) s_h[k] = k == n - 1 ? (uint16_t)0xFFFF : row[k];
    __syncthreads();
    // the last column never bounds or joins a rectangle: `right` is built for k = w-1..1 only and stops at w-1
    for (int b = threadIdx.x; b < n1; b += 256) {
        uint16_t m = 0xFFFF;
        for (int k = b * 32; k < min(n, b * 32 + 32); ++k) m = min(m, s_h[k]);
        m1[b] = m;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < n2; b += 256) {
        uint16_t m = 0xFFFF;
        for (int q = b * 32; q < min(n1, b * 32 + 32); ++q) m = min(m, m1[q]);
        m2[b] = m;
    }
    __syncthreads();
    u64 key = 0;
    int my_l = 0, my_r = 0, my_h = 0;
    for (int k = threadIdx.x; k < n - 1; k += 256) {
        const uint16_t v = s_h[k];
        if (v == 0) continue;
        // nearest j < k with s_h[j] < v
        int j = k - 1;
        while (j >= 0 && (j & 31) != 31 && s_h[j] >= v) --j;
        if (j >= 0 && (j & 31) == 31 && s_h[j] >= v) {
            int b = j >> 5;  // j is the last element of block b
            while (b >= 0 && m1[b] >= v) {
                if ((b & 31) == 31 && m2[b >> 5] >= v) b -= 32;
                else --b;
            }
            if (b < 0) {
                j = -1;
            } else {
                j = b * 32 + 31;
                while (s_h[j] >= v) --j;  // the block holds a smaller element
            }
        }
        const int left0 = j + 1;
        // nearest j > k with s_h[j] < v (the sentinel in column n-1 is never smaller)
        j = k + 1;
        while (j < n && (j & 31) != 0 && s_h[j] >= v) ++j;
        if (j < n && (j & 31) == 0 && s_h[j] >= v) {
            int b = j >> 5;  // j is the first element of block b
            while (b < n1 && m1[b] >= v) {
                if ((b & 31) == 0 && m2[b >> 5] >= v) b += 32;
                else ++b;
            }
            if (b >= n1) {
                j = n;
            } else {
                j = b * 32;
                while (s_h[j] >= v) ++j;
            }
        }
        const int right0 = min(j, n - 1) - 1;  // never beyond column w-1 (1-based), i.e. 0-based n-2
        const u64 area = (u64)(right0 - left0 + 1) * (u64)v;
        const u64 kk = (area << 16) | (u64)(0xFFFF - k);
        if (kk > key) {
            key = kk;
            my_l = left0 + 1;
            my_r = right0 + 1;
            my_h = v;
        }
    }
    u64 wk = key;
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u64 o = __shfl_xor(wk, s);
        wk = o > wk ? o : wk;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = wk;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = s_key[0];
        for (int q = 1; q < 4; ++q) m = s_key[q] > m ? s_key[q] : m;
        s_win = m;
        if (m == 0) best[line] = LineBest{0, 0, 0, 0, 0};
    }
    __syncthreads();
    if (key != 0 && key == s_win) best[line] = LineBest{key, my_l, my_r, my_h, 0};
}

// first maximum over the lines (panoramaCropper.m:139-147: strict `<`), then the crop indices (:153-157)
__global__ __launch_bounds__(256) void crop_pick_kernel(const LineBest* __restrict__ best, int64_t h, int64_t w,
                                                        int32_t* __restrict__ rect) {
    __shared__ u64 s_key[4];
    u64 key = 0;
    for (int64_t l = threadIdx.x; l < h; l += 256) {
        const u64 area = best[l].key >> 16;
        if (area == 0) continue;
        const u64 kk = (area << 16) | (u64)(0xFFFF - l);
        key = kk > key ? kk : key;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const u64 o = __shfl_xor(key, s);
        key = o > key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 m = s_key[0];
        for (int q = 1; q < 4; ++q) m = s_key[q] > m ? s_key[q] : m;
        int64_t ll = 0, rr = 0, hh = 0, nl = 0;
        if (m != 0) {
            const int64_t l = 0xFFFF - (int64_t)(m & 0xFFFF);
            ll = best[l].ll;
            rr = best[l].rr;
            hh = best[l].hh;
            nl = l + 1;
        }
        const int64_t cropH = hh + 1, cropW = rr - ll + 1, offsetx = ll, offsety = nl - hh + 1;
        rect[0] = (int32_t)offsetx;
        rect[1] = (int32_t)offsety;
        rect[2] = (int32_t)cropW;
        rect[3] = (int32_t)cropH;
        rect[4] = offsetx >= 1 && offsety >= 1 && offsety + cropH <= h && offsetx + cropW <= w;
    }
}

}  // namespace aps

using namespace aps;

extern "C" int aps_crop_rect(const uint8_t* img, int64_t h, int64_t w, int layout, int canvas_white, double range,
                             int32_t* rect, int32_t* valid) {
    return guarded([&] {
        APS_REQUIRE(img && rect && valid, APS_E_ARG, "NULL argument");
        APS_REQUIRE(h >= 1 && w >= 1, APS_E_DIM, "empty image");
        APS_REQUIRE(h <= 65534 && w <= 60000, APS_E_DIM,
                    "crop is built for canvases up to 65534 rows x 60000 columns (got %lld x %lld)", (long long)h, (long long)w);
        APS_REQUIRE(layout == APS_IMG_U8_HWC || layout == APS_IMG_U8_MATLAB, APS_E_TYPE, "unknown layout");
        APS_REQUIRE(canvas_white == 0 || canvas_white == 1, APS_E_ARG, "canvas_white must be 0 or 1");
        APS_REQUIRE(range >= 0.0 && range <= 255.0, APS_E_ARG, "range must be in [0,255]");
        ctx();
        const int64_t W64 = (w + 63) / 64;
        In<uint8_t> di(img, (size_t)h * w * 3);
        Ws<u64> bg((size_t)h * W64), out((size_t)h * W64);
        Ws<int> changed(1);
        const int64_t n_chunks = (h + kChunk - 1) / kChunk;
        Ws<u64> cG((size_t)n_chunks * W64), cP((size_t)n_chunks * W64);
        Ws<uint16_t> H((size_t)h * w);
        Ws<LineBest> best((size_t)h);
        Ws<int32_t> drect(5);
        const double t = (range / 255.0) * 255.0;
        {
            Prof prof("crop_mask");
            crop_mask_kernel<<<dim3(cdiv(W64, 4), (unsigned)h), 256, 0, stream()>>>(di, h, w, layout, canvas_white, t, W64, bg);
            crop_seed_kernel<<<cdiv((size_t)h * W64, 256), 256, 0, stream()>>>(bg, h, w, W64, out);
        }
        check_launch("crop_mask_kernel");
        {
            // rounds of (row pass, column pass) until one changes nothing; the count is the "winding depth" of the
            // background, 2-4 for a panorama outline, bounded by the pixel count for a maze
            Prof prof("crop_fill");
            const int64_t max_rounds = h * W64 * 64 + 2;
            for (int64_t it = 0; it < max_rounds; ++it) {
                APS_HIP(hipMemsetAsync(changed, 0, sizeof(int), stream()));
                crop_fill_rows_kernel<<<cdiv(h, 64), 64, 0, stream()>>>(bg, h, W64, out, changed);
                const unsigned gc = cdiv((size_t)n_chunks * W64, 256), gs = cdiv(W64, 64);
                crop_col_gp_kernel<true><<<gc, 256, 0, stream()>>>(bg, out, h, W64, n_chunks, cG, cP);
                crop_col_scan_kernel<true><<<gs, 64, 0, stream()>>>(cG, cP, W64, n_chunks);
                crop_col_apply_kernel<true><<<gc, 256, 0, stream()>>>(bg, h, W64, n_chunks, cG, out, changed);
                crop_col_gp_kernel<false><<<gc, 256, 0, stream()>>>(bg, out, h, W64, n_chunks, cG, cP);
                crop_col_scan_kernel<false><<<gs, 64, 0, stream()>>>(cG, cP, W64, n_chunks);
                crop_col_apply_kernel<false><<<gc, 256, 0, stream()>>>(bg, h, W64, n_chunks, cG, out, changed);
                int hc = 0;
                APS_HIP(hipMemcpyAsync(&hc, changed, sizeof(int), hipMemcpyDeviceToHost, stream()));
                APS_HIP(hipStreamSynchronize(stream()));
                if (!hc) break;
            }
        }
        check_launch("crop_fill_kernel");
        {
            Prof prof("crop_scan");
            Ws<uint32_t> hc((size_t)n_chunks * w);
            const unsigned gh = cdiv((size_t)n_chunks * w, 256);
            crop_height_gp_kernel<<<gh, 256, 0, stream()>>>(out, h, w, W64, n_chunks, hc);
            crop_height_scan_kernel<<<cdiv(w, 256), 256, 0, stream()>>>(hc, w, n_chunks);
            crop_heights_kernel<<<gh, 256, 0, stream()>>>(out, h, w, W64, n_chunks, hc, H);
            const int n1 = (int)((w + 31) >> 5), n2 = (int)((w + 1023) >> 10);
            const size_t lds = sizeof(uint16_t) * (size_t)(((w + 7) & ~7) + ((n1 + 7) & ~7) + ((n2 + 7) & ~7));
            static thread_local size_t attr_set = 0;
            if (lds > 48 * 1024 && lds > attr_set) {
                APS_HIP(hipFuncSetAttribute((const void*)crop_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr_set = lds;
            }
            crop_scan_kernel<<<(unsigned)h, 256, lds, stream()>>>(H, w, best);
            crop_pick_kernel<<<1, 256, 0, stream()>>>(best, h, w, drect);
        }
        check_launch("crop_scan_kernel");
        int32_t hr[5];
        APS_HIP(hipMemcpyAsync(hr, drect, sizeof hr, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        for (int e = 0; e < 4; ++e) rect[e] = hr[e];
        *valid = hr[4];
    });
}

// ------------------------------------------------------------------------------------------------
// cropNonzeroBbox (renderPanorama.m:1459-1504): bounding box of rgb2gray(pano) > 0 (black canvas) or < 255 (white)
// ------------------------------------------------------------------------------------------------
namespace aps {
// A fixed grid strides over the image; every thread keeps its own box in registers (interleaved RGB with an aligned
// base: four pixels per step as three dwords), the workgroup folds the boxes through shuffles and LDS, and only then
// touches the four global words: a few thousand same-address atomics per image instead of one quartet per patch
// (those serialise in L2: 3.8 M of them cost 40 ms on a 245 MPix canvas).
__global__ __launch_bounds__(256) void crop_bbox_kernel(const uint8_t* __restrict__ img, int64_t h, int64_t w, int layout,
                                                        int white, int* __restrict__ box) {
    __shared__ int s_box[4];  // rmin, rmax, cmin, cmax
    if (threadIdx.x < 4) s_box[threadIdx.x] = (threadIdx.x & 1) ? -1 : INT_MAX;
    __syncthreads();
    int rmin = INT_MAX, rmax = -1, cmin = INT_MAX, cmax = -1;
    const int64_t npx = h * w;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto see = [&](int64_t p, uint32_t r_, uint32_t g_, uint32_t b_) {
        const uint8_t g = crop_gray((uint8_t)r_, (uint8_t)g_, (uint8_t)b_);
        if (white ? g < 255 : g > 0) {
            const int r = (int)(p / w), c = (int)(p - (int64_t)r * w);
            rmin = min(rmin, r);
            rmax = max(rmax, r);
            cmin = min(cmin, c);
            cmax = max(cmax, c);
        }
    };
    if (layout == APS_IMG_U8_HWC && (reinterpret_cast<uintptr_t>(img) & 3u) == 0) {
        const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(img);
        const int64_t nq = npx / 4;
        for (int64_t q = t0; q < nq; q += stride) {
            const uint32_t a = s32[3 * q], b = s32[3 * q + 1], c = s32[3 * q + 2];
            if (!white && (a | b | c) == 0) continue;  // twelve zero bytes: four canvas pixels
            see(4 * q, a & 255u, (a >> 8) & 255u, (a >> 16) & 255u);
            see(4 * q + 1, a >> 24, b & 255u, (b >> 8) & 255u);
            see(4 * q + 2, (b >> 16) & 255u, b >> 24, c & 255u);
            see(4 * q + 3, (c >> 8) & 255u, (c >> 16) & 255u, c >> 24);
        }
        for (int64_t p = nq * 4 + t0; p < npx; p += stride) see(p, img[3 * p], img[3 * p + 1], img[3 * p + 2]);
    } else {
        for (int64_t p = t0; p < npx; p += stride) {
            if (layout == APS_IMG_U8_HWC) {
                see(p, img[3 * p], img[3 * p + 1], img[3 * p + 2]);
            } else {  // MATLAB h x w x 3 planes, column-major: walk memory order, (row, col) = (p % h, p / h)
                const int64_t c = p / h, r = p - c * h;
                see(r * w + c, img[p], img[npx + p], img[2 * npx + p]);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, off));
        rmax = max(rmax, __shfl_xor(rmax, off));
        cmin = min(cmin, __shfl_xor(cmin, off));
        cmax = max(cmax, __shfl_xor(cmax, off));
    }
    if ((threadIdx.x & 63) == 0 && rmax >= 0) {
        atomicMin(&s_box[0], rmin);
        atomicMax(&s_box[1], rmax);
        atomicMin(&s_box[2], cmin);
        atomicMax(&s_box[3], cmax);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_box[1] >= 0) {
        atomicMin(&box[0], s_box[0]);
        atomicMax(&box[1], s_box[1]);
        atomicMin(&box[2], s_box[2]);
        atomicMax(&box[3], s_box[3]);
    }
}
}  // namespace aps

extern "C" int aps_crop_nonzero_bbox(const uint8_t* img, int64_t h, int64_t w, int layout, int canvas_white, int64_t* rect,
                                     int* did_crop) {
    using namespace aps;
    return guarded([&] {
        APS_REQUIRE(img && rect && did_crop, APS_E_ARG, "NULL argument");
        APS_REQUIRE(h >= 1 && w >= 1 && h < INT_MAX && w < INT_MAX, APS_E_DIM, "bad image size");
        APS_REQUIRE(layout == APS_IMG_U8_HWC || layout == APS_IMG_U8_MATLAB, APS_E_TYPE, "unknown layout");
        ctx();
        In<uint8_t> di(img, (size_t)h * w * 3);
        Ws<int> box(4);
        const int init[4] = {INT_MAX, -1, INT_MAX, -1};
        APS_HIP(hipMemcpyAsync(box, init, sizeof init, hipMemcpyHostToDevice, stream()));
        {
            Prof prof("crop_bbox");
            crop_bbox_kernel<<<2048, 256, 0, stream()>>>(di, h, w, layout, canvas_white, box);
        }
        check_launch("crop_bbox_kernel");
        int hb[4];
        APS_HIP(hipMemcpyAsync(hb, box, sizeof hb, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
        if (hb[1] < 0) {  // no foreground: the whole image, didCrop = false (:1499-1502)
            rect[0] = 1;
            rect[1] = h;
            rect[2] = 1;
            rect[3] = w;
            *did_crop = 0;
            return;
        }
        const int64_t pad = 6;  // :1492
        rect[0] = std::max<int64_t>(1, (int64_t)hb[0] + 1 - pad);
        rect[1] = std::min<int64_t>(h, (int64_t)hb[1] + 1 + pad);
        rect[2] = std::max<int64_t>(1, (int64_t)hb[2] + 1 - pad);
        rect[3] = std::min<int64_t>(w, (int64_t)hb[3] + 1 + pad);
        *did_crop = 1;
    });
}
