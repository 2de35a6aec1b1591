// Microbenchmark: HBM read rate of the access patterns the SIFT kernels can choose from, on NP planes of W x H floats
// (octave 0 of a 4K view: 7680 x 4320, 133 MB per plane, 7 planes):
//   linear      every thread a float4 at the same linear offset of each plane (the copy-kernel pattern)
//   tile TWxTH  one 256-thread workgroup per TW x TH tile plus a 1-pixel halo rounded to float4 (extrema_kernel's fill)
//   march SWxCH one workgroup per SW-wide strip segment of CH rows, marching down RS rows per step with the next step's
//               loads requested before the current step is consumed (every row read once, + 2 halo rows per segment)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_pattern scripts/probe/mem_pattern.hip ; run: /tmp/mem_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NP = 7;
struct Planes { const float* p[NP]; };

__device__ __forceinline__ float sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

__global__ __launch_bounds__(256) void k_linear(Planes P, size_t n4, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 g[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) g[p] = reinterpret_cast<const float4*>(P.p[p])[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) acc += sum4(g[p]);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int TW, int TH>
__global__ __launch_bounds__(256) void k_tile(Planes P, int w, int h, float* out) {
    constexpr int NV = (TW + 8) / 4, IH = TH + 2;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    float acc = 0.f;
    for (int e = threadIdx.x; e < IH * NV; e += 256) {
        const int ly = e / NV, v = e - ly * NV;
        const int gy = min(max(y0 + ly - 1, 0), h - 1), gx = min(max(x0 - 4 + 4 * v, 0), w - 4);
        const size_t off = (size_t)gy * w + gx;
        float4 g[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) g[p] = *reinterpret_cast<const float4*>(P.p[p] + off);
#pragma unroll
        for (int p = 0; p < NP; ++p) acc += sum4(g[p]);
    }
    if (acc == 123.456f) out[0] = acc;
}

// strip march: SW columns (SW/4 float4 per row; 256 threads cover 256*4/SW rows per step)
template <int SW, int CH>
__global__ __launch_bounds__(256) void k_march(Planes P, int w, int h, float* out) {
    constexpr int LPR = SW / 4, RS = 256 / LPR;  // lanes per row, rows per step
    static_assert(256 % LPR == 0 && CH % RS == 0, "shape");
    const int x0 = blockIdx.x * SW, y0 = blockIdx.y * CH;
    const int lx = threadIdx.x % LPR, ly = threadIdx.x / LPR;
    const int gx = min(x0 + 4 * lx, w - 4);
    float acc = 0.f;
    float4 cur[NP], nxt[NP];
    auto issue = [&](int row, float4* dst) {
        const int gy = min(max(row, 0), h - 1);
        const size_t off = (size_t)gy * w + gx;
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[p] = *reinterpret_cast<const float4*>(P.p[p] + off);
    };
    issue(y0 - 1 + ly, cur);  // (first step includes the halo row above)
    for (int r = RS; r < CH + 2 + RS; r += RS) {
        if (r < CH + 2) issue(y0 - 1 + r + ly, nxt);
#pragma unroll
        for (int p = 0; p < NP; ++p) acc += sum4(cur[p]);
#pragma unroll
        for (int p = 0; p < NP; ++p) cur[p] = nxt[p];
    }
    if (acc == 123.456f) out[0] = acc;
}

// write-side: linear float4 stores to one plane vs tile-shaped stores
__global__ __launch_bounds__(256) void k_wlinear(float* o, size_t n4) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        reinterpret_cast<float4*>(o)[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_wtile(float* o, int w, int h) {
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    for (int e = threadIdx.x; e < TH * (TW / 4); e += 256) {
        const int ly = e / (TW / 4), v = e - ly * (TW / 4);
        const int gy = y0 + ly, gx = x0 + 4 * v;
        if (gy < h && gx + 3 < w) *reinterpret_cast<float4*>(o + (size_t)gy * w + gx) = make_float4(1.f, 2.f, 3.f, (float)e);
    }
}

template <class F>
static float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int w = 7680, h = 4320;
    const size_t n = (size_t)w * h;
    Planes P;
    std::vector<float*> bufs;
    for (int p = 0; p < NP; ++p) {
        float* d;
        CK(hipMalloc(&d, n * 4));
        CK(hipMemset(d, 0, n * 4));
        bufs.push_back(d);
        P.p[p] = d;
    }
    float *out, *wr;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&wr, n * 4));
    const double gb = (double)NP * n * 4 / 1e9;
    auto report = [&](const char* name, float ms, double bytes_gb) {
        printf("%-28s %8.3f ms  %7.2f TB/s (useful bytes %.2f GB)\n", name, ms, bytes_gb / ms, bytes_gb);
    };
    report("linear 7 planes", time_ms([&] { k_linear<<<256 * 16, 256>>>(P, n / 4, out); }), gb);
    report("tile 128x8 (+halo)", time_ms([&] { k_tile<128, 8><<<dim3(w / 128, h / 8), 256>>>(P, w, h, out); }), gb);
    report("tile 256x8 (+halo)", time_ms([&] { k_tile<256, 8><<<dim3(w / 256, h / 8), 256>>>(P, w, h, out); }), gb);
    report("tile 512x4 (+halo)", time_ms([&] { k_tile<512, 4><<<dim3(w / 512, h / 4), 256>>>(P, w, h, out); }), gb);
    report("tile 128x16 (+halo)", time_ms([&] { k_tile<128, 16><<<dim3(w / 128, h / 16), 256>>>(P, w, h, out); }), gb);
    report("tile 256x16 (+halo)", time_ms([&] { k_tile<256, 16><<<dim3(w / 256, h / 16), 256>>>(P, w, h, out); }), gb);
    report("march 256 x 72", time_ms([&] { k_march<256, 72><<<dim3(w / 256, h / 72), 256>>>(P, w, h, out); }), gb);
    report("march 256 x 144", time_ms([&] { k_march<256, 144><<<dim3(w / 256, h / 144), 256>>>(P, w, h, out); }), gb);
    report("march 512 x 72", time_ms([&] { k_march<512, 72><<<dim3(w / 512, h / 72), 256>>>(P, w, h, out); }), gb);
    report("march 512 x 144", time_ms([&] { k_march<512, 144><<<dim3(w / 512, h / 144), 256>>>(P, w, h, out); }), gb);
    report("march 1024 x 72", time_ms([&] { k_march<1024, 72><<<dim3((w + 1023) / 1024, h / 72), 256>>>(P, w, h, out); }), gb);
    const double gw = (double)n * 4 / 1e9;
    report("write linear 1 plane", time_ms([&] { k_wlinear<<<256 * 16, 256>>>(wr, n / 4); }), gw);
    report("write tile 64x32", time_ms([&] { k_wtile<64, 32><<<dim3(w / 64, h / 32), 256>>>(wr, w, h); }), gw);
    report("write tile 128x8", time_ms([&] { k_wtile<128, 8><<<dim3(w / 128, h / 8), 256>>>(wr, w, h); }), gw);
    report("write tile 256x8", time_ms([&] { k_wtile<256, 8><<<dim3(w / 256, h / 8), 256>>>(wr, w, h); }), gw);
    return 0;
}
