// Round-5 probe: does a block-scaled FP6 / FP4 screening pass pay?
//   (1) sustained rate of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 / e2m3 (fp6) / e2m1 (fp4) operands against
//       v_mfma_i32_16x16x64_i8 under a dense stream on random operands (bare loops, operands in registers, 256 workgroups,
//       two waves per SIMD) - rate = cycles per instruction x the clock the chip holds under that stream;
//   (2) is an e2m3 dot product with unit block scales EXACT?  Every e2m3 value is a multiple of 1/8 of magnitude <= 7.5, a
//       product a multiple of 1/64 <= 56.25, a sum of 128 products an integer multiple of 1/64 below 2^13: exactly
//       representable in f32 - if the pipe adds without intermediate rounding below f32.  Checked on random 6-bit fields
//       against a host evaluation (which also checks the operand layout assumed: lane l = row / column l & 15, elements
//       k = 32 (l >> 4) + j, j-th 6-bit field of the lane's 192 bits, little endian).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f8f6f4 scripts/probe/mfma_f8f6f4.hip
// run:   /tmp/mfma_f8f6f4 [iters]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned mix(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// FMT: -1 = int8 16x16x64 (two instructions per 128-deep step), 0 = e4m3, 2 = e2m3, 4 = e2m1 through the scaled instruction
template <int FMT>
__global__ __launch_bounds__(512) void rate_kernel(float* out, unsigned long long* stamps, int iters, int zero) {
    i32x8 a[4], b[4];
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 8; ++e) {
            unsigned va = zero ? 0u : mix(threadIdx.x * 977u + s * 31u + e + blockIdx.x * 7919u);
            unsigned vb = zero ? 0u : mix(threadIdx.x * 613u + s * 17u + e + 12345u);
            if (FMT == 0) {  // no e4m3 NaN patterns (S.1111.111)
                va &= 0x7e7e7e7eu;
                vb &= 0x7e7e7e7eu;
            }
            a[s][e] = (int)va;
            b[s][e] = (int)vb;
        }
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    float sum = 0.f;
    if (FMT < 0) {
        i32x4 acc[4] = {{0}, {0}, {0}, {0}};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // one 16 x 16 x 128 step = two 16x16x64 instructions
                    const i32x4 a0 = {a[g][0], a[g][1], a[g][2], a[g][3]}, a1 = {a[g][4], a[g][5], a[g][6], a[g][7]};
                    const i32x4 b0 = {b[(g + u) & 3][0], b[(g + u) & 3][1], b[(g + u) & 3][2], b[(g + u) & 3][3]};
                    const i32x4 b1 = {b[(g + u) & 3][4], b[(g + u) & 3][5], b[(g + u) & 3][6], b[(g + u) & 3][7]};
                    acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc[g], 0, 0, 0);
                    acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc[g], 0, 0, 0);
                }
        }
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 4; ++e) sum += (float)acc[g][e];
    } else {
        f32x4 acc[4] = {{0}, {0}, {0}, {0}};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    acc[g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[g], b[(g + u) & 3], acc[g], FMT, FMT, 0, 127, 0, 127);
        }
        for (int g = 0; g < 4; ++g)
            for (int e = 0; e < 4; ++e) sum += acc[g][e];
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

// one 16 x 16 x 128 product of random e2m3 fields, unit scales, C = 0
__global__ void exact_kernel(const unsigned* A, const unsigned* B, float* C) {
    const int l = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = 0; e < 6; ++e) {
        a[e] = (int)A[l * 6 + e];
        b[e] = (int)B[l * 6 + e];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2, 2, 0, 127, 0, 127);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];  // C/D map: col = lane & 15, row = 4 (lane >> 4) + reg
}

static double e2m3(unsigned f) {
    const int s = (f >> 5) & 1, e = (f >> 3) & 3, m = f & 7;
    const double v = e == 0 ? m / 8.0 : (1.0 + m / 8.0) * (double)(1 << (e - 1));
    return s ? -v : v;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int nwg = 256, threads = 512;
    float* out;
    unsigned long long* st;
    hipMalloc(&out, nwg * threads * sizeof(float));
    hipMalloc(&st, nwg * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* names[4] = {"i8   16x16x64 x2", "e4m3 16x16x128  ", "e2m3 16x16x128  ", "e2m1 16x16x128  "};
    for (int zero = 0; zero < 2; ++zero)
        for (int v = 0; v < 4; ++v)
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (v) {
                    case 0: rate_kernel<-1><<<nwg, threads>>>(out, st, iters, zero); break;
                    case 1: rate_kernel<0><<<nwg, threads>>>(out, st, iters, zero); break;
                    case 2: rate_kernel<2><<<nwg, threads>>>(out, st, iters, zero); break;
                    default: rate_kernel<4><<<nwg, threads>>>(out, st, iters, zero); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long h[2];
                hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
                // per wave and iteration: 16 steps of 16 x 16 x 128 = 16 * 32768 multiply-adds
                const double macs = (double)nwg * 8 * iters * 16.0 * 32768.0;
                if (rep == 2)
                    printf("%s%s: %8.3f ms  %6.0f TOP/s  clock %.3f GHz  cycles per 16x16x128 step and SIMD %.1f\n", names[v],
                           zero ? " (zero operands)" : "                ", ms, 2.0 * macs / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0),
                           (double)h[0] / ((double)iters * 16.0 * 2.0));
            }
    // exactness
    std::vector<unsigned> A(64 * 6), B(64 * 6);
    for (size_t i = 0; i < A.size(); ++i) {
        A[i] = mix(1000u + (unsigned)i);
        B[i] = mix(5000u + (unsigned)i);
    }
    unsigned *dA, *dB;
    float* dC;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    exact_kernel<<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(256);
    hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    auto field = [](const std::vector<unsigned>& X, int lane, int j) {
        const int bit = 6 * j, w = bit >> 5, o = bit & 31;
        unsigned long long two = X[lane * 6 + w];
        if (w + 1 < 6) two |= (unsigned long long)X[lane * 6 + w + 1] << 32;
        return (unsigned)((two >> o) & 63u);
    };
    double worst = 0.0, biggest = 0.0;
    int inexact = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0.0;
            for (int kb = 0; kb < 4; ++kb)
                for (int e = 0; e < 32; ++e) s += e2m3(field(A, 16 * kb + i, e)) * e2m3(field(B, 16 * kb + j, e));
            const double d = std::fabs((double)C[i * 16 + j] - s);
            worst = d > worst ? d : worst;
            biggest = std::fabs(s) > biggest ? std::fabs(s) : biggest;
            inexact += d != 0.0;
        }
    printf("e2m3 16x16x128, unit scales, random fields: %d of 256 results differ from the exact dot product, largest difference %.6g "
           "(largest |dot| %.4g)\n", inexact, worst, biggest);
    return 0;
}
