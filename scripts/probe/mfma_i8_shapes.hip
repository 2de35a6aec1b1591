// Microbenchmark: v_mfma_i32_32x32x32_i8 against v_mfma_i32_16x16x64_i8 under a sustained dense stream on random int8
// operands - the same multiply-adds per cycle on paper; which one does the (power-limited) chip run faster?
// (MI355X_MICROARCH.md, DVFS give-back item 7: for bf16 the 16x16x32 shape delivered ~1.15x the 32x32x16 shape.)
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_i8_shapes scripts/probe/mfma_i8_shapes.hip
// run:   /tmp/mfma_i8_shapes [waves_per_simd] [iters] [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <bool WIDE>
__global__ __launch_bounds__(512) void k(int* out, unsigned long long* stamps, int iters, int zero) {
    i32x4 a[8], b[4];
    for (int s = 0; s < 8; ++s)
        for (int e = 0; e < 4; ++e) a[s][e] = zero ? 0 : (int)mix(threadIdx.x * 977u + s * 31u + e + blockIdx.x * 7919u);
    for (int s = 0; s < 4; ++s)
        for (int e = 0; e < 4; ++e) b[s][e] = zero ? 0 : (int)mix(threadIdx.x * 613u + s * 17u + e + 12345u);
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    int sum = 0;
    if (WIDE) {
        i32x16 acc0 = {0}, acc1 = {0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {  // 8 MFMAs = one 32-column block of the screen (2 row blocks x 4 k-steps)
                    acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[(s + u) & 3], a[s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b[(s + u) & 3], a[4 + s], acc1, 0, 0, 0);
                }
            }
        }
        for (int e = 0; e < 16; ++e) sum += acc0[e] + acc1[e];
    } else {
        i32x4 acc[4] = {{0}, {0}, {0}, {0}};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {  // the same multiply-adds: 16 MFMAs of half the size
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[g] = __builtin_amdgcn_mfma_i32_16x16x64_i8(b[(s + u + g) & 3], a[(2 * g + s) & 7], acc[g], 0, 0, 0);
                }
            }
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

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 40000;
    const int zero = argc > 3 ? atoi(argv[3]) : 0;
    const int nwg = 256, threads = 64 * 4 * wps;
    int* out;
    unsigned long long* st;
    hipMalloc(&out, nwg * threads * sizeof(int));
    hipMalloc(&st, nwg * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int shape = 0; shape < 2; ++shape)
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (shape == 0) k<true><<<nwg, threads>>>(out, st, iters, zero); else k<false><<<nwg, threads>>>(out, st, iters, zero);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[2];
            hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
            // per wave and iteration: 32 x (32x32x32) or 64 x (16x16x64) MFMAs = 32 * 32768 multiply-adds
            const double macs = (double)nwg * 4 * wps * iters * 32.0 * 32768.0;
            printf("%s waves/SIMD %d%s: %.3f ms  %.0f TOP/s  clock %.3f GHz\n", shape == 0 ? "32x32x32" : "16x16x64", wps,
                   zero ? " (zero operands)" : "", ms, 2.0 * macs / ms / 1e9, (double)h[0] / ((double)h[1] * 10.0));
        }
    return 0;
}
