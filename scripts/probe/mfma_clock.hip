// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 issue rate and the shader clock under that load.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock scripts/probe/mfma_clock.hip ; run: /tmp/mfma_clock [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// variant: 32 resident operand vectors (128 VGPRs) cycled as in the matching kernel, streamed operand constant
__global__ __launch_bounds__(512) void k2(float* out, unsigned long long* stamps, int iters) {
    bf16x8 ah[2][8], al[2][8], b, b2;
    for (int rb = 0; rb < 2; ++rb)
        for (int s = 0; s < 8; ++s)
            for (int e = 0; e < 8; ++e) {
                ah[rb][s][e] = (__bf16)(0.001f * (threadIdx.x + e + s + rb));
                al[rb][s][e] = (__bf16)(0.0001f * (threadIdx.x + e - s + rb));
            }
    for (int e = 0; e < 8; ++e) {
        b[e] = (__bf16)(0.002f * (threadIdx.x - e));
        b2[e] = (__bf16)(0.003f * (threadIdx.x - e));
    }
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, ah[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, ah[1][s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, al[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, al[1][s], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, ah[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, ah[1][s], acc1, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(0.001f * (threadIdx.x + e));
        b[e] = (__bf16)(0.002f * (threadIdx.x - e));
    }
    f32x16 acc0 = {0}, acc1 = {0};
    const unsigned long long c0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = w1 - w0;
    }
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;  // waves per SIMD
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int nwg = 256, threads = 64 * 4 * wps;
    float* out;
    unsigned long long* st;
    hipMalloc(&out, nwg * threads * sizeof(float));
    hipMalloc(&st, nwg * 2 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (argc > 3) k2<<<nwg, threads>>>(out, st, iters); else k<<<nwg, threads>>>(out, st, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
        const double mfma_per_simd = (double)iters * 48 * wps;
        const double flops = (double)nwg * 4 * mfma_per_simd * 32768.0;
        printf("waves/SIMD %d: %.3f ms  %.0f TFLOP/s  shader cycles/MFMA/SIMD %.2f  clock %.3f GHz (cycles %llu / wall %llu ticks @100MHz)\n",
               wps, ms, flops / ms / 1e9, (double)h[0] / mfma_per_simd, (double)h[0] / ((double)h[1] * 10.0), h[0], h[1]);
    }
    return 0;
}
