// Probe: can VALU work overlap a dense v_mfma_f32_32x32x16_f16 stream on one SIMD?
//   mode 0: waves 0-3 MFMA only (waves 4-7 idle)        mode 1: waves 4-7 VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (cross-wave)  mode 3: every wave MFMA with VPM VALU ops between MFMAs
//   mode 4: both waves of a SIMD alternate phases of 16 MFMAs and 16*VPM VALU ops (the matching kernel's shape)
// build: hipcc --offload-arch=gfx950 -O3 -DVPM=6 -o scripts/probe/mfma_valu_coexec.bin scripts/probe/mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef VPM
#define VPM 6
#endif

#define VALU_OPS(n)                                                         \
    _Pragma("unroll") for (int q = 0; q < (n); ++q) {                       \
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(y), "v"(z)); \
    }

__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)(0.001f * (threadIdx.x + e));
        b[e] = (_Float16)(0.002f * (threadIdx.x - e));
    }
    f32x16 acc0 = {0}, acc1 = {0};
    float x[8], y = threadIdx.x * 0.5f, z = threadIdx.x * 0.25f;
    for (int q = 0; q < 8; ++q) x[q] = q;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    const bool do_mfma = mode == 0 ? wave < 4 : mode == 1 ? false : mode == 2 ? wave < 4 : true;
    const bool do_valu = mode == 0 ? false : mode == 1 ? wave >= 4 : mode == 2 ? wave >= 4 : true;
    if (mode <= 2) {
        if (do_mfma)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                }
            }
        if (do_valu)
            for (int i = 0; i < iters; ++i) { VALU_OPS(16 * VPM) }
    } else if (mode == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                VALU_OPS(VPM)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                VALU_OPS(VPM)
            }
        }
    } else {
        if (wave >= 4) { VALU_OPS(16 * VPM) }
        for (int i = 0; i < iters; ++i) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            VALU_OPS(16 * VPM)
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    for (int q = 0; q < 8; ++q) s += x[q];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 7 && (threadIdx.x == 0 || threadIdx.x == 256))
        printf("mode %d wave %d: %.1f cycles per 16-MFMA block-equivalent (%.2f GHz)\n", mode, wave,
               (double)(c1 - c0) / iters, (double)(c1 - c0) / ((double)(w1 - w0) * 10.0));
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    printf("VALU ops per MFMA: %d (16 MFMAs = 512 cycles of pipe per wave; %d VALU = %d issue cycles per wave)\n", VPM, 16 * VPM, 64 * VPM);
    for (int mode = 0; mode <= 4; ++mode) {
        k<<<256, 512>>>(out, 20000, mode);
        hipDeviceSynchronize();
    }
    return 0;
}
