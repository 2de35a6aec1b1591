// Debug target only (`make debug`): the co-runner of scripts/probe/probe_overlap_race3.py with its int8 MFMA accumulators
// in AGPRs.  This file alone is compiled with -amdgpu-mfma-vgpr-form=0 (Makefile), the rest of the library with =1:
// DESIGN.md section 5 ("Co-residency finding") - SIFT kernels sharing a SIMD with VGPR-form v_mfma_i32_32x32x32_i8
// waves returned different bits; is it the VGPR form?
#include <hip/hip_runtime.h>

namespace aps {
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void dbg_corun_agpr_kernel(int mode, int spin, int* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[96 * 1024];  // the same LDS footprint as dbg_corun_kernel
    const int lane = threadIdx.x & 63;
    i32x16 c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    i32x4 c4 = {0, 0, 0, 0};
    const i32x4 a = {lane, 1, 2, 3};
    for (int it = 0; it < spin; ++it) {
        if (mode & 128) {
            // ("a": the accumulator is pinned to AGPRs - left alone the compiler kept it in VGPRs even with vgpr-form=0)
            asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %1, %0\n\tv_mfma_i32_32x32x32_i8 %0, %1, %1, %0" : "+a"(c) : "v"(a));
        }
        if (mode & 256) {
            asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %1, %0\n\tv_mfma_i32_16x16x64_i8 %0, %1, %1, %0\n\t"
                         "v_mfma_i32_16x16x64_i8 %0, %1, %1, %0\n\tv_mfma_i32_16x16x64_i8 %0, %1, %1, %0" : "+a"(c4) : "v"(a));
        }
    }
    if (c[0] + c4[0] == 0x7fffffff) sink[0] = lds[threadIdx.x];
}
void dbg_corun_agpr_launch(int mode, int n_wg, int spin, int* sink, hipStream_t st) {
    dbg_corun_agpr_kernel<<<n_wg, 512, 0, st>>>(mode, spin, sink);
}
}  // namespace aps

