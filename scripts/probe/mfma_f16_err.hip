// Probe: accumulation error of a chain of v_mfma_f32_32x32x16_f16 (K = 144, as in the matching kernel) against f64,
// and whether f16 subnormal inputs are honoured.  The split-precision matcher's certification bound assumes
// |acc - exact| <= 2^-15 * sum|a_k b_k|; this prints the measured maximum in units of 2^-24 * sum|a_k b_k|.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_err scripts/probe/mfma_f16_err.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int K = 144;

__global__ void chain(const _Float16* A, const _Float16* B, float* C) {
    const int l = threadIdx.x, c = l & 31, h = l >> 5;
    f32x16 acc = {0};
    for (int s = 0; s < K / 16; ++s) {
        const f16x8 a = *reinterpret_cast<const f16x8*>(A + c * K + 16 * s + 8 * h);
        const f16x8 b = *reinterpret_cast<const f16x8*>(B + c * K + 16 * s + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c] = acc[r];
}

int main() {
    std::vector<_Float16> A(32 * K), B(32 * K);
    _Float16 *dA, *dB;
    float* dC;
    hipMalloc(&dA, A.size() * 2);
    hipMalloc(&dB, B.size() * 2);
    hipMalloc(&dC, 32 * 32 * 4);
    std::vector<float> C(32 * 32);
    double worst = 0, worst_signed = 0;
    srand(1);
    for (int trial = 0; trial < 2000; ++trial) {
        const int mode = trial % 4;
        for (size_t i = 0; i < A.size(); ++i) {
            float u = rand() / (float)RAND_MAX, v = rand() / (float)RAND_MAX;
            float x = u * u * u * 0.5f, y = v * v * v * 0.5f;
            if (mode == 1) { x *= 200.f; y *= 200.f; }
            if (mode == 2) { x = (u - 0.5f); y = (v - 0.5f); }
            if (mode == 3) { x = ldexpf(u, -(rand() % 12)); y = ldexpf(v, -(rand() % 12)); }
            A[i] = (_Float16)x;
            B[i] = (_Float16)y;
        }
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        chain<<<1, 64>>>(dA, dB, dC);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ex = 0, ab = 0;
                for (int k = 0; k < K; ++k) {
                    const double p = (double)(float)A[i * K + k] * (double)(float)B[j * K + k];
                    ex += p;
                    ab += fabs(p);
                }
                const double e = (C[i * 32 + j] - ex) / (ab * ldexp(1.0, -24));
                if (fabs(e) > worst) worst = fabs(e);
                if (fabs(e) > fabs(worst_signed)) worst_signed = e;
            }
    }
    printf("max |acc - exact| = %.2f x 2^-24 x sum|a b|  (signed %.2f; assumed bound 512)\n", worst, worst_signed);
    // subnormal inputs: a = 2^-20 (f16 subnormal), b = 2^10 -> 2^-10 if honoured, 0 if flushed
    for (size_t i = 0; i < A.size(); ++i) A[i] = B[i] = (_Float16)0.f;
    A[0] = (_Float16)ldexpf(1.f, -20);
    B[0] = (_Float16)1024.f;
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    chain<<<1, 64>>>(dA, dB, dC);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    printf("subnormal f16 input: product = %g (expected %g if honoured)\n", C[0], ldexp(1.0, -10));
    return 0;
}
