// checks IEEE-correct rounding of f32 sqrt/div/add/mul intrinsics on gfx950 against the host
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const float* a, const float* b, float* s, float* d, float* d2, float* s2, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = __fsqrt_rn(a[i]);
    d[i] = __fdiv_rn(a[i], b[i]);
    d2[i] = a[i] / b[i];
    s2[i] = sqrtf(a[i]);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), s(n), d(n), d2(n), s2(n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = (float)rand() / RAND_MAX * 300.f; b[i] = (float)rand() / RAND_MAX * 600.f + 0.001f; }
    float *da, *db, *ds, *dd, *dd2, *ds2;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dd2, n * 4); hipMalloc(&ds2, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, ds, dd, dd2, ds2, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(d2.data(), dd2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(s2.data(), ds2, n * 4, hipMemcpyDeviceToHost);
    long bs = 0, bd = 0, bd2 = 0, bs2 = 0;
    for (int i = 0; i < n; ++i) {
        volatile float hs = sqrtf(a[i]); volatile float hd = a[i] / b[i];
        bs += s[i] != hs; bd += d[i] != hd; bd2 += d2[i] != hd; bs2 += s2[i] != hs;
    }
    printf("mismatch: __fsqrt_rn %ld  __fdiv_rn %ld  operator/ %ld  sqrtf %ld of %d\n", bs, bd, bd2, bs2, n);
    return 0;
}
