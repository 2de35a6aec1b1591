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

template <int mode>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
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
    if constexpr (mode <= 2) {
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
    } else if constexpr (mode == 3) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                VALU_OPS(VPM)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                VALU_OPS(VPM)
            }
        }
    } else if constexpr (mode == 15 || mode == 16) {
        // mode 13 with the matching kernel's LDS image: 128 rows x 256 B per tile, 16-byte chunk position XOR-swizzled
        // with (row & 15), three tile buffers; mode 16 adds the LDS-DMA refill (one 1-KiB piece per wave and block)
        // and the hand-over (vmcnt(0) + barrier) every four blocks
        __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * 32768];
        for (int q = threadIdx.x; q < 3 * 32768 / 16; q += 512) reinterpret_cast<float4*>(lds)[q] = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
        __syncthreads();
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int c = threadIdx.x & 31, h = (threadIdx.x >> 5) & 1;
        const int hx = (16 * h) ^ (16 * (c & 15));
        float thr = 1e30f;
        f16x8 ares[2][8];
        for (int r = 0; r < 2; ++r)
            for (int q = 0; q < 8; ++q)
                for (int e = 0; e < 8; ++e) ares[r][q][e] = (_Float16)(0.001f * (threadIdx.x + e + 3 * q + 7 * r));
        f16x8 ring[4];
        for (int s = 0; s < 3; ++s) ring[s] = *reinterpret_cast<const f16x8*>(lds + c * 256 + ((32 * s) ^ hx));
        f32x16 accs[2][2] = {{acc0, acc1}, {acc0, acc1}};
        const f32x16 z16 = {0};
        float msc = 0.f;
        const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
        const unsigned short* gsrc = reinterpret_cast<const unsigned short*>(out) + (threadIdx.x & 63) * 8;
        int buf = 0;
        for (int i = 0; i < iters; i += 4) {
            const int nbuf = buf == 2 ? 0 : buf + 1;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                constexpr int dummy = 0;
                (void)dummy;
                const int par = cb & 1;
                if constexpr (mode == 16) {
                    const unsigned short* src = gsrc + (size_t)((i + cb) & 255) * 4096 + wv * 512;
                    const uint32_t dst = lds_base + (nbuf == 2 ? 0 : nbuf + 1) * 32768 + (wv * 4 + cb) * 1024;
                    uint32_t keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
                }
                const unsigned char* blk = lds + buf * 32768 + cb * 8192 + c * 256;
                const unsigned char* nblk = cb < 3 ? blk + 8192 : lds + nbuf * 32768 + c * 256;
                accs[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[3], ares[0][0], z16, 0, 0, 0);
                accs[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[3], ares[1][1], z16, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const f16x8 bv = ring[s & 3];
                    if (s + 3 < 8) ring[(s + 3) & 3] = *reinterpret_cast<const f16x8*>(blk + ((32 * (s + 3)) ^ hx));
                    else ring[(s + 3) & 3] = *reinterpret_cast<const f16x8*>(nblk + ((32 * (s + 3 - 8)) ^ hx));
                    accs[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[0][s], accs[par][0], 0, 0, 0);
                    accs[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[1][s], accs[par][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x16& pa = accs[par ^ 1][s & 1];
                    const int g = s >> 1;
                    unsigned long long anyh;
                    asm("v_max3_f32 %1, %2, %3, %4\n\tv_max_f32 %1, %1, %5\n\tv_cmp_gt_f32 %0, %1, %6"
                        : "=s"(anyh), "+v"(msc)
                        : "v"(pa[4 * g]), "v"(pa[4 * g + 1]), "v"(pa[4 * g + 2]), "v"(pa[4 * g + 3]), "v"(thr));
                    if (__builtin_expect(anyh != 0, 0)) {
                        x[s & 7] += 1.0f;
                        thr *= 2.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (cb == 2) {
                    if constexpr (mode == 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
            }
            buf = nbuf;
        }
        acc0 = accs[0][0] + accs[1][0];
        acc1 = accs[0][1] + accs[1][1];
        x[0] += msc;
    } else if constexpr (mode == 13 || mode == 14) {
        const int tl = mode == 14 ? (threadIdx.x & 63) : threadIdx.x;  // mode 14: all eight waves read the SAME addresses
        // the matching kernel's block: two accumulator sets; block i multiplies into set i&1 while the slices search
        // set (i&1)^1 (registers written by the previous block's MFMAs)
        __shared__ float4 sh[1024];
        sh[threadIdx.x] = make_float4(1, 2, 3, 4);
        sh[threadIdx.x + 512] = make_float4(1, 2, 3, 4);
        __syncthreads();
        float thr = 1e30f;
        f16x8 ares[2][8];
        for (int r = 0; r < 2; ++r)
            for (int q = 0; q < 8; ++q)
                for (int e = 0; e < 8; ++e) ares[r][q][e] = (_Float16)(0.001f * (threadIdx.x + e + 3 * q + 7 * r));
        float4 ring[4];
        for (int s = 0; s < 3; ++s) ring[s] = sh[(tl + 64 * s) & 1023];
        f32x16 accs[2][2] = {{acc0, acc1}, {acc0, acc1}};
        const f32x16 z16 = {0};
        float msc = 0.f;
        for (int i = 0; i < iters; i += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                f16x8 bv0;
                __builtin_memcpy(&bv0, &ring[3], 16);
                accs[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv0, ares[0][0], z16, 0, 0, 0);
                accs[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv0, ares[1][1], z16, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float4 v = ring[s & 3];
                    ring[(s + 3) & 3] = sh[(tl + 64 * (s + 3) + i) & 1023];
                    f16x8 bv;
                    __builtin_memcpy(&bv, &v, 16);
                    accs[par][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[0][s], accs[par][0], 0, 0, 0);
                    accs[par][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[1][s], accs[par][1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x16& pa = accs[par ^ 1][s & 1];
                    const int g = s >> 1;
                    unsigned long long anyh;
                    asm("v_max3_f32 %1, %2, %3, %4\n\tv_max_f32 %1, %1, %5\n\tv_cmp_gt_f32 %0, %1, %6"
                        : "=s"(anyh), "+v"(msc)
                        : "v"(pa[4 * g]), "v"(pa[4 * g + 1]), "v"(pa[4 * g + 2]), "v"(pa[4 * g + 3]), "v"(thr));
                    if (__builtin_expect(anyh != 0, 0)) {
                        x[s & 7] += 1.0f;
                        thr *= 2.f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if ((i & 2) == 2) __syncthreads();
        }
        acc0 = accs[0][0] + accs[1][0];
        acc1 = accs[0][1] + accs[1][1];
        x[0] += msc;
    } else if constexpr (mode >= 5 && mode <= 12) {
        // the matching kernel's slice: two MFMAs, one 16-byte LDS read, max of four, compare, (never taken) branch
        __shared__ float4 sh[1024];
        sh[threadIdx.x] = make_float4(1, 2, 3, 4);
        sh[threadIdx.x + 512] = make_float4(1, 2, 3, 4);
        __syncthreads();
        float thr = 1e30f;
        f16x8 ares[2][8];
        for (int r = 0; r < 2; ++r)
            for (int q = 0; q < 8; ++q)
                for (int e = 0; e < 8; ++e) ares[r][q][e] = (_Float16)(0.001f * (threadIdx.x + e + 3 * q + 7 * r));
        float4 ring[4];
        for (int s = 0; s < 3; ++s) ring[s] = sh[(threadIdx.x + 64 * s) & 1023];
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const float4 v = ring[s & 3];
                if (mode != 8 || (s & 1)) ring[(s + 3) & 3] = sh[(threadIdx.x + 64 * (s + 3) + i) & 1023];  // three slices ahead
                if constexpr (mode == 9) { a[0] = (_Float16)v.x; }  // the MFMA operand comes from the LDS read
                if constexpr (mode == 12) {  // operands as in the matching kernel: B from the LDS ring, A from 16 resident vectors
                    f16x8 bv;
                    __builtin_memcpy(&bv, &v, 16);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[0][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bv, ares[1][s], acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                float m;
                if constexpr (mode >= 7) {
                    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                    continue;
                }
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(x[s & 7]), "v"(v.x), "v"(v.y));
                asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(v.z));
                if constexpr (mode == 6 || mode >= 10) {
                    if (__builtin_expect(__any(m > thr), 0)) {
                        x[s & 7] += 1.0f;
                        thr *= 2.f;
                    }
                } else {
                    x[s & 7] = m;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (mode == 10 || mode == 12) {
                if ((i & 3) == 3) __syncthreads();
            }
            if constexpr (mode == 11) {  // + a 17th MFMA with a zero accumulator input per block
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                if ((i & 3) == 3) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                }
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
    hipMalloc(&out, 256 * 512 * 4 + (4 << 20));  // + a 4 MiB source region for the LDS-DMA of mode 16
    printf("VALU ops per MFMA: %d (16 MFMAs = 512 cycles of pipe per wave; %d VALU = %d issue cycles per wave)\n", VPM, 16 * VPM, 64 * VPM);
    for (int mode = 0; mode <= 16; ++mode) {
        switch (mode) {
            case 0: k<0><<<256, 512>>>(out, 20000); break;
            case 1: k<1><<<256, 512>>>(out, 20000); break;
            case 2: k<2><<<256, 512>>>(out, 20000); break;
            case 3: k<3><<<256, 512>>>(out, 20000); break;
            case 4: k<4><<<256, 512>>>(out, 20000); break;
            case 5: k<5><<<256, 512>>>(out, 20000); break;
            case 6: k<6><<<256, 512>>>(out, 20000); break;
            case 7: k<7><<<256, 512>>>(out, 20000); break;
            case 8: k<8><<<256, 512>>>(out, 20000); break;
            case 9: k<9><<<256, 512>>>(out, 20000); break;
            case 10: k<10><<<256, 512>>>(out, 20000); break;
            case 11: k<11><<<256, 512>>>(out, 20000); break;
            case 12: k<12><<<256, 512>>>(out, 20000); break;
            case 13: k<13><<<256, 512>>>(out, 20000); break;
            case 14: k<14><<<256, 512>>>(out, 20000); break;
            case 15: k<15><<<256, 512>>>(out, 20000); break;
            case 16: k<16><<<256, 512>>>(out, 20000); break;
        }
        hipDeviceSynchronize();
    }
    return 0;
}
