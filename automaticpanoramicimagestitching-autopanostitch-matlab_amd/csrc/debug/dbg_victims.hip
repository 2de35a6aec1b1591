// Debug target only (`make debug`): known-answer victims for the co-residency finding.  Each
// variant runs one packed-f32 instruction pattern in a loop and checks every result against the same arithmetic done with
// plain v_mul_f32 / v_add_f32 / v_fma_f32 in the same lane; scripts/probe/probe_corun_victims.py runs them on a worker
// stream while the main thread keeps the int8-MFMA co-runner of dbg_corun.hip resident.  Not part of libaps_hip.so.
#include "../aps_internal.h"

namespace aps {
typedef float v2f __attribute__((ext_vector_type(2)));

// scalar reference of a*b then +c (two roundings), one component
__device__ __forceinline__ float ref_mul_add(float a, float b, float c) {
    float m, r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(b));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(m), "v"(c));
    return r;
}

template <int VAR>
__global__ __launch_bounds__(256) void dbg_victim_kernel(int iters, unsigned long long* __restrict__ mism, float* __restrict__ buf) {
    const int lane = threadIdx.x & 63;
    // operands in [0.5, 2): products and sums stay finite and change every iteration
    v2f p = {1.0f + 0.001f * lane, 1.5f - 0.002f * lane};
    v2f q = {0.75f + 0.0005f * (blockIdx.x & 1023), 1.25f - 0.0003f * (threadIdx.x >> 6)};
    const v2f t = {0.125f, -0.0625f};
    unsigned int bad = 0;
    for (int it = 0; it < iters; ++it) {
        v2f r;
        float ex, ey;
        if (VAR <= 3) {
            // dependent pair, the compiler's pattern in refine_kernel: v_pk_mul_f32 -> (s_nop N) -> v_pk_add_f32 on the result
            if (VAR == 0)
                asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(r) : "v"(p), "v"(q), "v"(t));
            else if (VAR == 1)
                asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 1\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(r) : "v"(p), "v"(q), "v"(t));
            else if (VAR == 2)
                asm volatile("v_pk_mul_f32 %0, %1, %2\n\ts_nop 7\n\tv_pk_add_f32 %0, %0, %3" : "=&v"(r) : "v"(p), "v"(q), "v"(t));
            else {  // the consumer is a plain VALU instruction per half (v_add_f32), one wait state in between
                float rx, ry;
                asm volatile("v_pk_mul_f32 v[100:101], %2, %3\n\ts_nop 0\n\tv_add_f32 %0, v100, %4\n\tv_add_f32 %1, v101, %5"
                             : "=&v"(rx), "=&v"(ry)
                             : "v"(p), "v"(q), "v"(t.x), "v"(t.y)
                             : "v100", "v101");
                r.x = rx;
                r.y = ry;
            }
            ex = ref_mul_add(p.x, q.x, t.x);
            ey = ref_mul_add(p.y, q.y, t.y);
        } else if (VAR == 4) {
            // independent packed instructions only (nothing reads a packed result within eight instructions)
            v2f r0, r1, r2, r3;
            asm volatile("v_pk_mul_f32 %0, %4, %5\n\tv_pk_mul_f32 %1, %4, %6\n\tv_pk_mul_f32 %2, %5, %6\n\tv_pk_mul_f32 %3, %6, %6\n\t"
                         "s_nop 7\n\ts_nop 7"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                         : "v"(p), "v"(q), "v"(t));
            float e0x, e0y;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0x) : "v"(p.x), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e0y) : "v"(p.y), "v"(q.y));
            float e1x, e1y;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1x) : "v"(p.x), "v"(t.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e1y) : "v"(p.y), "v"(t.y));
            bad += (__float_as_uint(r1.x) != __float_as_uint(e1x)) + (__float_as_uint(r1.y) != __float_as_uint(e1y));
            r = r0;
            ex = e0x;
            ey = e0y;
            (void)r2;
            (void)r3;
        } else if (VAR == 5) {
            // v_pk_mov_b32 with op_sel (the 64-bit shuffle the compiler emits), consumed by a packed multiply after one wait state
            v2f s;
            asm volatile("v_pk_mov_b32 %0, %2, %3 op_sel:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %1, %0, %3" : "=&v"(s), "=&v"(r) : "v"(p), "v"(q));
            // op_sel:[1,0]: low result = src0's HIGH half, high result = src1's LOW half
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(p.y), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(q.x), "v"(q.y));
        } else if (VAR == 6) {
            // dependent v_pk_fma_f32 back to back (no wait state at all), the blur chains' instruction at distance 1
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3\n\tv_pk_fma_f32 %0, %0, %2, %3" : "=&v"(r) : "v"(p), "v"(q), "v"(t));
            float f0x, f0y;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0x) : "v"(p.x), "v"(q.x), "v"(t.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f0y) : "v"(p.y), "v"(q.y), "v"(t.y));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ex) : "v"(f0x), "v"(q.x), "v"(t.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ey) : "v"(f0y), "v"(q.y), "v"(t.y));
        } else if (VAR == 8) {
            // plain VALU writes ONE half of a pair, the packed instruction reads the pair at once (refine_kernel: v_mov_b32 v1, s26;
            // v_mov_b32 v7, 1.0; v_pk_add_f32 v[0:1], v[0:1], v[6:7])
            float rx, ry;
            asm volatile("v_mov_b64 v[100:101], %2\n\tv_mov_b64 v[102:103], %3\n\ts_nop 7\n\t"
                         "v_mov_b32 v101, %4\n\tv_mov_b32 v103, 1.0\n\tv_pk_add_f32 v[100:101], v[100:101], v[102:103]\n\ts_nop 7\n\t"
                         "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
                         : "=&v"(rx), "=&v"(ry)
                         : "v"(p), "v"(q), "v"(t.x)
                         : "v100", "v101", "v102", "v103");
            r.x = rx;
            r.y = ry;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(ex) : "v"(p.x), "v"(q.x));
            ey = t.x + 1.0f;
        } else if (VAR == 9) {
            // broadcast of src1's low half (op_sel_hi:[1,0]) and the negating add, each consumed after one wait state
            v2f u;
            asm volatile("v_pk_mul_f32 %1, %2, %3 op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %0, %1, 0 neg_lo:[1,1] neg_hi:[1,1]\n\ts_nop 0\n\t"
                         "v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]"
                         : "=&v"(r), "=&v"(u)
                         : "v"(p), "v"(q), "v"(t));
            float mx, my;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(mx) : "v"(p.x), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(my) : "v"(p.y), "v"(q.x));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ex) : "v"(-mx), "v"(t.x));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ey) : "v"(-my), "v"(t.y));
        } else if (VAR == 10) {
            // the packed instruction runs under an EXEC mask written just before it (divergent code: s_and_b64 exec -> packed op);
            // lanes outside the mask must keep their sentinel
            const unsigned long long mask = __ballot(((lane * 2654435761u + it * 40503u) >> 7) & 1);
            v2f rr = {-7.0f, -9.0f};
            unsigned long long save;
            asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %4\n\tv_pk_mul_f32 %0, %2, %3\n\ts_mov_b64 exec, %1"
                         : "+v"(rr), "=&s"(save)
                         : "v"(p), "v"(q), "s"(mask));
            r = rr;
            const bool on = (mask >> lane) & 1ull;
            float mx, my;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(mx) : "v"(p.x), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(my) : "v"(p.y), "v"(q.y));
            ex = on ? mx : -7.0f;
            ey = on ? my : -9.0f;
        } else if (VAR == 11) {
            // v_pk_mov_b32 with an SGPR pair whose high half was written by s_mov_b32 just before (refine_kernel's sequence)
            float rx, ry;
            asm volatile("s_mov_b32 s40, %4\n\ts_mov_b32 s41, %4\n\tv_pk_mov_b32 v[100:101], s[40:41], %2 op_sel:[1,0]\n\ts_nop 0\n\t"
                         "v_pk_mul_f32 v[100:101], v[100:101], %3\n\ts_nop 7\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101"
                         : "=&v"(rx), "=&v"(ry)
                         : "v"(p), "v"(q), "s"(__builtin_amdgcn_readfirstlane(__float_as_int(1.5f + 0.001f * (it & 255))))
                         : "v100", "v101", "s40", "s41");
            r.x = rx;
            r.y = ry;
            const float sv = 1.5f + 0.001f * (it & 255);
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(sv), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(p.x), "v"(q.y));
        } else if (VAR == 12) {
            // EXEC shrinks, one plain instruction runs, EXEC grows back to the full mask (s_or_b64), then the packed instruction at once:
            // every lane must hold the product (descr_kernel: s_or_b64 exec -> v_add_u32 -> v_pk_mov_b32)
            const unsigned long long mask = __ballot(((lane * 2654435761u + it * 40503u) >> 9) & 1);
            v2f rr = {-7.0f, -9.0f};
            unsigned long long save;
            float dummy;
            asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %5\n\tv_add_f32 %2, 1.0, %6\n\ts_or_b64 exec, exec, %1\n\tv_pk_mul_f32 %0, %3, %4"
                         : "+v"(rr), "=&s"(save), "=&v"(dummy)
                         : "v"(p), "v"(q), "s"(mask), "v"(t.x));
            r = rr;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(p.x), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(p.y), "v"(q.y));
        } else if (VAR == 13) {
            // the packed result is STORED at once (global_store_dwordx2 reads the pair with no wait state), then read back
            float* slot = buf + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
            v2f back;
            asm volatile("v_pk_mul_f32 %0, %2, %3\n\tglobal_store_dwordx2 %4, %0, off\n\ts_waitcnt vmcnt(0)\n\tglobal_load_dwordx2 %1, %4, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(r), "=&v"(back)
                         : "v"(p), "v"(q), "v"(slot)
                         : "memory");
            r = back;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(p.x), "v"(q.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(p.y), "v"(q.y));
        } else if (VAR == 14) {
            // v_pk_mov_b32 builds a 64-bit ADDRESS that a load uses at once (descr_kernel's 64-bit shuffles); the destination pair
            // holds another valid address before, so a stale read loads the other slot's value instead of faulting
            float* slot_a = buf + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
            float* slot_b = slot_a + 2;
            slot_a[0] = 111.0f + (float)(it & 7);
            slot_b[0] = 222.0f;
            __builtin_amdgcn_s_waitcnt(0);
            float got;
            unsigned long long a = (unsigned long long)slot_a, b = (unsigned long long)slot_b;
            // sw = (hi(a), lo(a)) swapped halves of slot_a's address; the pk_mov puts them back in order over a pair that held slot_b
            unsigned long long sw = (a >> 32) | (a << 32), dst = b;
            asm volatile("s_nop 7\n\tv_pk_mov_b32 %1, %2, %2 op_sel:[1,0]\n\tglobal_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(got), "+v"(dst)
                         : "v"(sw)
                         : "memory");
            r.x = got;
            r.y = 0.f;
            ex = 111.0f + (float)(it & 7);
            ey = 0.f;
        } else if (VAR >= 15 && VAR <= 18) {
            // the packed result is read by the NEXT plain VALU instruction, NO wait state - what the compiler emits in refine_kernel
            // (v_pk_fma_f32 v[8:9], .. op_sel_hi:[0,1,1]; v_cmp_nlt_f32 .., |v8|, ..) and descr_kernel (v_pk_mul_f32 v[14:15], ..
            // op_sel:[1,0] op_sel_hi:[0,0]; v_sub_f32 v18, .., v15).  15: pk_mul, low half read; 16: pk_mul, high half read;
            // 17: pk_fma op_sel_hi:[0,1,1], low half read; 18: as 15 with one wait state (s_nop 0) in between
            float got;
            if (VAR == 15)
                asm volatile("v_pk_mul_f32 v[100:101], %1, %2\n\tv_add_f32 %0, 1.0, v100\n\ts_nop 7" : "=&v"(got) : "v"(p), "v"(q) : "v100", "v101");
            else if (VAR == 16)
                asm volatile("v_pk_mul_f32 v[100:101], %1, %2\n\tv_add_f32 %0, 1.0, v101\n\ts_nop 7" : "=&v"(got) : "v"(p), "v"(q) : "v100", "v101");
            else if (VAR == 17)
                asm volatile("v_pk_fma_f32 v[100:101], %1, %2, %3 op_sel_hi:[0,1,1]\n\tv_add_f32 %0, 1.0, v100\n\ts_nop 7"
                             : "=&v"(got)
                             : "v"(p), "v"(q), "v"(t)
                             : "v100", "v101");
            else
                asm volatile("v_pk_mul_f32 v[100:101], %1, %2\n\ts_nop 0\n\tv_add_f32 %0, 1.0, v100\n\ts_nop 7" : "=&v"(got) : "v"(p), "v"(q) : "v100", "v101");
            float m;
            if (VAR == 16)
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(p.y), "v"(q.y));
            else if (VAR == 17)
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(m) : "v"(p.x), "v"(q.x), "v"(t.x));
            else
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m) : "v"(p.x), "v"(q.x));
            asm volatile("v_add_f32 %0, 1.0, %1" : "=v"(ex) : "v"(m));
            r.x = got;
            r.y = 0.f;
            ey = 0.f;
        } else if (VAR == 19 || VAR == 20) {
            // operand kinds of the compiler's code that the variants above lack: 19 = an SGPR pair as the second source with the low
            // half broadcast (refine_kernel: v_pk_mul_f32 V, V, S op_sel_hi:[1,0]), 20 = an inline constant (descr_kernel:
            // v_pk_add_f32 V, V, 2.0 op_sel_hi:[1,0]); each consumed by a packed add after one wait state
            const float sv = 1.25f + 0.001f * (it & 127);
            if (VAR == 19) {
                asm volatile("s_mov_b32 s40, %3\n\ts_mov_b32 s41, %3\n\tv_pk_mul_f32 %0, %1, s[40:41] op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %2"
                             : "=&v"(r)
                             : "v"(p), "v"(t), "s"(__builtin_amdgcn_readfirstlane(__float_as_int(sv)))
                             : "s40", "s41");
                float mx, my;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(mx) : "v"(sv), "v"(p.x));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(my) : "v"(sv), "v"(p.y));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(ex) : "v"(mx), "v"(t.x));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(ey) : "v"(my), "v"(t.y));
            } else {
                asm volatile("v_pk_add_f32 %0, %1, 2.0 op_sel_hi:[1,0]\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, %2" : "=&v"(r) : "v"(p), "v"(q));
                float ax, ay;
                asm volatile("v_add_f32 %0, 2.0, %1" : "=v"(ax) : "v"(p.x));
                asm volatile("v_add_f32 %0, 2.0, %1" : "=v"(ay) : "v"(p.y));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ex) : "v"(ax), "v"(q.x));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(ey) : "v"(ay), "v"(q.y));
            }
        } else if (VAR == 21) {
            // packed instructions inside a lane-divergent loop (per-lane trip counts: the compiler's exec masking around them)
            v2f acc = p;
            const int trips = 1 + ((lane * 7 + it) & 7);
            float ax = p.x, ay = p.y;
            for (int k = 0; k < trips; ++k) {
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\ts_nop 0\n\tv_pk_add_f32 %0, %0, %2" : "+v"(acc) : "v"(q), "v"(t));
                ax = ref_mul_add(ax, q.x, t.x);
                ay = ref_mul_add(ay, q.y, t.y);
                if (ax > 64.0f) {  // (keeps the values bounded; data dependent like the kernels' early exits)
                    acc.x = ax = 1.0f;
                    acc.y = ay = 1.5f;
                }
            }
            r = acc;
            ex = ax;
            ey = ay;
        } else {
            // VAR 7: the same work without any packed instruction (control)
            r.x = ref_mul_add(p.x, q.x, t.x);
            r.y = ref_mul_add(p.y, q.y, t.y);
            ex = ref_mul_add(p.x, q.x, t.x);
            ey = ref_mul_add(p.y, q.y, t.y);
        }
        bad += (__float_as_uint(r.x) != __float_as_uint(ex)) + (__float_as_uint(r.y) != __float_as_uint(ey));
        // next operands from the EXPECTED values (a wrong result is counted once and does not propagate), kept in [0.5, 2)
        p.x = 0.5f + 0.5f * (ex - floorf(ex)) + 0.25f;
        p.y = 0.5f + 0.5f * (ey - floorf(ey)) + 0.125f;
        q.x = 2.0f - 0.5f * p.y;
        q.y = 0.5f + 0.5f * p.x;
    }
    unsigned long long m = bad;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off);
    if (lane == 0 && m) atomicAdd(mism, m);
}
}  // namespace aps

// Runs variant `var` (n_wg workgroups of 256 threads, `iters` iterations per lane) on the calling thread's stream and returns
// the number of results that differed from their scalar reference in *mismatches.
extern "C" int aps_dbg_victim(int var, int n_wg, int iters, unsigned long long* mismatches) {
    using namespace aps;
    return guarded([&] {
        ctx();
        Ws<unsigned long long> d(1);
        Ws<float> buf((size_t)n_wg * 256 * 4);
        APS_HIP(hipMemsetAsync(d, 0, 8, stream()));
        switch (var) {
            case 0: dbg_victim_kernel<0><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 1: dbg_victim_kernel<1><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 2: dbg_victim_kernel<2><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 3: dbg_victim_kernel<3><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 4: dbg_victim_kernel<4><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 5: dbg_victim_kernel<5><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 6: dbg_victim_kernel<6><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 8: dbg_victim_kernel<8><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 9: dbg_victim_kernel<9><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 10: dbg_victim_kernel<10><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 11: dbg_victim_kernel<11><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 12: dbg_victim_kernel<12><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 13: dbg_victim_kernel<13><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 14: dbg_victim_kernel<14><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 15: dbg_victim_kernel<15><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 16: dbg_victim_kernel<16><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 17: dbg_victim_kernel<17><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 18: dbg_victim_kernel<18><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 19: dbg_victim_kernel<19><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 20: dbg_victim_kernel<20><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            case 21: dbg_victim_kernel<21><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
            default: dbg_victim_kernel<7><<<n_wg, 256, 0, stream()>>>(iters, d, buf); break;
        }
        check_launch("dbg_victim_kernel");
        APS_HIP(hipMemcpyAsync(mismatches, d, 8, hipMemcpyDeviceToHost, stream()));
        APS_HIP(hipStreamSynchronize(stream()));
    });
}
