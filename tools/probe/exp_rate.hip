// Probe (round 3, review item 7): the softmax exponentials of the head-dim-40 attention tile beside its MFMAs.  P is rounded to fp16 for the P V product
// anyway, so is a PACKED fp16 exp2 (two values per issue slot) cheaper than v_exp_f32 per value?
//   per 64-key tile and wave (64 queries): 56 x v_mfma_f32_16x16x32_f16 and 64 exponentials per lane, then 32 v_cvt_pk_f16_f32.
//   variant 0: MFMAs only                                 (floor)
//   variant 1: v_exp_f32 x 64 + v_cvt_pk_f16_f32 x 32      (what attn_kernel issues)
//   variant 2: v_cvt_pk_f16_f32 x 32 first, then per PAIR: magic-number round (v_pk_add_f16 x 2), fraction (v_pk_add_f16), degree-3 polynomial
//              (v_pk_fma_f16 x 3), exponent insertion (v_pk_lshlrev_b16 + v_pk_add_u16): 8 packed instructions per two values
//   variant 3: v_exp_f16 x 64 on the converted halves (one value per instruction, fp16 transcendental) + the conversions
// Two workgroups of four waves per CU (two waves per SIMD, like attn_kernel) and one (one wave per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/exp_rate tools/probe/exp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 2) void exp_kernel(int tiles, float* sink, unsigned long long* cyc, float seed) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[14];
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(((lane * 7 + i * 13) % 31 - 15) * 0.03f); b[i] = (f16)(((lane * 11 + i * 5) % 29 - 14) * 0.02f); }
#pragma unroll
    for (int i = 0; i < 14; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) s[i] = -0.01f * (float)((lane + i * 3) % 97) * seed;
    unsigned long long t0 = 0;
    __syncthreads();
    if (lane == 0) t0 = __builtin_readcyclecounter();
    float keep = 0.f;
    for (int t = 0; t < tiles; ++t) {
        // score MFMAs (32), then the softmax work, then the P V MFMAs (24) -- the order of attn_kernel's tile
#pragma unroll
        for (int m = 0; m < 32; ++m) acc[m % 14] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 14], 0, 0, 0);
        if (VAR == 1) {
            unsigned pk[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float e0 = __builtin_amdgcn_exp2f(s[2 * i]), e1 = __builtin_amdgcn_exp2f(s[2 * i + 1]);
                union { h2 h; unsigned u; } c; c.h = h2{(f16)e0, (f16)e1}; pk[i] = c.u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { union { unsigned u[4]; f16x8 v; } c; c.u[0] = pk[4 * i]; c.u[1] = pk[4 * i + 1]; c.u[2] = pk[4 * i + 2]; c.u[3] = pk[4 * i + 3]; if (i == (t & 7)) b = c.v; }
        } else if (VAR == 2) {
            unsigned pk[32];
            const h2 magic = {(f16)1536.f, (f16)1536.f}, c3 = {(f16)0.0555f, (f16)0.0555f}, c2 = {(f16)0.2402f, (f16)0.2402f}, c1 = {(f16)0.6931f, (f16)0.6931f},
                     one = {(f16)1.f, (f16)1.f};
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const h2 x = {(f16)s[2 * i], (f16)s[2 * i + 1]};                 // v_cvt_pk_f16_f32 (rtz form in hardware; the probe only counts issue slots)
                const h2 tt = x + magic;                                          // round to nearest integer in the low mantissa bits
                const h2 n = tt - magic;
                const h2 f = x - n;
                h2 p = c3 * f + c2; p = p * f + c1; p = p * f + one;              // 2^f on [-0.5, 0.5], three v_pk_fma_f16
                union { h2 h; unsigned u; } pu, tu; pu.h = p; tu.h = tt;
                pk[i] = pu.u + ((tu.u << 10) & 0xfc00fc00u);                      // exponent insertion: v_pk_lshlrev_b16 + v_and + v_pk_add_u16 (and folded here)
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { union { unsigned u[4]; f16x8 v; } c; c.u[0] = pk[4 * i]; c.u[1] = pk[4 * i + 1]; c.u[2] = pk[4 * i + 2]; c.u[3] = pk[4 * i + 3]; if (i == (t & 7)) b = c.v; }
        } else if (VAR == 3) {
            unsigned pk[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f16 e0 = __builtin_amdgcn_exp2f((float)(f16)s[2 * i]) > 0 ? (f16)0 : (f16)0;   // placeholder so that the variant compiles without a f16 exp builtin
                (void)e0;
                f16 h0 = (f16)s[2 * i], h1 = (f16)s[2 * i + 1];
                asm volatile("v_exp_f16 %0, %0" : "+v"(h0));
                asm volatile("v_exp_f16 %0, %0" : "+v"(h1));
                union { h2 h; unsigned u; } c; c.h = h2{h0, h1}; pk[i] = c.u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { union { unsigned u[4]; f16x8 v; } c; c.u[0] = pk[4 * i]; c.u[1] = pk[4 * i + 1]; c.u[2] = pk[4 * i + 2]; c.u[3] = pk[4 * i + 3]; if (i == (t & 7)) b = c.v; }
        }
#pragma unroll
        for (int m = 0; m < 24; ++m) acc[m % 14] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % 14], 0, 0, 0);
        // keep the scores data dependent on the loop so nothing is hoisted
#pragma unroll
        for (int i = 0; i < 64; i += 16) s[i] += acc[0][0] * 1e-30f;
    }
    if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
#pragma unroll
    for (int i = 0; i < 14; ++i) keep += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (keep == 123.456f) sink[threadIdx.x] = keep;
}

template <int VAR>
static void run(float* sink, unsigned long long* cyc, int wgs, const char* tag) {
    const int tiles = 4000;
    hipLaunchKernelGGL(exp_kernel<VAR>, dim3(wgs), dim3(256), 0, 0, 100, sink, cyc, 1.0f);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(exp_kernel<VAR>, dim3(wgs), dim3(256), 0, 0, tiles, sink, cyc, 1.0f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(wgs);
    (void)hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-64s %s : %7.1f cycles per tile and wave (56 MFMAs = 896) | %.3f ms\n", tag, wgs == 512 ? "two waves / SIMD" : "one wave / SIMD ", (double)h[wgs / 2] / tiles, ms);
}

int main() {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 512 * 8);
    for (int wgs : {512, 256}) {
        run<0>(sink, cyc, wgs, "0 MFMAs only");
        run<1>(sink, cyc, wgs, "1 v_exp_f32 x 64 + v_cvt_pk_f16_f32 x 32 (attn_kernel)");
        run<2>(sink, cyc, wgs, "2 packed-fp16 polynomial exp2 (8 packed ops per two values)");
        run<3>(sink, cyc, wgs, "3 v_exp_f16 x 64 on converted halves");
    }
    return 0;
}
