// Probe (round 3, review item 7 and the design question behind attn40_lw_kernel): what do the softmax exponentials of the head-dim-40 attention tile cost
// beside its MFMAs, by WHERE they are issued?  One slot = the work of 32 keys x 64 queries of one wave: 28 x v_mfma_f32_16x16x32_f16 (16 score + 12 P V),
// 32 x v_exp_f32 + 16 x v_cvt_pk_f16_f32, every instruction an asm volatile with real data flow (scores -> exp -> cvt -> B operand of the P V MFMAs of a later
// slot, as in the kernel), no LDS traffic.  (The first version of this probe kept its exponent arguments loop-invariant and hipcc hoisted 60 of the 64
// exponentials out of the loop: its numbers were withdrawn.)
//   mode 0: MFMAs only                      mode 1: exponentials + conversions only
//   mode 2: interleaved, one or two VALU instructions behind every MFMA (attn40_lw_kernel's first stream)
//   mode 3: blocked: 28 MFMAs, then the 48 VALU instructions
//   mode 4: blocked, waves 4-7 run the VALU block FIRST: SIMD partners in anti-phase (two waves per SIMD only)
//   mode 5: packed-fp16 polynomial exp2 in the interleaved stream (8 packed instructions per two values; numerically unusable)
//   mode 6: mode 2 with the second k step of the score MFMAs (head dim 32..39 of 40) as v_mfma_f32_16x16x16_f16: same instruction count and pipe time, half the
//           multiplies of those 16 instructions -- does the lower energy buy clock?  (compare the ms column, not the cycles)
// waves per SIMD: 1 (256-thread workgroup, one per CU) or 2 (512 threads); one s_barrier per two slots in every mode.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probe/bin/exp_rate tools/probe/exp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <utility>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_ip(f32x4& c, const f16x8& a, const f16x8& b) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_ip16(f32x4& c, const f16x4& a, const f16x4& b) { asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_c(f32x4& d, const f16x8& a, const f16x8& b, const f32x4& c) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c)); }
#define p_exp(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define p_cvt(d, a, b) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
template <int N, class F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

template <int MODE, int WPS>
__global__ __launch_bounds__(WPS * 256) void slot_kernel(int slots2, float* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f16x8 qf[4][2], kf[4], vf[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { qf[t][0][i] = (f16)(((lane * 7 + i * 13 + t) % 31 - 15) * 0.02f); qf[t][1][i] = (f16)(((lane * 5 + i * 3 + t) % 29 - 14) * 0.02f); }
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j][i] = (f16)(((lane * 11 + i * 5 + j) % 29 - 14) * 0.03f);
#pragma unroll
        for (int a = 0; a < 3; ++a) vf[a][i] = (f16)(((lane * 3 + i * 7 + a) % 23 - 11) * 0.03f);
    }
    f32x4 sc[2][2][4], o[3][4], negm[4];
    u32x4 pfw[2][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        negm[t] = f32x4{-1.f, -1.f, -1.f, -1.f};
#pragma unroll
        for (int a = 0; a < 3; ++a) o[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) { sc[u][0][t] = f32x4{-1.f, -2.f, -3.f, -4.f}; sc[u][1][t] = f32x4{-1.5f, -2.5f, -3.5f, -0.5f}; pfw[u][t] = u32x4{0x3c003c00u, 0x38003800u, 0x34003400u, 0x30003000u}; }
    }
    // E instruction Q (0..47) of the slot whose scores are sc[SP] and whose P fragment is pfw[SP]: items (t, kt') of 4 exp + 2 cvt, conversions one item behind
    auto e_instr = [&](auto sp_tag, auto q_tag) {
        constexpr int SP = decltype(sp_tag)::value, Q = decltype(q_tag)::value;
        (void)sc; (void)pfw;                                  // (named outside the discarded branches: clang decides a generic lambda's implicit captures there)
        if constexpr (MODE == 5) {
            // packed-fp16 polynomial: per item (4 values = 2 pairs): cvt x2, then per pair 8 packed instructions; issued as 6 "instructions" of this list = 18 / 6 = 3 each
            constexpr int K = Q / 6, R = Q % 6;
            if constexpr (R < 2) {
                unsigned x; p_cvt(x, sc[SP][K % 2][K / 2][2 * R], sc[SP][K % 2][K / 2][2 * R + 1]);
                unsigned tt, n, f, pp;
                asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(tt) : "v"(x), "v"(0x66006600u));
                asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(n) : "v"(tt), "v"(0x66006600u));
                asm volatile("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(f) : "v"(x), "v"(n));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(pp) : "v"(f), "v"(0x2b1b2b1bu), "v"(0x33b033b0u));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(pp) : "v"(pp), "v"(f), "v"(0x398c398cu));
                asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(pp) : "v"(pp), "v"(f), "v"(0x3c003c00u));
                asm volatile("v_pk_lshlrev_b16 %0, 10, %1" : "=v"(tt) : "v"(tt));
                asm volatile("v_pk_add_u16 %0, %1, %2" : "=v"(pfw[SP][K / 2][2 * (K % 2) + R]) : "v"(pp), "v"(tt));
            }
        } else {
            if constexpr (Q < 4) p_exp(sc[SP][0][0][Q]);
            else {
                constexpr int QQ = Q - 4, K = 1 + QQ / 6, R = QQ % 6;
                if constexpr (K <= 7) {
                    if constexpr (R < 4) p_exp(sc[SP][K % 2][K / 2][R]);
                    else { constexpr int KP = K - 1, C = R - 4; p_cvt(pfw[SP][KP / 2][2 * (KP % 2) + C], sc[SP][KP % 2][KP / 2][2 * C], sc[SP][KP % 2][KP / 2][2 * C + 1]); }
                } else { constexpr int C = QQ - 42; p_cvt(pfw[SP][3][2 + C], sc[SP][1][3][2 * C], sc[SP][1][3][2 * C + 1]); }
            }
        }
    };
    auto mfma_m = [&](auto par_tag, auto m_tag) {          // MFMA m (0..27) of a slot of parity PAR, in the kernel's group order S | PV | S | PV | S | PV | S
        constexpr int PAR = decltype(par_tag)::value, M = decltype(m_tag)::value, G = M / 4, T = M % 4;
        if constexpr (G % 2 == 0) {
            constexpr int J = G / 2, KT = J / 2, KS = J % 2;
            if constexpr (KS == 0) mfma_c(sc[PAR][KT][T], kf[J], qf[T][0], negm[T]);
            else if constexpr (MODE == 6) {
                const f16x4 a4 = {kf[J][0], kf[J][1], kf[J][2], kf[J][3]}, b4 = {qf[T][1][0], qf[T][1][1], qf[T][1][2], qf[T][1][3]};
                mfma_ip16(sc[PAR][KT][T], a4, b4);
            } else mfma_ip(sc[PAR][KT][T], kf[J], qf[T][1]);
        } else {
            constexpr int A = G / 2;
            union { u32x4 u; f16x8 f; } b; b.u = pfw[PAR][T];
            mfma_ip(o[A][T], vf[A], b.f);
        }
    };
    auto slot = [&](auto par_tag, auto vfirst_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool VFIRST = decltype(vfirst_tag)::value;
        using SP = std::integral_constant<int, 1 - PAR>;
        if constexpr (MODE == 0) sfor<28>([&](auto m) { mfma_m(par_tag, m); });
        else if constexpr (MODE == 1) sfor<48>([&](auto q) { e_instr(SP{}, q); });
        else if constexpr (MODE == 2 || MODE == 5 || MODE == 6) {
            sfor<28>([&](auto m) {
                constexpr int M = decltype(m)::value, LO = 48 * M / 28, HI = 48 * (M + 1) / 28;
                mfma_m(par_tag, m);
                sfor<HI - LO>([&](auto i) { e_instr(SP{}, std::integral_constant<int, LO + decltype(i)::value>{}); });
            });
        } else {
            if constexpr (VFIRST) { sfor<48>([&](auto q) { e_instr(SP{}, q); }); sfor<28>([&](auto m) { mfma_m(par_tag, m); }); }
            else { sfor<28>([&](auto m) { mfma_m(par_tag, m); }); sfor<48>([&](auto q) { e_instr(SP{}, q); }); }
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 4 && w >= 4) {
        for (int s = 0; s < slots2; ++s) { slot(I0{}, std::true_type{}); slot(I1{}, std::true_type{}); __builtin_amdgcn_s_barrier(); }
    } else {
        for (int s = 0; s < slots2; ++s) { slot(I0{}, std::false_type{}); slot(I1{}, std::false_type{}); __builtin_amdgcn_s_barrier(); }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && w == 0) cyc[blockIdx.x] = t1 - t0;
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int a = 0; a < 3; ++a) r += o[a][t][0] + o[a][t][1] + o[a][t][2] + o[a][t][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) r += sc[u][0][t][0] + sc[u][1][t][3] + (float)pfw[u][t][0];
    }
    if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int MODE, int WPS>
static void run(float* sink, unsigned long long* cyc, const char* tag) {
    const int slots2 = 4000;
    auto k = slot_kernel<MODE, WPS>;
    hipLaunchKernelGGL(k, dim3(256), dim3(WPS * 256), 0, 0, 100, sink, cyc);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(WPS * 256), 0, 0, slots2, sink, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_tile = (double)h[128] / slots2;              // two slots = one 64-key tile of one wave
    printf("%-72s %d wave(s) / SIMD : %7.1f cycles per 64-key tile and wave = %7.1f of SIMD time per wave-tile (56 MFMAs = 896) | clock %.2f GHz | %.3f ms\n", tag, WPS, per_tile,
           per_tile / WPS, (double)h[128] / (ms * 1e-3) / 1e9, ms);
}

int main() {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 256 * 8);
    run<0, 1>(sink, cyc, "0 MFMAs only");
    run<1, 1>(sink, cyc, "1 v_exp_f32 x 64 + v_cvt_pk_f16_f32 x 32 only");
    run<2, 1>(sink, cyc, "2 interleaved (1-2 VALU behind every MFMA)");
    run<3, 1>(sink, cyc, "3 blocked (28 MFMAs, then 48 VALU)");
    run<5, 1>(sink, cyc, "5 interleaved, packed-fp16 polynomial exp2");
    run<6, 1>(sink, cyc, "6 interleaved, k step 2 of the scores as 16x16x16");
    run<2, 1>(sink, cyc, "2 interleaved (again)");
    run<6, 1>(sink, cyc, "6 interleaved, k step 2 of the scores as 16x16x16 (again)");
    run<0, 2>(sink, cyc, "0 MFMAs only");
    run<1, 2>(sink, cyc, "1 v_exp_f32 x 64 + v_cvt_pk_f16_f32 x 32 only");
    run<2, 2>(sink, cyc, "2 interleaved");
    run<3, 2>(sink, cyc, "3 blocked, both waves of a SIMD in the same order");
    run<4, 2>(sink, cyc, "4 blocked, waves 4-7 VALU block first (anti-phase)");
    run<5, 2>(sink, cyc, "5 interleaved, packed-fp16 polynomial exp2");
    return 0;
}
