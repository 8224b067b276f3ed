// Probe (round 3): what does one MFMA-issuing wave per SIMD lose to the other instructions of a GEMM / conv step?  Four waves per workgroup (one per SIMD), one
// workgroup per CU, step = 20 items x 4 v_mfma_f32_16x16x32_f16 (the conv3_lw_kernel step), one s_barrier per step.  Variants:
//   ACC    : 'a' accumulators pinned to AGPRs (inline-asm MFMA, "+a") | 'v' pinned to VGPRs ("+v") -- conv3_lw_kernel's builtin MFMAs got VGPRs from hipcc
//   READS  : LDS fragment reads per item (0, 1, 2) as inline-asm ds_read_b128 with a counted lgkmcnt in front of each item's MFMAs (lookahead 4 items)
//   VALU   : extra v_add_u32 per item (0, 2, 4) -- address arithmetic
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mfma_stream tools/probe/mfma_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <utility>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <char ACC>
__device__ __forceinline__ void mfma(f32x4& c, const f16x8& a, const f16x8& b) {
    if (ACC == 'a') asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void lds_read(f16x8& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
template <int N>
__device__ __forceinline__ void lds_wait(f16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int NQ = 20, NACC = 40, LA = 4, RS = 5;

template <char ACC, int READS, int VALU, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void stream_kernel(int steps, float* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 65536 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0x3c003c00u + (i * 2654435761u >> 20);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 fa[4], fw[RS], fx[RS];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[j][i] = (f16)(((lane * 7 + i * 13 + j) % 31 - 15) * 0.03f);
#pragma unroll
        for (int j = 0; j < RS; ++j) { fw[j][i] = (f16)(((lane * 11 + i * 5 + j) % 29 - 14) * 0.02f); fx[j] = fw[j]; }
    }
    unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (w & 3) * 8192 + (lane & 15) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) * 16);
    unsigned va = lane;
    unsigned long long t0 = 0;
    if (w >= 4) {                                     // (NWAVES = 8: idle partner waves that only take part in the barriers)
        for (int s = 0; s < steps; ++s) __builtin_amdgcn_s_barrier();
        return;
    }
    if (READS > 0) {                                  // prologue = the reads of items NQ - LA .. NQ - 1
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            if (r == 0) lds_read<0>(fw[0], base); if (r == 1) lds_read<2048>(fw[1], base); if (r == 2) lds_read<4096>(fw[2], base); if (r == 3) lds_read<6144>(fw[3], base);
            if (READS > 1) { if (r == 0) lds_read<128>(fx[0], base); if (r == 1) lds_read<2048 + 128>(fx[1], base); if (r == 2) lds_read<4096 + 128>(fx[2], base); if (r == 3) lds_read<6144 + 128>(fx[3], base); }
        }
    }
    if (lane == 0) t0 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (READS > 0) {
#pragma unroll
        for (int r = 0; r < LA; ++r) {
            if (r == 0) lds_read<0>(fw[0], base); if (r == 1) lds_read<2048>(fw[1], base); if (r == 2) lds_read<4096>(fw[2], base); if (r == 3) lds_read<6144>(fw[3], base);
            if (READS > 1) { if (r == 0) lds_read<128>(fx[0], base); if (r == 1) lds_read<2048 + 128>(fx[1], base); if (r == 2) lds_read<4096 + 128>(fx[2], base); if (r == 3) lds_read<6144 + 128>(fx[3], base); }
        }
    }
    for (int s = 0; s < steps; ++s) {
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            if constexpr (q == NQ - LA) __builtin_amdgcn_s_barrier();
            if constexpr (READS > 0) {
                lds_read<((q + LA) % 16) * 2048>(fw[(q + LA) % RS], base);
                if constexpr (READS > 1) lds_read<((q + LA) % 16) * 2048 + 1024>(fx[(q + LA) % RS], base);
                lds_wait<LA * READS>(fw[q % RS]);
                if constexpr (READS > 1) asm volatile("" : "+v"(fx[q % RS]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma<ACC>(acc[(q * 4 + j) % NACC], READS > 1 && (j & 1) ? fx[q % RS] : fw[q % RS], fa[j]);
#pragma unroll
            for (int v = 0; v < VALU; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(va) : "v"(base));
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0 && w == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    float r = (float)va;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 123.456f) sink[tid] = r;
}

template <char ACC, int READS, int VALU, int NWAVES>
static void run(float* sink, unsigned long long* cyc, const char* tag) {
    auto k = stream_kernel<ACC, READS, VALU, NWAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int steps = 2000;
    hipLaunchKernelGGL(k, dim3(256), dim3(NWAVES * 64), 65536, 0, 200, sink, cyc);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(NWAVES * 64), 65536, 0, steps, sink, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-78s : %7.1f cycles/step (MFMA floor 1280) | %.3f ms, %.2f PFLOP/s, clock %.2f GHz\n", tag, (double)h[128] / steps, ms,
           256.0 * 4 * 80 * 16384.0 * steps / (ms * 1e-3) / 1e15, (double)h[128] / (ms * 1e-3) / 1e9);
}

int main() {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 4096); (void)hipMalloc(&cyc, 256 * 8);
    run<'a', 0, 0, 4>(sink, cyc, "AGPR accumulators, MFMAs + barrier only");
    run<'v', 0, 0, 4>(sink, cyc, "VGPR accumulators, MFMAs + barrier only");
    run<'a', 1, 0, 4>(sink, cyc, "AGPR, + 1 ds_read_b128 and a counted wait per item");
    run<'v', 1, 0, 4>(sink, cyc, "VGPR, + 1 ds_read_b128 and a counted wait per item");
    run<'a', 2, 0, 4>(sink, cyc, "AGPR, + 2 ds_read_b128 per item");
    run<'v', 2, 0, 4>(sink, cyc, "VGPR, + 2 ds_read_b128 per item");
    run<'a', 1, 2, 4>(sink, cyc, "AGPR, + 1 ds_read_b128 + 2 v_add_u32 per item");
    run<'v', 1, 2, 4>(sink, cyc, "VGPR, + 1 ds_read_b128 + 2 v_add_u32 per item");
    run<'a', 1, 4, 4>(sink, cyc, "AGPR, + 1 ds_read_b128 + 4 v_add_u32 per item");
    run<'a', 0, 4, 4>(sink, cyc, "AGPR, + 4 v_add_u32 per item, no reads");
    run<'a', 1, 2, 8>(sink, cyc, "AGPR, + 1 ds_read + 2 v_add per item, 4 idle partner waves (barrier only)");
    run<'v', 1, 2, 8>(sink, cyc, "VGPR, + 1 ds_read + 2 v_add per item, 4 idle partner waves (barrier only)");
    return 0;
}
