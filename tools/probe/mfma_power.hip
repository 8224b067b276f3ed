// Probe (round 3): the sustained runs are power-limited -- does the MFMA SHAPE change what the board sustains?  The same FLOPs per wave as back-to-back
// v_mfma_f32_16x16x32_f16 (A / B operand registers read once per 16 K MACs... 8192 MACs per instruction) and as v_mfma_f32_32x32x16_f16 (16384 MACs per instruction:
// half the operand-register reads per FLOP), random fp16 operands, 160 accumulator registers per wave, one or two waves per SIMD, ~50 ms per measurement so that the
// power management has settled.  Also: zero operands (the data-dependent part of the power), and the 16x16x16 form (half the MACs per instruction).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/mfma_power tools/probe/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int WPS>
__global__ __launch_bounds__(WPS * 256) void k(int iters, float scale, float* sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[10];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j][i] = (f16)(scale * (((lane * 7 + i * 13 + j * 3) % 31 - 15) * 0.03f));
#pragma unroll
        for (int j = 0; j < 10; ++j) b[j][i] = (f16)(scale * (((lane * 11 + i * 5 + j * 7) % 29 - 14) * 0.02f));
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    float r = 0.f;
    if constexpr (SHAPE == 0) {                  // 16x16x32: 4 x 10 tiles, 40 MFMAs per k32 step
        f32x4 acc[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 40; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[i % 10]), "v"(a[i / 10]));
        }
#pragma unroll
        for (int i = 0; i < 40; ++i) r += acc[i][0] + acc[i][3];
    } else if constexpr (SHAPE == 1) {           // 32x32x16: 2 x 5 tiles, 10 MFMAs per k16 step = the same FLOPs as 20 of the above
        f32x16 acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 10; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b[(i + rep) % 10]), "v"(a[(i / 5 + rep) % 4]));
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) r += acc[i][0] + acc[i][15];
    } else {                                      // 16x16x16: half the MACs per instruction
        f32x4 acc[40];
#pragma unroll
        for (int i = 0; i < 40; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 40; ++i) {
                const f16x4 a4 = {a[i / 10][0], a[i / 10][1], a[i / 10][2], a[i / 10][3]}, b4 = {b[i % 10][0], b[i % 10][1], b[i % 10][2], b[i % 10][3]};
                asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b4), "v"(a4));
            }
        }
#pragma unroll
        for (int i = 0; i < 40; ++i) r += acc[i][0] + acc[i][3];
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (lane == 0 && (threadIdx.x >> 6) == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (r == 123.456f) sink[threadIdx.x] = r;
}

template <int SHAPE, int WPS>
static void run(float scale, float* sink, unsigned long long* cyc, const char* tag) {
    const int iters = WPS == 1 ? 60000 : 30000;
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256), dim3(WPS * 256), 0, 0, 2000, scale, sink, cyc);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, WPS>), dim3(256), dim3(WPS * 256), 0, 0, iters, scale, sink, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double macs_per_iter = SHAPE == 2 ? 40.0 * 16 * 16 * 16 : 40.0 * 16 * 16 * 32;       // per wave (SHAPE 1: 20 x 32 x 32 x 16 = the same)
    const double flops = 2.0 * macs_per_iter * iters * 256.0 * 4 * WPS;
    printf("%-44s %d wave(s) / SIMD, %s operands: %7.1f ms  %6.0f TFLOP/s  clock %.2f GHz  (%.1f cycles per 16x16x32-equivalent MFMA and wave)\n", tag, WPS,
           scale == 0.f ? "zero  " : "random", ms, flops / (ms * 1e-3) / 1e12, (double)h[128] / (ms * 1e-3) / 1e9, (double)h[128] / iters / 40.0);
}

int main() {
    float* sink; unsigned long long* cyc;
    (void)hipMalloc(&sink, 8192); (void)hipMalloc(&cyc, 256 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>(1.f, sink, cyc, "v_mfma_f32_16x16x32_f16");
        run<1, 1>(1.f, sink, cyc, "v_mfma_f32_32x32x16_f16");
        run<0, 2>(1.f, sink, cyc, "v_mfma_f32_16x16x32_f16");
        run<1, 2>(1.f, sink, cyc, "v_mfma_f32_32x32x16_f16");
    }
    run<2, 1>(1.f, sink, cyc, "v_mfma_f32_16x16x16_f16 (half the MACs)");
    run<0, 1>(0.f, sink, cyc, "v_mfma_f32_16x16x32_f16");
    run<1, 1>(0.f, sink, cyc, "v_mfma_f32_32x32x16_f16");
    return 0;
}
