// Probe: cycles per v_mfma_f32_16x16x32_f16 vs the legacy v_mfma_f32_16x16x16_f16 on gfx950 (one wave per SIMD, independent accumulators).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/probe/mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K32>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a8, b8; f16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(0.01f * (threadIdx.x % 7 + i)); b8[i] = (_Float16)(0.02f * (threadIdx.x % 5 + i)); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (K32) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc; const int blocks = 256, iters = 20000;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    long long h[256];
    for (int v = 0; v < 2; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
        printf("%s: %.2f shader cycles per MFMA (one wave per SIMD, 8 independent accumulators, every CU busy)\n",
               v ? "v_mfma_f32_16x16x32_f16" : "v_mfma_f32_16x16x16_f16", m / (8.0 * iters));
    }
    return 0;
}
