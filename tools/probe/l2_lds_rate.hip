// Probe: how many bytes per second can all CUs pull out of L2 into LDS (distinct 128-byte lines, the GEMM staging pattern: one wave-instruction = 8 rows x
// 128 B)?  Each workgroup streams 64 KB "stages" out of a region that is small enough to stay in its XCD's 4 MB L2 (or, with a big region, out of
// Infinity Cache / HBM), by LDS-DMA or by global_load_dwordx4 + ds_write_b128, with 1 or 2 stages in flight.  No arithmetic.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/l2_lds_rate tools/probe/l2_lds_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, void* lds) { __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds, 16, 0, 0); }

// region_bytes per XCD-group of workgroups (workgroup id % 8 = its XCD); row pitch 2 KB; a stage = 512 rows x one 128-byte column block
template <bool DMA>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ src, size_t region_bytes, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int xcd = blockIdx.x & 7;
    const char* base = src + (size_t)xcd * region_bytes;
    const size_t pitch = 2048;                                    // bytes per row (a K = 1024 fp16 matrix); region >= 1 MB + 512 rows
    const size_t rows_in_region = region_bytes / pitch;
    unsigned acc = 0;
    for (int s = 0; s < steps; ++s) {
        // stage s of this workgroup: 512 rows starting at a workgroup- and step-dependent row, 128-byte column block (s % 128)
        const size_t row0 = ((size_t)(blockIdx.x >> 3) * 512 + (size_t)s * 37 * 64) % (rows_in_region - 511);      // rows_in_region >= 1024
        const size_t col = (size_t)(s % 16) * 128;
        char* dst = smem + (s & 1) * 65536;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int piece = w + 8 * j;                           // 64 pieces of 8 rows
            const char* p = base + (row0 + piece * 8 + (lane >> 3)) * pitch + col + (lane & 7) * 16;
            if (DMA) glds16(p, dst + piece * 1024);
            else { const u32x4 v = *reinterpret_cast<const u32x4*>(p); *reinterpret_cast<u32x4*>(dst + piece * 1024 + lane * 16) = v; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<unsigned*>(smem + (s & 1) * 65536 + tid * 4);
    }
    if (acc == 0x12345678) sink[0] = acc;
}

int main() {
    const size_t total = (size_t)2 << 30;
    char* src; hipMalloc(&src, total); hipMemset(src, 1, total);
    unsigned* sink; hipMalloc(&sink, 4);
    printf("%-10s %12s %8s %10s %10s\n", "mode", "region/XCD", "steps", "us", "TB/s");
    for (int dma = 1; dma >= 0; --dma)
        for (size_t region : {(size_t)2 << 20, (size_t)3 << 20, (size_t)16 << 20, (size_t)256 << 20}) {
            const int steps = 200, grid = 256;
            auto k = dma ? stream_kernel<true> : stream_kernel<false>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
            for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, 0, src, region, steps, sink);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            const int reps = 5;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 131072, 0, src, region, steps, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double us = ms / reps * 1000.0, bytes = (double)grid * steps * 65536;
            printf("%-10s %9zu MB %8d %10.1f %10.2f\n", dma ? "lds-dma" : "reg+write", region >> 20, steps, us, bytes / us / 1e6);
        }
    return 0;
}
