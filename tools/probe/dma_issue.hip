// Probe (round 3): what does ONE LDS-DMA piece (1 KiB per wave-instruction) cost a wave that is issuing MFMAs back to back, by the form of the
// instruction and by who issues it?  The 4-wave / AGPR GEMM probe of round 2 lost a third of its MFMA rate to 16 pieces per 128 MFMAs; the halo
// conv needs ~6 pieces per 80 MFMAs and wave.
//   mode 0: no staging (MFMAs + one barrier per step)
//   mode 1: global_load_lds_dwordx4, per-lane 64-bit address (what the product kernels issue)
//   mode 2: buffer_load_dwordx4 ... offen lds, constant voffset (lane * 16), the piece address in soffset (scalar)
//   mode 3: buffer_load_dwordx4 ... offen lds, per-piece voffset (one v_add per piece)
//   mode 4: eight waves: waves 0-3 multiply (never issue a DMA), waves 4-7 issue every piece (mode 1 form)
//   mode 5: eight waves as mode 4 with the mode 2 form
// One workgroup per CU, one step = 80 x v_mfma_f32_16x16x32_f16 per multiplying wave, PIECES pieces per wave and step spread evenly, source
// L2-resident (a 64 KB window that every workgroup reads, like a weight tile) or distinct per workgroup (activations).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/dma_issue tools/probe/dma_issue.hip ; run: ./dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int NM = 80, NACC = 16;

template <int MODE, int PIECES, bool DISTINCT>
__global__ __launch_bounds__(MODE >= 4 ? 512 : 256, 1) void probe_kernel(const char* __restrict__ src, long src_bytes, int steps, float* sink, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = MODE >= 4 && w >= 4;
    const bool computes = !loader;
    const int wl = w & 3;
    const long win = DISTINCT ? (long)(blockIdx.x % 128) * 65536 : 0;
    const char* base = src + win;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 65536, 0x00020000);
    f32x4 acc[NACC];
    f16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (f16)(((lane * 7 + i * 13) % 31 - 15) * 0.03f); b[i] = (f16)(((lane * 11 + i * 5) % 29 - 14) * 0.02f); }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int voff0 = lane * 16;
    unsigned long long t0 = 0;
    __syncthreads();
    if (lane == 0) t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        char* lds = smem + (s & 1) * 32768 + wl * 8192;
        const int sbase = (s * 4096 + wl * 1024) & 32767;
        if (computes) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m % NACC], 0, 0, 0);
                if (MODE >= 1 && MODE <= 3 && PIECES > 0 && (m % (NM / (PIECES > 0 ? PIECES : 1))) == 0 && m / (NM / (PIECES > 0 ? PIECES : 1)) < PIECES) {
                    const int pc = m / (NM / (PIECES > 0 ? PIECES : 1));
                    __builtin_amdgcn_sched_barrier(0);
                    const int off = (sbase + pc * 4096) & 65535 & ~1023;
                    if (MODE == 1) __builtin_amdgcn_global_load_lds((gptr_t)(base + off + voff0), (lptr_t)(lds + (pc & 7) * 1024), 16, 0, 0);
                    if (MODE == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (pc & 7) * 1024), 16, voff0, off, 0, 0);
                    if (MODE == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (pc & 7) * 1024), 16, voff0 + off, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int pc = 0; pc < PIECES; ++pc) {
                const int off = (sbase + pc * 4096) & 65535 & ~1023;
                if (MODE == 4) __builtin_amdgcn_global_load_lds((gptr_t)(base + off + voff0), (lptr_t)(lds + (pc & 7) * 1024), 16, 0, 0);
                if (MODE == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(lds + (pc & 7) * 1024), 16, voff0, off, 0, 0);
                __builtin_amdgcn_s_sleep(2);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (lane == 0 && w == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (r == 123.456f) sink[tid] = r + ((float*)smem)[tid];
}

template <int MODE, int PIECES, bool DISTINCT>
static void run(const char* src, long bytes, float* sink, unsigned long long* cyc, const char* tag) {
    auto k = probe_kernel<MODE, PIECES, DISTINCT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int steps = 2000, threads = MODE >= 4 ? 512 : 256;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 65536, 0, src, bytes, 200, sink, cyc);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 65536, 0, src, bytes, steps, sink, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cps = (double)h[128] / steps;
    const double pf = 256.0 * 4 * NM * 16384.0 * steps / (ms * 1e-3) / 1e15;
    printf("%-58s pieces/wave/step %2d %s : %7.1f cycles/step (MFMA floor %d) | %.3f ms, %.2f PFLOP/s, clock %.2f GHz\n", tag, PIECES, DISTINCT ? "distinct" : "shared  ",
           cps, NM * 16, ms, pf, (double)h[128] / (ms * 1e-3) / 1e9);
}

int main() {
    const long bytes = 128L * 65536;
    char* src; float* sink; unsigned long long* cyc;
    hipMalloc(&src, bytes); hipMalloc(&sink, 4096); hipMalloc(&cyc, 256 * 8);
    std::vector<unsigned short> h(bytes / 2);
    unsigned s = 1; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3000 + ((s >> 10) & 0xfff)); }
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    run<0, 0, false>(src, bytes, sink, cyc, "0 no staging");
    run<1, 5, false>(src, bytes, sink, cyc, "1 global_load_lds (64-bit per-lane address)");
    run<2, 5, false>(src, bytes, sink, cyc, "2 buffer_load lds, const voffset + scalar soffset");
    run<3, 5, false>(src, bytes, sink, cyc, "3 buffer_load lds, per-piece voffset");
    run<4, 5, false>(src, bytes, sink, cyc, "4 loader waves 4-7 (global_load_lds), waves 0-3 multiply");
    run<5, 5, false>(src, bytes, sink, cyc, "5 loader waves 4-7 (buffer_load lds)");
    run<1, 8, false>(src, bytes, sink, cyc, "1 global_load_lds");
    run<2, 8, false>(src, bytes, sink, cyc, "2 buffer_load lds scalar");
    run<4, 8, false>(src, bytes, sink, cyc, "4 loader waves (global_load_lds)");
    run<5, 8, false>(src, bytes, sink, cyc, "5 loader waves (buffer_load lds)");
    run<1, 16, false>(src, bytes, sink, cyc, "1 global_load_lds");
    run<2, 16, false>(src, bytes, sink, cyc, "2 buffer_load lds scalar");
    run<4, 16, false>(src, bytes, sink, cyc, "4 loader waves (global_load_lds)");
    run<5, 16, false>(src, bytes, sink, cyc, "5 loader waves (buffer_load lds)");
    run<1, 8, true>(src, bytes, sink, cyc, "1 global_load_lds");
    run<2, 8, true>(src, bytes, sink, cyc, "2 buffer_load lds scalar");
    run<5, 8, true>(src, bytes, sink, cyc, "5 loader waves (buffer_load lds)");
    return 0;
}
