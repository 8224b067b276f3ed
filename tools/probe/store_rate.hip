// Probe: what bounds the GEMM epilogue's stores?  Emulates gemm_big_kernel's phase-2 store pattern (8 waves, each streaming a
// 64-row x 160-column fp16 sub-tile of a 256 x 320 tile as 16-byte lanes) with no compute, and varies
//   * the number of workgroups (= CUs storing at the same time; one workgroup per CU is forced with 140 KB of LDS),
//   * plain / nontemporal stores,
//   * 320-byte row segments per wave (the shipped split) or whole 640-byte rows per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_rate tools/probe/store_rate.hip ; run: ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // bit 0: nontemporal, bit 1: whole rows per wave
__global__ __launch_bounds__(512) void store_kernel(f16* out, int M, int N, int tiles_n, int tiles_per_wg) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    f16x8 v;
    for (int r = 0; r < 8; ++r) v[r] = (f16)(float)(lane + r);
    if (smem[tid] == 77) v[0] = (f16)3.f;      // keep the LDS allocation alive
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int id = blockIdx.x + t * gridDim.x;
        const int tm = id / tiles_n, tn = id - tm * tiles_n;
        if (tm * 256 >= M) return;
        if (!(MODE & 2)) {
            const int wm = w >> 1, wn = w & 1;
            for (int k = 0; k < 20; ++k) {
                const int idx = lane + 64 * k, row = idx / 20, ch = idx - row * 20;
                f16* dst = out + (size_t)(tm * 256 + wm * 64 + row) * N + tn * 320 + wn * 160 + ch * 8;
                if (MODE & 1) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(dst));
                else *reinterpret_cast<f16x8*>(dst) = v;
            }
        } else {
            for (int k = 0; k < 20; ++k) {
                const int idx = lane + 64 * k, row = idx / 40, ch = idx - row * 40;
                f16* dst = out + (size_t)(tm * 256 + w * 32 + row) * N + tn * 320 + ch * 8;
                if (MODE & 1) __builtin_nontemporal_store(v, reinterpret_cast<f16x8*>(dst));
                else *reinterpret_cast<f16x8*>(dst) = v;
            }
        }
    }
}

template <int MODE>
static float run(f16* out, int M, int N, int grid, int tiles_per_wg) {
    const int tiles_n = N / 320;
    hipFuncSetAttribute((const void*)store_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(512), 140 * 1024, 0, out, M, N, tiles_n, tiles_per_wg);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(grid), dim3(512), 140 * 1024, 0, out, M, N, tiles_n, tiles_per_wg);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1000.f;
}

int main() {
    const int M = 131072;
    f16* out; hipMalloc(&out, (size_t)M * 2560 * 2);
    printf("%-8s %6s %6s %6s %10s %10s %12s\n", "mode", "N", "grid", "t/wg", "us", "TB/s", "us/tile/CU");
    const int Ns[] = {320, 960, 2560};
    for (int N : Ns) {
        const int tiles = (M / 256) * (N / 320);
        const int grids[] = {tiles, 256, 128, 64, 32};
        for (int gi = 0; gi < 5; ++gi) {
            const int grid = grids[gi];
            const int tpw = gi == 0 ? 1 : 6;                       // limited grids: six tiles per workgroup, one after another
            const int ntiles = gi == 0 ? tiles : grid * tpw;
            if (ntiles > tiles) continue;
            float us[4];
            us[0] = run<0>(out, M, N, grid, tpw); us[1] = run<1>(out, M, N, grid, tpw);
            us[2] = run<2>(out, M, N, grid, tpw); us[3] = run<3>(out, M, N, grid, tpw);
            const char* names[] = {"plain", "nt", "rows", "rows+nt"};
            for (int m = 0; m < 4; ++m) {
                const double bytes = (double)ntiles * 256 * 320 * 2;
                const double per = gi == 0 ? us[m] / ((double)tiles / 256) : us[m] / tpw;
                printf("%-8s %6d %6d %6d %10.1f %10.2f %12.2f\n", names[m], N, grid, tpw, us[m], bytes / us[m] / 1e6, per);
            }
        }
    }
    return 0;
}
