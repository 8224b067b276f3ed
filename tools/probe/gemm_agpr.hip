// Probe (round 2): can a source-level HIP kernel hold a 256 x 256 x 64 GEMM tile on FOUR waves, one per SIMD -- wave tile 128 x 128, 256 accumulator
// registers per lane -- if the accumulators are pinned to AGPRs by writing the MFMAs as inline asm with "+a" operands?  (gemm_w4_kernel with
// compiler-placed accumulators ran 3-4x slower than the 8-wave kernel: profiles/r02_ab_gemm_w4.txt.)  Variants: LDS-DMA staging / register
// staging (global_load_dwordx4 -> ds_write_b128).  C[M][N] = A[M][K] W[N][K]^T, fp16 in / out, fp32 accumulate; M, N % 256 == 0, K % 64 == 0.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/gemm_agpr tools/probe/gemm_agpr.hip ; run: ./gemm_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, void* lds) { __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds, 16, 0, 0); }

constexpr int BM = 256, BN = 256, BK = 64, MT = 8, NT = 8;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;

__device__ __forceinline__ void mfma_a(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int REGSTAGE, bool ASM>
__global__ __launch_bounds__(256, 1) void gemm_agpr_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = N / BN;
    int id;
    { const int nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nb >> 3, r = nb & 7; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int GM = 4, band = id / (GM * tiles_n), rem = id - band * (GM * tiles_n), gsz = min(GM, M / BM - band * GM);
    const int tn = rem / gsz, tm = band * GM + (rem - tn * gsz);
    const int m_blk = tm * BM, n_blk = tn * BN;
    const int KT = K / BK;

    const int pch = lane & 7, lrow = lane >> 3;
    // piece q (0..31) of a tile = rows 8q .. 8q+7; wave w stages pieces w + 4 j (j = 0..7) of both tiles; swizzle term (row >> 1) & 7 = (4 q + (lrow >> 1)) & 7
    const f16* a_src[8]; const f16* b_src[8]; int lds_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = w + 4 * j, r = 8 * q + lrow, sw = (pch ^ ((r >> 1) & 7)) * 8;
        a_src[j] = A + (size_t)(m_blk + r) * K + sw;
        b_src[j] = W + (size_t)(n_blk + r) * K + sw;
        lds_off[j] = q * 1024;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int swz = (lane >> 1) & 7;
    const int fo0 = (lane & 15) * 128 + (((lane >> 4)) ^ swz) * 16, fo1 = (lane & 15) * 128 + ((4 + (lane >> 4)) ^ swz) * 16;

    // stage 0
#pragma unroll
    for (int j = 0; j < 8; ++j) { glds16(a_src[j], smem + lds_off[j]); glds16(b_src[j], smem + A_BYTES + lds_off[j]); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");

    u32x4 st[16];
    if (REGSTAGE == 2 && KT > 1) {           // pipelined register staging: st[] holds stage kt + 1 at the top of step kt
#pragma unroll
        for (int q = 0; q < 16; ++q) st[q] = *reinterpret_cast<const u32x4*>(((q & 1) ? b_src[q >> 1] : a_src[q >> 1]) + (size_t)BK);
    }
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < KT;
        const char* ta = smem + buf * STAGE + (wm * 128) * 128;
        const char* tb = smem + buf * STAGE + A_BYTES + (wn * 128) * 128;
        char* nb = smem + (buf ^ 1) * STAGE;
        constexpr int NQ = 2 * NT;       // 16 items of 8 MFMAs
        auto wr = [&](int q) { const int ks = q / NT, i = q - ks * NT; return *reinterpret_cast<const f16x8*>(tb + i * 2048 + (ks ? fo1 : fo0)); };
        f16x8 fa[2][MT], fw[3];
#pragma unroll
        for (int j = 0; j < MT; ++j) fa[0][j] = *reinterpret_cast<const f16x8*>(ta + j * 2048 + fo0);
        fw[0] = wr(0); fw[1] = wr(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ks = q / NT, i = q - ks * NT;
            if (q + 2 < NQ) fw[(q + 2) % 3] = wr(q + 2);
            if (ks == 0) fa[1][i] = *reinterpret_cast<const f16x8*>(ta + i * 2048 + fo1);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                if (ASM) mfma_a(acc[i][j], fw[q % 3], fa[ks][j]);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[q % 3], fa[ks][j], acc[i][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (REGSTAGE == 2) {
                // item q: write piece q of stage kt + 1 (loaded during step kt - 1) into the other buffer, then reload the register with stage kt + 2
                const int j = q >> 1;
                // (unconditional: past the end the last stage is re-loaded and written into a buffer nobody reads again, so that the step stays one
                //  basic block and the compiler's vmcnt stays a COUNT -- with branches around them every write waited for vmcnt(0))
                *reinterpret_cast<u32x4*>(nb + ((q & 1) ? A_BYTES : 0) + lds_off[j] + lane * 16) = st[q];
                st[q] = *reinterpret_cast<const u32x4*>(((q & 1) ? b_src[j] : a_src[j]) + (size_t)(kt + 2 < KT ? kt + 2 : KT - 1) * BK);
            } else if (more) {                  // one piece pair position per item: item q carries piece (q >> 1) of A (even q) or W (odd q)
                const int j = q >> 1;
                const f16* src = ((q & 1) ? b_src[j] : a_src[j]) + (size_t)(kt + 1) * BK;
                if (REGSTAGE) st[q] = *reinterpret_cast<const u32x4*>(src);
                else glds16(src, nb + ((q & 1) ? A_BYTES : 0) + lds_off[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (REGSTAGE == 1 && more) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(nb + ((q & 1) ? A_BYTES : 0) + lds_off[q >> 1] + lane * 16) = st[q];
        }
        if (REGSTAGE != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    asm volatile("s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
    // plain epilogue: lane holds, per (i, j), row m = .. + j*16 + (lane & 15), channels n = .. + i*16 + 4*(lane>>4) .. +3
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m_blk + wm * 128 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n_blk + wn * 128 + i * 16 + 4 * (lane >> 4);
            f16x4 o = {(f16)acc[i][j][0], (f16)acc[i][j][1], (f16)acc[i][j][2], (f16)acc[i][j][3]};
            *reinterpret_cast<f16x4*>(C + (size_t)m * N + n) = o;
        }
    }
}

// Variant 3: the pipelined step of gemm_big_kernel<.,.,2> on this tile (LDS-DMA staging, AGPR accumulators): barrier at item 12 of 16 with all of the
// stage's fragments in registers (eight-slot weight-fragment ring); after it the released buffer is re-filled in place with stage kt + 2 (pieces
// 0..3) and the next step's first fragments are read under the last 32 MFMAs; the other 12 pieces of a stage go one per item into the next step.
template <int MODE>   // 0: LDS-DMA pieces; 1: no staging inside the loop (timing floor: MFMAs + fragment reads + barrier; results wrong);
                      // 2: register staging at the same positions: ds_write_b128 of the piece loaded one step earlier, then its register is re-loaded with the following stage
__global__ __launch_bounds__(256, 1) void gemm_agpr_pipe_kernel(const f16* __restrict__ A, const f16* __restrict__ W, f16* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = N / BN;
    int id;
    { const int nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nb >> 3, r = nb & 7; id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); }
    const int GM = 4, band = id / (GM * tiles_n), rem = id - band * (GM * tiles_n), gsz = min(GM, M / BM - band * GM);
    const int tn = rem / gsz, tm = band * GM + (rem - tn * gsz);
    const int m_blk = tm * BM, n_blk = tn * BN;
    const int KT = K / BK;
    const int pch = lane & 7, lrow = lane >> 3;
    const f16* a_src[8]; const f16* b_src[8]; int lds_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int q = w + 4 * j, r = 8 * q + lrow, sw = (pch ^ ((r >> 1) & 7)) * 8;
        a_src[j] = A + (size_t)(m_blk + r) * K + sw;
        b_src[j] = W + (size_t)(n_blk + r) * K + sw;
        lds_off[j] = q * 1024;
    }
    // piece n = 0..15 of a stage: even -> activations, odd -> weights
    auto piece = [&](int kt, int buf, int n) {
        const int j = n >> 1;
        glds16(((n & 1) ? b_src[j] : a_src[j]) + (size_t)kt * BK, smem + buf * STAGE + ((n & 1) ? A_BYTES : 0) + lds_off[j]);
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swz = (lane >> 1) & 7;
    const int fo0 = (lane & 15) * 128 + (((lane >> 4)) ^ swz) * 16, fo1 = (lane & 15) * 128 + ((4 + (lane >> 4)) ^ swz) * 16;
    constexpr int NQ = 2 * NT, QB = 12, NPOST = NQ - QB;
#pragma unroll
    for (int n = 0; n < 16; ++n) piece(0, 0, n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (KT > 1) {
#pragma unroll
        for (int n = 0; n < NPOST; ++n) piece(1, 1, n);
    }
    auto rd_w = [&](const char* tb, int q) { const int ks = q / NT, i = q - ks * NT; return *reinterpret_cast<const f16x8*>(tb + i * 2048 + (ks ? fo1 : fo0)); };
    f16x8 fa[2][MT], fw[8];
    {
        const char* ta0 = smem + (wm * 128) * 128; const char* tb0 = smem + A_BYTES + (wn * 128) * 128;
#pragma unroll
        for (int j = 0; j < MT; ++j) fa[0][j] = *reinterpret_cast<const f16x8*>(ta0 + j * 2048 + fo0);
        fw[0] = rd_w(tb0, 0); fw[1] = rd_w(tb0, 1);
    }
    // MODE 2: st[n] holds piece n of the stage that position n writes next: stage 2 for the post-barrier pieces 0..3, stage 1 for pieces 4..15
    auto psrc = [&](int kt, int n) { const int j = n >> 1; const int kc = kt < KT ? kt : KT - 1; return ((n & 1) ? b_src[j] : a_src[j]) + (size_t)kc * BK; };
    auto pdst = [&](int buf, int n) { return smem + buf * STAGE + ((n & 1) ? A_BYTES : 0) + lds_off[n >> 1] + lane * 16; };
    u32x4 st[16];
    if (MODE == 2) {
#pragma unroll
        for (int n = 0; n < 16; ++n) st[n] = *reinterpret_cast<const u32x4*>(psrc(n < NPOST ? 2 : 1, n));
    }
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        const char* ta = smem + buf * STAGE + (wm * 128) * 128;
        const char* tb = smem + buf * STAGE + A_BYTES + (wn * 128) * 128;
        const char* tan = smem + (buf ^ 1) * STAGE + (wm * 128) * 128;
        const char* tbn = smem + (buf ^ 1) * STAGE + A_BYTES + (wn * 128) * 128;
        const int k1 = kt + 1 < KT ? kt + 1 : KT - 1, k2 = kt + 2 < KT ? kt + 2 : KT - 1;      // (past the end: harmless re-loads, the step stays branch-free)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ks = q / NT, i = q - ks * NT;
            if (q == QB) {
                if (MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (register staging: the loads in flight are for LATER stages)
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q < QB - 1) fw[(q + 2) % 8] = rd_w(tb, q + 2);
            else if (q == QB - 1) {
#pragma unroll
                for (int r = q + 2; r < NQ; ++r) fw[r % 8] = rd_w(tb, r);
            } else if (q + 2 >= NQ) fw[(q + 2) % 8] = rd_w(tbn, q + 2 - NQ);
            if (ks == 0) fa[1][i] = *reinterpret_cast<const f16x8*>(ta + i * 2048 + fo1);
            if (q >= QB) {
#pragma unroll
                for (int t = 0; t < 2; ++t) fa[0][(q - QB) * 2 + t] = *reinterpret_cast<const f16x8*>(tan + ((q - QB) * 2 + t) * 2048 + fo0);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) mfma_a(acc[i][j], fw[q % 8], fa[ks][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
                if (q >= QB) piece(k2, buf, q - QB);                   // refill of the buffer just released
                else piece(k1, buf ^ 1, NPOST + q);                    // the rest of the stage started after the previous step's barrier
            }
            if (MODE == 2) {
                // (loads as inline asm with a hand-counted vmcnt(15) instead of the compiler's waits -- it puts vmcnt(0) in front of the first write
                //  after the loop's back edge -- measured SLOWER: 1.05 vs 0.96 ms on 8192^3)
                const int n = q >= QB ? q - QB : NPOST + q;
                *reinterpret_cast<u32x4*>(pdst(q >= QB ? buf : (buf ^ 1), n)) = st[n];
                st[n] = *reinterpret_cast<const u32x4*>(psrc(q >= QB ? kt + 3 : kt + 2, n));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 7\n s_nop 7\n s_nop 7" ::: "memory");
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m_blk + wm * 128 + j * 16 + (lane & 15);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n_blk + wn * 128 + i * 16 + 4 * (lane >> 4);
            f16x4 o = {(f16)acc[i][j][0], (f16)acc[i][j][1], (f16)acc[i][j][2], (f16)acc[i][j][3]};
            *reinterpret_cast<f16x4*>(C + (size_t)m * N + n) = o;
        }
    }
}

template <int MODE>
static float run_pipe(const f16* A, const f16* W, f16* C, int M, int N, int K, int reps) {
    auto gemm_agpr_pipe_kernel = ::gemm_agpr_pipe_kernel<MODE>;
    hipFuncSetAttribute((const void*)gemm_agpr_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    const int grid = (M / BM) * (N / BN);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm_agpr_pipe_kernel, dim3(grid), dim3(256), 2 * STAGE, 0, A, W, C, M, N, K);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_agpr_pipe_kernel, dim3(grid), dim3(256), 2 * STAGE, 0, A, W, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int REGSTAGE, bool ASM>
static float run(const f16* A, const f16* W, f16* C, int M, int N, int K, int reps) {
    auto k = gemm_agpr_kernel<REGSTAGE, ASM>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
    const int grid = (M / BM) * (N / BN);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 2 * STAGE, 0, A, W, C, M, N, K);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 2 * STAGE, 0, A, W, C, M, N, K);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int shapes[][3] = {{8192, 8192, 8192}, {131072, 256, 2880}, {32768, 512, 5760}, {8192, 3840, 1280}, {8192, 10240, 1280}, {32768, 5120, 640}, {131072, 2560, 320}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 9) % 2001 - 1000) / 1000.0f; };
        for (auto& v : hA) v = (f16)rnd();
        for (auto& v : hW) v = (f16)(rnd() * 0.05f);
        f16 *A, *W, *C;
        hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
        hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
        const double fl = 2.0 * M * N * K;
        float t[5];
        t[0] = run<0, false>(A, W, C, M, N, K, 5);
        t[1] = run<0, true>(A, W, C, M, N, K, 5);
        t[2] = run<1, false>(A, W, C, M, N, K, 5);
        t[3] = run<1, true>(A, W, C, M, N, K, 5);
        t[4] = run<2, true>(A, W, C, M, N, K, 5);
        const float tn = run_pipe<1>(A, W, C, M, N, K, 5);
        const float tp = run_pipe<0>(A, W, C, M, N, K, 5);
        const float tr = run_pipe<2>(A, W, C, M, N, K, 5);
        // check the last variant on a few entries
        std::vector<f16> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
        double maxrel = 0;
        for (int c = 0; c < 64; ++c) {
            const int m = (c * 7919) % M, n = (c * 104729) % N;
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)(float)hA[(size_t)m * K + k] * (double)(float)hW[(size_t)n * K + k];
            const double got = (double)(float)hC[(size_t)m * N + n];
            maxrel = fmax(maxrel, fabs(got - ref) / (fabs(ref) + 0.05));
        }
        printf("M=%6d N=%6d K=%5d  dma/builtin %.3f ms (%.2f PF) | dma/asm-agpr %.3f (%.2f) | regstage/builtin %.3f (%.2f) | regstage/asm-agpr %.3f (%.2f) | pipelined regstage/asm-agpr %.3f (%.2f) | pipelined-barrier dma/asm-agpr %.3f (%.2f), without staging %.3f (%.2f), register staging %.3f (%.2f) | check (last variant) max rel %.2e\n",
               M, N, K, t[0], fl / t[0] / 1e12, t[1], fl / t[1] / 1e12, t[2], fl / t[2] / 1e12, t[3], fl / t[3] / 1e12, t[4], fl / t[4] / 1e12, tp, fl / tp / 1e12, tn, fl / tn / 1e12, tr, fl / tr / 1e12, maxrel);
        hipFree(A); hipFree(W); hipFree(C);
    }
    return 0;
}
