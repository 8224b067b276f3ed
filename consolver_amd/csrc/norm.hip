// HBM-bound glue of the denoiser: GroupNorm(+SiLU) over NHWC (two-source concat aware),
// LayerNorm over token rows.  All loads/stores are 16 bytes per lane; reductions are
// wave-shuffle + LDS, deterministic (no atomics).
#include "ops.h"

// 16-byte vectors per thread of gn_apply_kernel: 4 = exactly one batch of four loads per lane.  Same-box A/B of the UNet forward (f16x2): 32 -> 29.73 ms, 16 -> 29.54,
// 8 -> 29.52 / 30.60 (two boxes), 4 -> 30.45, 2 -> 30.42: the more workgroups, the better -- and a second batch's loads would queue behind the first batch's stores.
#ifndef GN_VPT
#define GN_VPT 4
#endif
#ifndef GN_COL
#define GN_COL 1
#endif

namespace {

// ---------------------------------------------------------------------------- GroupNorm
// pass 1: per (sample, row split) partial sums of channel PAIRS (2q, 2q+1) -- a group always holds whole pairs (C / groups is even) --
// into partial[b][s][C/2][2] (sum, sum of squares).  The same layout the conv / GEMM epilogues write when they produce the statistics of
// their own output (IgemmArgs::gn_stats).  blockDim = (C/8) * R with R rows in flight, so a thread's 8-channel vector index is fixed
// while it strides over rows.
__global__ void gn_stats_kernel(const f16* __restrict__ x, int C, int HW, float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [R][C/8][16]
    const int CV = C >> 3;
    const int cv = threadIdx.x % CV, rr = threadIdx.x / CV, R = blockDim.x / CV;
    const int s = blockIdx.x, b = blockIdx.y, S = gridDim.x;
    const int rows = HW / S, r0 = s * rows;
    float sum[8], sq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { sum[k] = 0.f; sq[k] = 0.f; }
    const f16* base = x + ((size_t)b * HW + r0) * C + cv * 8;
    int r = rr;
    for (; r + 3 * R < rows; r += 4 * R) {            // four independent 16-byte loads in flight per lane
        f16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f16x8*>(base + (size_t)(r + u * R) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float f = (float)v[u][k]; sum[k] += f; sq[k] += f * f; }
    }
    for (; r < rows; r += R) {
        const f16x8 v = *reinterpret_cast<const f16x8*>(base + (size_t)r * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; sum[k] += f; sq[k] += f * f; }
    }
    float* mine = red + ((size_t)rr * CV + cv) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) { mine[k] = sum[k]; mine[8 + k] = sq[k]; }
    __syncthreads();
    if (rr == 0) {
        for (int j = 1; j < R; ++j) {
            const float* o = red + ((size_t)j * CV + cv) * 16;
#pragma unroll
            for (int k = 0; k < 8; ++k) { sum[k] += o[k]; sq[k] += o[8 + k]; }
        }
        float* dst = partial + (((size_t)b * S + s) * (C >> 1) + cv * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { dst[2 * k] = sum[2 * k] + sum[2 * k + 1]; dst[2 * k + 1] = sq[2 * k] + sq[2 * k + 1]; }
    }
}

// pass 2: one workgroup per (sample, group): reduce the splits of both sources, group statistics, per-channel scale/shift.
// A group of the concatenated channel axis may straddle the two sources (960 = 640 + 320 channels: 30 per group), so the source is
// chosen per pair.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ p0, int S0, int c0, const float* __restrict__ p1, int S1, int c1,
                                                          int groups, int HW, float eps, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                          float* __restrict__ scale_shift) {
    __shared__ float red[2][4];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Ctot = c0 + c1, cpg = Ctot / groups, ppg = cpg >> 1, q0 = c0 >> 1, q1 = c1 >> 1;
    float a = 0.f, q = 0.f;
    const int Smax = S0 > S1 ? S0 : S1;
    for (int i = tid; i < Smax * ppg; i += 256) {            // (split, pair) items of this group
        const int s = i / ppg, pr = g * ppg + (i - s * ppg);
        if (pr < q0) { if (s < S0) { const float* p = p0 + (((size_t)b * S0 + s) * q0 + pr) * 2; a += p[0]; q += p[1]; } }
        else if (s < S1) { const float* p = p1 + (((size_t)b * S1 + s) * q1 + (pr - q0)) * 2; a += p[0]; q += p[1]; }
    }
    a = wave_sum(a); q = wave_sum(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = q; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float n = (float)cpg * (float)HW;
    const float mean = a / n;
    const float rstd = rsqrtf(fmaxf(q / n - mean * mean, 0.f) + eps);
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const float sc = (float)gamma[c] * rstd;
        scale_shift[((size_t)b * Ctot + c) * 2] = sc;
        scale_shift[((size_t)b * Ctot + c) * 2 + 1] = (float)beta[c] - mean * sc;
    }
}

// pass 3: y = [silu](x * scale + shift), two-source read, single [B][HW][Ctot] output.  SPLIT: the sources are split-fp16 residual-stream tensors
// (value = hi + lo, GroupNormArgs::x0_lo / x1_lo; a null lo plane reads as zero)
template <bool SILU, bool SPLIT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int c0, int c1,
                                                       int HW, const float* __restrict__ scale_shift, f16* __restrict__ out,
                                                       const f16* __restrict__ x0_lo, const f16* __restrict__ x1_lo) {
    extern __shared__ __attribute__((aligned(16))) float ss[];    // [Ctot][2]
    const int Ctot = c0 + c1, CV = Ctot >> 3;
    const int b = blockIdx.y;
    {   // the sample's (scale, shift) table -> LDS: 16-byte loads, four per lane issued together (clamped index, branch-free).  As a loop of 4-byte loads this was
        // load, s_waitcnt vmcnt(0), ds_write per element: 2 Ctot / 256 memory round trips one behind the other in front of the barrier (20 at Ctot = 2560)
        const f32x4* src = reinterpret_cast<const f32x4*>(scale_shift + (size_t)b * Ctot * 2);
        const int nv = Ctot >> 1;                                      // (Ctot is a multiple of 8)
        for (int i0 = threadIdx.x; i0 < nv; i0 += 4 * (int)blockDim.x) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * (int)blockDim.x; t[u] = src[i < nv ? i : nv - 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * (int)blockDim.x; if (i < nv) reinterpret_cast<f32x4*>(ss)[i] = t[u]; }
        }
    }
    __syncthreads();
    const int rows_per_blk = (HW + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rows_per_blk;
    const int r1 = min(HW, r0 + rows_per_blk);
    const int nvec = (r1 - r0) * CV;
    // (Round 4 tried the batches software-pipelined -- batch n + 1 loaded in front of batch n's stores, the first batch in front of the table's barrier: 124 VGPRs
    //  instead of 58-82, 4 waves per SIMD instead of 5-8, GroupNorm class 1.86 -> 2.21 ms per forward (profiles/r04_ab_gn_pipeline_unet.txt).  Occupancy hides this kernel's latency.)
    for (int i0 = threadIdx.x; i0 < nvec; i0 += 4 * (int)blockDim.x) {      // four independent 16-byte loads in flight per lane
        f16x8 v[4], vl[4]; int cc[4]; size_t oo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x;
            const int ic = i < nvec ? i : i0;                             // tail lanes re-read a valid vector (not stored)
            const int rl = ic / CV, c = (ic - rl * CV) * 8, r = r0 + rl;
            const size_t so = (c < c0) ? ((size_t)b * HW + r) * c0 + c : ((size_t)b * HW + r) * c1 + (c - c0);
            v[u] = *reinterpret_cast<const f16x8*>(((c < c0) ? x0 : x1) + so);
            if constexpr (SPLIT) {
                const f16* lo = (c < c0) ? x0_lo : x1_lo;
                vl[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (lo) vl[u] = *reinterpret_cast<const f16x8*>(lo + so);
            }
            cc[u] = c; oo[u] = ((size_t)b * HW + r) * Ctot + c;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * (int)blockDim.x >= nvec) break;
            f16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float xv = (float)v[u][k];
                if constexpr (SPLIT) xv += (float)vl[u][k];
                float y = xv * ss[2 * (cc[u] + k)] + ss[2 * (cc[u] + k) + 1];
                if (SILU) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
                o[k] = (f16)y;
            }
            *reinterpret_cast<f16x8*>(out + oo[u]) = o;
        }
    }
}

// pass 3, column form (round 4): a thread keeps ONE 8-channel column of the sample -- its 16 (scale, shift) values live in registers, loaded together with the data --
// and walks RPT rows of it: no LDS table, no barrier in front of the first load.  blockDim = CV * R threads (CV = Ctot / 8 columns, R rows side by side), a
// workgroup covers R * RPT rows of one sample.  Same arithmetic as gn_apply_kernel (bit-identical).  Same-box A/B of the UNet forward (f16x2): GroupNorm class
// 1.82 -> 1.72 ms, forward -0.15 ms; RPT 2 / 4 / 8: 1.75 / 1.72 / 1.83 ms.
template <bool SILU, bool SPLIT, int RPT>
__global__ __launch_bounds__(320) void gn_apply_col_kernel(const f16* __restrict__ x0, const f16* __restrict__ x1, int c0, int c1,
                                                           int HW, const float* __restrict__ scale_shift, f16* __restrict__ out,
                                                           const f16* __restrict__ x0_lo, const f16* __restrict__ x1_lo, int CV, int R, f16* __restrict__ out_lo = nullptr) {
    const int Ctot = c0 + c1;
    const int b = blockIdx.y;
    const int cv = threadIdx.x % CV, rr = threadIdx.x / CV;
    const int c = cv * 8;
    const int row0 = blockIdx.x * (R * RPT) + rr;
    const bool first = c < c0;
    const f16* __restrict__ src = first ? x0 : x1;
    const f16* __restrict__ slo = first ? x0_lo : x1_lo;
    const int cs = first ? c0 : c1, co = first ? c : c - c0;
    const f32x4* tp = reinterpret_cast<const f32x4*>(scale_shift + ((size_t)b * Ctot + c) * 2);
    const f32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];           // (scale, shift) of channels c .. c + 7
    f16x8 v[RPT], vl[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int r = row0 + u * R, rc = r < HW ? r : HW - 1;                // (rows past the end re-read the last row; nothing is stored for them)
        const size_t so = ((size_t)b * HW + rc) * cs + co;
        v[u] = *reinterpret_cast<const f16x8*>(src + so);
        if constexpr (SPLIT) {
            vl[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (slo) vl[u] = *reinterpret_cast<const f16x8*>(slo + so);
        }
    }
    const float sc[8] = {t0[0], t0[2], t1[0], t1[2], t2[0], t2[2], t3[0], t3[2]}, sh[8] = {t0[1], t0[3], t1[1], t1[3], t2[1], t2[3], t3[1], t3[3]};
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int r = row0 + u * R;
        if (r >= HW) break;
        f16x8 o;
        float yk[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float xv = (float)v[u][k];
            if constexpr (SPLIT) xv += (float)vl[u][k];
            float y = xv * sc[k] + sh[k];
            if (SILU) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * y));
            o[k] = (f16)y; yk[k] = y;
        }
        *reinterpret_cast<f16x8*>(out + ((size_t)b * HW + r) * Ctot + c) = o;
        if constexpr (SPLIT) {
            if (out_lo) {                                       // (GroupNormArgs::out_lo: the output head; uniform branch)
                f16x8 ol;
#pragma unroll
                for (int k = 0; k < 8; ++k) ol[k] = (f16)(yk[k] - (float)o[k]);
                *reinterpret_cast<f16x8*>(out_lo + ((size_t)b * HW + r) * Ctot + c) = ol;
            }
        }
    }
}

// ---------------------------------------------------------------------------- LayerNorm
// one wave per ROWS token rows, rows kept in registers (C <= 8 * 64 * MAXV), two-pass variance; the ROWS independent row loads keep
// ROWS x 16 bytes in flight per lane (at C = 320 only 40 of the 64 lanes carry data: one row per wave left the kernel latency-bound)
template <int MAXV, int ROWS, bool SPLIT = false>
__global__ __launch_bounds__(256) void ln_kernel(const f16* __restrict__ x, const f16* __restrict__ gamma, const f16* __restrict__ beta,
                                                 f16* __restrict__ out, int M, int C, float eps, const f16* __restrict__ x_lo = nullptr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + w) * ROWS;
    if (row0 >= M) return;
    const int CV = C >> 3;
    float v[ROWS][MAXV][8];
    float sum[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const int row = min(row0 + q, M - 1);                 // rows past the end re-read the last row (never stored)
        sum[q] = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int cv = lane + 64 * j;
            if (cv < CV) {
                const f16x8 t = *reinterpret_cast<const f16x8*>(x + (size_t)row * C + cv * 8);
                if constexpr (SPLIT) {                          // split-fp16 residual stream: value = hi + lo
                    const f16x8 t2 = *reinterpret_cast<const f16x8*>(x_lo + (size_t)row * C + cv * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[q][j][k] = (float)t[k] + (float)t2[k]; sum[q] += v[q][j][k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) { v[q][j][k] = (float)t[k]; sum[q] += v[q][j][k]; }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[q][j][k] = 0.f;
            }
        }
    }
    f16x8 gv[MAXV], bv[MAXV];
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (lane + 64 * j < CV) {
            gv[j] = *reinterpret_cast<const f16x8*>(gamma + (lane + 64 * j) * 8);
            bv[j] = *reinterpret_cast<const f16x8*>(beta + (lane + 64 * j) * 8);
        }
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        const float mean = wave_sum(sum[q]) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < MAXV; ++j)
            if (lane + 64 * j < CV) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = v[q][j][k] - mean; sq += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        if (row0 + q >= M) break;
#pragma unroll
        for (int j = 0; j < MAXV; ++j) {
            const int cv = lane + 64 * j;
            if (cv < CV) {
                f16x8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (f16)((v[q][j][k] - mean) * rstd * (float)gv[j][k] + (float)bv[j][k]);
                *reinterpret_cast<f16x8*>(out + (size_t)(row0 + q) * C + cv * 8) = o;
            }
        }
    }
}

}  // namespace

// standalone statistics of one [B][HW][C] tensor into partial[B][S][C/2][2]
static int launch_gn_stats(const f16* x, int B, int HW, int C, int S, float* partial, hipStream_t s) {
    const int CV = C / 8;
    int R = 256 / CV; if (R < 1) R = 1;
    const int T = CV * R;
    if (T > 1024) CS_FAIL(CS_E_SHAPE, "group_norm: C=%d too wide", C);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(S, B), dim3(T), (size_t)T * 16 * sizeof(float), s, x, C, HW, partial);
    return CS_OK;
}

int launch_gn_stats64(const f16* x, int B, int HW, int C, float* partial, hipStream_t s) {
    if (HW % 64 || C % 8) CS_FAIL(CS_E_SHAPE, "gn_stats64: HW=%d C=%d", HW, C);
    const int rc = launch_gn_stats(x, B, HW, C, HW / 64, partial, s);
    CS_CHECK_LAUNCH();
    return rc;
}

int launch_group_norm(const GroupNormArgs& a, hipStream_t s) {
    const int Ctot = a.c0 + a.c1;
    if (!a.x0 || !a.out || !a.partial || !a.gamma || !a.beta) CS_FAIL(CS_E_ARG, "group_norm: null pointer");
    if (a.c0 % 8 || a.c1 % 8 || Ctot % a.groups || (Ctot / a.groups) % 2) CS_FAIL(CS_E_SHAPE, "group_norm: channels (%d,%d) groups %d", a.c0, a.c1, a.groups);
    if (a.B <= 0 || a.HW <= 0) return a.B < 0 ? CS_E_SHAPE : CS_OK;
    const int smax = a.splits > 0 ? a.splits : GN_SPLITS;
    int S = smax;
    while (S > 1 && (a.HW % S)) S >>= 1;
    // workspace: [source 0 partials][source 1 partials][scale / shift]; a source whose producer already wrote its partial sums
    // (stats0 / stats1, S0 / S1 splits per sample) skips the statistics pass
    float* part0 = a.partial;
    float* part1 = a.partial + (size_t)a.B * smax * a.c0;
    float* scale_shift = a.partial + (size_t)a.B * smax * Ctot;
    const float* p0 = a.stats0; int S0 = a.stats0 ? a.S0 : S;
    const float* p1 = a.stats1; int S1 = a.stats1 ? a.S1 : S;
    if (!p0) { const int rc = launch_gn_stats(a.x0, a.B, a.HW, a.c0, S, part0, s); if (rc != CS_OK) return rc; p0 = part0; }
    if (a.c1 && !p1) { const int rc = launch_gn_stats(a.x1, a.B, a.HW, a.c1, S, part1, s); if (rc != CS_OK) return rc; p1 = part1; }
    if (!a.c1) { p1 = p0; S1 = 0; }
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.groups, a.B), dim3(256), 0, s,
                       p0, S0, a.c0, p1, S1, a.c1, a.groups, a.HW, a.eps, a.gamma, a.beta, scale_shift);
    {   // column form: wherever the columns of a row fit a workgroup (Ctot <= 2560)
        const int CV = Ctot / 8;
        if (GN_COL && CV <= 320) {
            const int R = 320 / CV;
            constexpr int RPT = 4;
            const dim3 grid((a.HW + R * RPT - 1) / (R * RPT), a.B), block(CV * R);
            const bool split = a.x0_lo || a.x1_lo;
            if (split) {
                if (a.silu) hipLaunchKernelGGL((gn_apply_col_kernel<true, true, RPT>), grid, block, 0, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, a.x0_lo, a.x1_lo, CV, R, a.out_lo);
                else hipLaunchKernelGGL((gn_apply_col_kernel<false, true, RPT>), grid, block, 0, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, a.x0_lo, a.x1_lo, CV, R, a.out_lo);
            } else {
                const f16* nul2 = nullptr;
                if (a.out_lo && hipMemsetAsync(a.out_lo, 0, (size_t)a.B * a.HW * Ctot * sizeof(f16), s) != hipSuccess) CS_FAIL(CS_E_HIP, "group_norm: memset of out_lo");
                if (a.silu) hipLaunchKernelGGL((gn_apply_col_kernel<true, false, RPT>), grid, block, 0, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, nul2, nul2, CV, R);
                else hipLaunchKernelGGL((gn_apply_col_kernel<false, false, RPT>), grid, block, 0, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, nul2, nul2, CV, R);
            }
            CS_CHECK_LAUNCH();
            return CS_OK;
        }
    }
    // (the general-shape kernel writes one plane: a requested lo plane is zeros there -- hi + 0 is what the consumer multiplies)
    if (a.out_lo && hipMemsetAsync(a.out_lo, 0, (size_t)a.B * a.HW * Ctot * sizeof(f16), s) != hipSuccess) CS_FAIL(CS_E_HIP, "group_norm: memset of out_lo");
    int chunks = (a.HW * (Ctot / 8) + 256 * GN_VPT - 1) / (256 * GN_VPT);      // ~GN_VPT vectors per thread
    if (chunks < 1) chunks = 1;
    if (chunks > a.HW) chunks = a.HW;
    const size_t lds = (size_t)2 * Ctot * sizeof(float);
    const f16* nul = nullptr;
    if (a.x0_lo || a.x1_lo) {                 // split-fp16 sources (the statistics above come from the hi planes: the lo planes move a group's mean / variance by < 1e-7 relative)
        if (a.silu) hipLaunchKernelGGL((gn_apply_kernel<true, true>), dim3(chunks, a.B), dim3(256), lds, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, a.x0_lo, a.x1_lo);
        else hipLaunchKernelGGL((gn_apply_kernel<false, true>), dim3(chunks, a.B), dim3(256), lds, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, a.x0_lo, a.x1_lo);
    } else if (a.silu) hipLaunchKernelGGL((gn_apply_kernel<true, false>), dim3(chunks, a.B), dim3(256), lds, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, nul, nul);
    else hipLaunchKernelGGL((gn_apply_kernel<false, false>), dim3(chunks, a.B), dim3(256), lds, s, a.x0, a.x1, a.c0, a.c1, a.HW, scale_shift, a.out, nul, nul);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_layer_norm(const f16* x, const f16* gamma, const f16* beta, f16* out, int M, int C, float eps, hipStream_t s, const f16* x_lo) {
    if (!x || !gamma || !beta || !out) CS_FAIL(CS_E_ARG, "layer_norm: null pointer");
    if (C % 8 || C > 8 * 64 * 4) CS_FAIL(CS_E_SHAPE, "layer_norm: C=%d unsupported", C);
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    const int nv = (C / 8 + 63) / 64;
    const dim3 block(256);
    if (x_lo) {
        if (nv <= 1) hipLaunchKernelGGL((ln_kernel<1, 4, true>), dim3((M + 15) / 16), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
        else if (nv == 2) hipLaunchKernelGGL((ln_kernel<2, 2, true>), dim3((M + 7) / 8), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
        else if (nv == 3) hipLaunchKernelGGL((ln_kernel<3, 1, true>), dim3((M + 3) / 4), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
        else hipLaunchKernelGGL((ln_kernel<4, 1, true>), dim3((M + 3) / 4), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    if (nv <= 1) hipLaunchKernelGGL((ln_kernel<1, 4>), dim3((M + 15) / 16), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
    else if (nv == 2) hipLaunchKernelGGL((ln_kernel<2, 2>), dim3((M + 7) / 8), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
    else if (nv == 3) hipLaunchKernelGGL((ln_kernel<3, 1>), dim3((M + 3) / 4), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
    else hipLaunchKernelGGL((ln_kernel<4, 1>), dim3((M + 3) / 4), block, 0, s, x, gamma, beta, out, M, C, eps, x_lo);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
