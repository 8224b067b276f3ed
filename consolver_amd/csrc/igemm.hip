// Implicit-GEMM convolution / GEMM for NHWC fp16 activations on gfx950 MFMA.
//
//   out[m][n] = sum_{tap, c} act[pixel(m, tap)][c] * w[n][tap*Cin + c]  (+ bias + temb + residual | GEGLU)
//
// One kernel serves conv3x3 (pad 1, stride 1/2, optional fused nearest-x2 upsample of the input),
// conv1x1 and nn.Linear (taps = 1), with an optional second activation source that is the
// channel-concatenated skip tensor (UNet up blocks).
//
// Tiling (MI355X-first, not a warp-shaped port):
//   * workgroup = 256 threads = 4 wave64, output tile 128 (pixels) x BN (channels), BN = 128 or 160
//     (160 divides the SD1.5 widths 320/960 that 128 does not), K step 64 halfs = one 128-byte row;
//   * both operands are staged global->LDS with `global_load_lds_dwordx4` (no VGPR round trip); the
//     im2col gather (halo, stride, upsample, concat, zero padding) happens in the per-lane SOURCE
//     address, out-of-image taps read a zero page;
//   * LDS rows are 128 B; the 16-byte chunk index is XOR-swizzled with (row>>1)&7 so that the
//     ds_read_b128 fragment reads of v_mfma_f32_16x16x32_f16 are bank-conflict free; because LDS-DMA
//     writes lane-linear, the swizzle is applied to the source address and again on the read;
//   * operands are swapped (weights = MFMA A, activations = MFMA B) so that each lane ends up with 4
//     consecutive output CHANNELS of one pixel -> 8-byte NHWC stores, bias/temb/residual/GEGLU fused;
//   * double-buffered LDS (64-72 KB -> 2 workgroups per CU), next tile's DMA is issued before the
//     current tile's MFMAs; blockIdx is remapped so that each XCD (private L2) owns a contiguous
//     run of tiles and the n-tiles of one pixel tile run back to back on it.
#include "ops.h"
#include <stdlib.h>
#include <type_traits>


namespace {

constexpr int BM = 128;
constexpr int BK = 64;

__device__ __attribute__((aligned(256))) unsigned g_zero_page[64];

struct IgemmParams {
    const f16* a0; const f16* a1; int c0, c1;
    int Hi, Wi, Ho, Wo, HoWo;
    int stride, upsample;
    int pad_lo;         // zero rows / columns before the image (1 = symmetric pad 1; 0 = the VAE encoder's (0,1,0,1) pad before its stride-2 conv)
    int M, N, KT, cpt;   // KT = K / 64 ; cpt = chunks (of 64 channels) per tap
    int Ktot;            // row length of w
    const f16* w; const f16* bias; const f16* temb; int temb_stride; const f16* res; f16* out;
    const f16* res_lo; f16* out_lo;   // split-fp16 residual stream (IgemmArgs::res_lo / out_lo) or null
    // split-fp16 A operand of a 1x1 / linear layer (IgemmArgs::a0_lo / a1_lo): k steps [0, KTh) multiply the hi planes a0 | a1, k steps [KTh, KT = 2 KTh) the lo
    // planes a0_lo | a1_lo against the SAME weight columns (the k offset into w wraps at KTh).  Without lo planes KTh == KT and nothing wraps.
    const f16* a0_lo; const f16* a1_lo; int KTh;
    // LayerNorm folded into the GEMM that consumes it (IgemmArgs::row_stats / ln_*):
    //   producer side: row_stats[M][N / COLS][2] = (sum, sum of squares) of every output row over each wave's COLS columns, from the fp32 values of the epilogue;
    //   consumer side: out = rstd_m (acc - mean_m ln_s[n]) + ln_b[n] with (mean, rstd) of row m from ln_stats[M][ln_groups][2] over ln_C channels.
    float* row_stats;
    const float* ln_stats; int ln_groups; float ln_inv_c, ln_eps; const float* ln_s; const float* ln_b;
    int tiles_n, nblk;
    float* partial;     // split-K scratch of the generic kernel ([splits][M][N] fp32) or null
    int debug;          // timing experiments only: bit0 skip epilogue, bit1 skip the k loop
    float* gn_stats;    // GroupNorm partial sums of the output, [M / 64][N / 2][2], written by the epilogue (IgemmArgs::gn_stats) or null
    int gm;             // gemm_big_kernel tile order: bands of gm tile rows walked column-major (1 = plain row-major)
    int pn;             // tile_of: 0 = contiguous run of tiles per XCD, > 0 = the XCDs as an (8 / pn) x pn grid over (row tiles, column tiles)
    int epi_fast;       // knob epi_fast: the FAST forms of the fp32-patch epilogue (igemm_epilogue_f32)
    int lo8;            // IgemmArgs::lo8: res_lo / out_lo are 8-bit (e5m2) planes, one byte per element
};

// The lo plane of a split residual-stream tensor as ONE BYTE per element (round 6): e5m2 -- sign, the fp16 exponent, two mantissa bits, i.e. the fp16 lo value
// rounded to its top byte; v_cvt_pk_bf8_f32 / v_cvt_pk_f32_bf8 convert two elements per instruction.  hi (11 significant bits) + lo8 (3) = 14 bits against 22 with
// an fp16 lo: tools/sim_precision_r06.py -- the transformer blocks' hidden state stored that way moves the per-forward eps error from 8.8224e-4 to 8.8209e-4 (nothing),
// EVERY stream tensor stored that way to 8.97e-4 (+1.6 %).  Half the lo plane's bytes through the epilogues' store-bound phase.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lo8_decode_add(float (&f)[8], const u32x2_t w) {
    const f32x2_t a = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[0], false), b = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[0], true);
    const f32x2_t c = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[1], false), d = __builtin_amdgcn_cvt_pk_f32_bf8((int)w[1], true);
    f[0] += a[0]; f[1] += a[1]; f[2] += b[0]; f[3] += b[1]; f[4] += c[0]; f[5] += c[1]; f[6] += d[0]; f[7] += d[1];
}
__device__ __forceinline__ u32x2_t lo8_encode(const float (&r)[8]) {
    int w0 = __builtin_amdgcn_cvt_pk_bf8_f32(r[0], r[1], 0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(r[2], r[3], w0, true);
    int w1 = __builtin_amdgcn_cvt_pk_bf8_f32(r[4], r[5], 0, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(r[6], r[7], w1, true);
    return u32x2_t{(unsigned)w0, (unsigned)w1};
}

// timing experiments only (debug bit 16384): per-workgroup wall-clock stamps of gemm_big_kernel, read back with cs_debug_trace_read
#define CS_TRACE_SLOTS 8192
#define CS_TRACE_W 12
__device__ unsigned long long g_trace[CS_TRACE_SLOTS * CS_TRACE_W];
__device__ __forceinline__ void trace_stamp(int debug, int slot, int which) {
    if ((debug & 16384) && threadIdx.x == 0 && slot < CS_TRACE_SLOTS) {
        g_trace[slot * CS_TRACE_W + which] = wall_clock64();
        if (which == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID
            const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);     // HW_REG_XCC_ID
            g_trace[slot * CS_TRACE_W + 5] = ((unsigned long long)xcc << 32) | hw;
        }
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (sum, sum of squares) of one row from its G partial pairs st[G][2], added in group order.  The pairs of a row are contiguous: G = 2 / 4 / 8 (C = 320 / 640 /
// 1280 behind 160-column wave tiles) come in as one / two / four 16-byte loads issued together.  As a loop of 8-byte loads hipcc waits for each load before it
// issues the next (load, s_waitcnt vmcnt(0), add): G memory round trips one behind the other in front of the tile's first barrier, 4 us at G = 8.
__device__ __forceinline__ void ln_row_moments(const float* __restrict__ st, int G, float& s1, float& s2) {
    s1 = 0.f; s2 = 0.f;
    if (G == 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(st), b = *reinterpret_cast<const f32x4*>(st + 4), c = *reinterpret_cast<const f32x4*>(st + 8),
                    d = *reinterpret_cast<const f32x4*>(st + 12);
        s1 += a[0]; s2 += a[1]; s1 += a[2]; s2 += a[3]; s1 += b[0]; s2 += b[1]; s1 += b[2]; s2 += b[3];
        s1 += c[0]; s2 += c[1]; s1 += c[2]; s2 += c[3]; s1 += d[0]; s2 += d[1]; s1 += d[2]; s2 += d[3];
    } else if (G == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(st), b = *reinterpret_cast<const f32x4*>(st + 4);
        s1 += a[0]; s2 += a[1]; s1 += a[2]; s2 += a[3]; s1 += b[0]; s2 += b[1]; s1 += b[2]; s2 += b[3];
    } else if (G == 2) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(st);
        s1 += a[0]; s2 += a[1]; s1 += a[2]; s2 += a[3];
    } else {
        for (int g = 0; g < G; ++g) { const f32x2 v = *reinterpret_cast<const f32x2*>(st + 2 * g); s1 += v[0]; s2 += v[1]; }
    }
}

// Folded-LayerNorm consumer, tile prologue (gemm_w8_kernel / gemm_lw_kernel, LNM == 1): everything the epilogue needs from global memory is fetched into a
// spare LDS region WHILE THE K LOOP RUNS -- per tile row (rstd, mean rstd) from the producer's partial sums, per weight row b' and s -- so that the epilogue
// itself issues no loads.  (With the loads in the epilogue the QKV GEMMs, which otherwise load nothing there, paid +4.7 us per tile: a dependent global load
// behind every CU's store burst.)  Layout: [rows][2] floats, then [tables][2][wc] floats (b' | s per column group of wc weight rows).
__device__ __forceinline__ void ln_tile_prologue(const IgemmParams& p, char* lnx, int t, int nthreads, int m_blk, int rows, int n_blk, int wc, int tables) {
    for (int r = t; r < rows; r += nthreads) {
        const int m = min(m_blk + r, p.M - 1);
        const float* st = p.ln_stats + (size_t)m * p.ln_groups * 2;
        float s1, s2;
        ln_row_moments(st, p.ln_groups, s1, s2);
        const float mean = s1 * p.ln_inv_c, rstd = __builtin_amdgcn_rsqf(fmaxf(__builtin_fmaf(-mean, mean, s2 * p.ln_inv_c), 0.f) + p.ln_eps);
        *reinterpret_cast<f32x2*>(lnx + r * 8) = f32x2{rstd, mean * rstd};
    }
    float* tab = reinterpret_cast<float*>(lnx + rows * 8);
    for (int c = t * 4; c < tables * wc; c += nthreads * 4) {
        const int tb = c / wc, cc = c - tb * wc;
        *reinterpret_cast<f32x4*>(tab + (tb * 2) * wc + cc) = *reinterpret_cast<const f32x4*>(p.ln_b + n_blk + c);
        *reinterpret_cast<f32x4*>(tab + (tb * 2 + 1) * wc + cc) = *reinterpret_cast<const f32x4*>(p.ln_s + n_blk + c);
    }
}

// The same for a plain bias (every other instantiation of those kernels): bias[n_blk .. n_blk + tables * wc) -> LDS [tables][wc] halfs behind the stages, so that
// the epilogue's per-16-column bias loads are LDS reads (the GEGLU epilogue measured 10-30 us faster per launch that way, profiles/r04_ab_lnfold_ops.txt).
__device__ __forceinline__ void bias_tile_prologue(const IgemmParams& p, char* tab, int t, int n_blk, int cols) {
    if (p.bias && t * 8 < cols) *reinterpret_cast<f16x8*>(tab + t * 16) = *reinterpret_cast<const f16x8*>(p.bias + n_blk + t * 8);
}

// Workgroup -> tile.  Dispatch deals workgroups to the 8 XCDs (private L2 each) round-robin, so XCD = bid & 7 and bid >> 3 is the slot inside it.
//   pn == 0: an XCD owns a contiguous run of tile ids (row-major; bands of gm tile rows walked column-major when gm > 1): every XCD streams ALL weight panels
//            and 1/8 of the activation rows through its L2 -- right when the activations dominate the bytes (the 64 x 64 / 32 x 32 levels);
//   pn  > 0: the XCDs form an (8 / pn) x pn grid over (row tiles, column tiles): an XCD streams 1/pn of the weights and pn/8 of the activations.  At the
//            16 x 16 / 8 x 8 levels the weights ARE the bytes (1280 x 1280 x 9: 29.5 MB against 5-21 MB of activations) and the contiguous order pulled them
//            into all eight L2s: 8.3x / 3.9x the algorithmic bytes (profiles/r04_pmc_traffic_by_kernel.txt).  The host picks pn (launch_igemm_impl).
__device__ __forceinline__ void tile_of(int bid, int nblk, int tiles_n, int gm, int pn, int& tm, int& tn) {
    const int xcd = bid & 7, slot = bid >> 3;
    if (pn > 0) {
        const int px = 8 / pn, sub_n = tiles_n / pn, sub_m = (nblk / tiles_n) / px;
        const int xm = xcd / pn, xn = xcd - xm * pn;
        const int lm = slot / sub_n, ln = slot - lm * sub_n;
        tm = xm * sub_m + lm; tn = xn * sub_n + ln;
        return;
    }
    const int q = nblk >> 3, r = nblk & 7;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    if (gm > 1) {
        const int tiles_m = nblk / tiles_n;
        const int band = id / (gm * tiles_n), rem = id - band * (gm * tiles_n);
        const int gsz = min(gm, tiles_m - band * gm);
        tn = rem / gsz; tm = band * gm + (rem - tn * gsz);
    } else { tm = id / tiles_n; tn = id - tm * tiles_n; }
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

// erf GELU (diffusers GEGLU uses F.gelu, approximate='none'): x Phi(x), Phi(x) = 1/2 + x Q(x^2) with Q a degree-9 polynomial
// fitted on |x| <= 4.5 (Chebyshev nodes) and x clamped to that range (Phi(+-4.5) = 1 - 3.4e-6 / 3.4e-6).  Max abs error of the
// GELU value 8.2e-5 over all x (at the clamp, where the true value is ~ -1.5e-5) -- an order below the fp16 rounding of the
// product that follows.  Two values per call so that the Horner chain is v_pk_fma_f32: ~7 VALU issue slots per value and no
// transcendental, against ~22 for the rcp + exp form (A&S 7.1.26) it replaces: the GEGLU epilogue evaluates this 168 M
// times per L0 layer and was VALU-bound.
// Two pairs per call, their Horner chains interleaved statement by statement: one chain alone is a string of DEPENDENT v_pk_fma_f32, and hipcc put an s_nop between
// every two of them (240 s_nop per wave in the GEGLU epilogue); with a second chain in between the dependent distance is two instructions and the nops go away.
__device__ __forceinline__ void gelu_erf4(f32x2 xa, f32x2 xb, f32x2& ga, f32x2& gb) {
    f32x2 ca, cb;
    ca[0] = __builtin_amdgcn_fmed3f(xa[0], -4.5f, 4.5f); ca[1] = __builtin_amdgcn_fmed3f(xa[1], -4.5f, 4.5f);
    cb[0] = __builtin_amdgcn_fmed3f(xb[0], -4.5f, 4.5f); cb[1] = __builtin_amdgcn_fmed3f(xb[1], -4.5f, 4.5f);
    const f32x2 ua = ca * ca, ub = cb * cb;
    f32x2 qa = f32x2{-2.092490677e-12f, -2.092490677e-12f}, qb = qa;
    qa = qa * ua + 2.374692942e-10f;  qb = qb * ub + 2.374692942e-10f;
    qa = qa * ua + -1.199396227e-08f; qb = qb * ub + -1.199396227e-08f;
    qa = qa * ua + 3.595010583e-07f;  qb = qb * ub + 3.595010583e-07f;
    qa = qa * ua + -7.229871699e-06f; qb = qb * ub + -7.229871699e-06f;
    qa = qa * ua + 1.050455248e-04f;  qb = qb * ub + 1.050455248e-04f;
    qa = qa * ua + -1.158194733e-03f; qb = qb * ub + -1.158194733e-03f;
    qa = qa * ua + 9.930972010e-03f;  qb = qb * ub + 9.930972010e-03f;
    qa = qa * ua + -6.646580249e-02f; qb = qb * ub + -6.646580249e-02f;
    qa = qa * ua + 3.989399076e-01f;  qb = qb * ub + 3.989399076e-01f;
    ga = xa * (ca * qa + 0.5f); gb = xb * (cb * qb + 0.5f);
}


// Epilogue.  After the MFMAs a lane holds, per (nt, mt) tile, pixel m = ..+(lane&15) and channels
// 4*(lane>>4)..+3: storing that directly means 8-byte lanes scattered over 16 rows per instruction and
// 2-3x more store (and residual-load) instructions than bytes justify -- the short-K layers were
// store-ISSUE bound.  Instead each wave converts (acc + bias [GEGLU]) to fp16 into its own LDS patch
// (the k-loop's stage buffers are free by now), then streams the patch out ROW-wise: 16 bytes per lane,
// whole 128/160-byte row segments per pixel, with the time-embedding and residual adds done on the way
// (fp16-rounded conv output + fp16 residual, i.e. the same two roundings torch's fp16 graph performs).
// row of the wave's 64-row patch -> output row m (or -1): consecutive rows for the GEMM-shaped kernels, the pixels of a
// 2-D image patch for the halo conv kernel
struct LinearRows {
    int m_base, M;
    __device__ __forceinline__ int operator()(int row) const { const int m = m_base + row; return m < M ? m : -1; }
    // index of the wave's 64-row block in the [B][HoWo / 64] statistics grid (rows are sample-major and HoWo % 64 == 0), or -1
    __device__ __forceinline__ int stat_slot(int) const { return m_base < M ? m_base >> 6 : -1; }
    __device__ __forceinline__ bool all_valid() const { return m_base + 64 <= M; }      // every one of the wave's 64 rows exists
};

// RSPLIT > 1 trades column groups for row groups: the patch holds 64 / RSPLIT rows x ALL the wave's columns, so a pass stores
// 320-byte (not 160-byte) row segments: whole 128-byte lines instead of halves of them.
// F32 (round 3, layers with a time-embedding or residual add): the patch holds acc + bias in FP32 and the adds happen in fp32 on the way out, so the
// output is rounded to fp16 ONCE (torch's fp16 graph -- and this epilogue until round 2 -- rounds the conv output and then the sum); half as many rows
// per pass.  Every residual-adding layer of the UNet paid that extra 2^-11 on the residual stream.
// LNM: 0 = no LayerNorm code at all, 1 = consumer of a folded LayerNorm (p.ln_stats set), 2 = producer of row statistics (p.row_stats set).  Separate kernel
// instantiations, NOT runtime branches: with both code paths compiled into gemm_w8_kernel<false> its allocation went from 240 to 256 VGPRs and every launch of
// the class paid for it, LayerNorm or not (+0.75 ms per UNet forward, same-box A/B profiles/r04_ab_ln_codegen.txt).
// FAST (round 4; F32 only, every row of the wave valid): 1 = residual, 2 = residual + its lo plane and a lo plane out, 3 = time embedding -- and nothing else.  The
// generic code loads what it adds where it adds it: load, s_waitcnt vmcnt(0), add -- and vmcnt counts the stores of the previous iteration too, so every one of a
// wave's 20 iterations waited a full memory round trip (two with a lo plane) plus the acknowledgement of its predecessor's stores: 28 us of the 71 us a
// 256 x 320 x 1280 tile took, 43 us in f16x2 mode (from the ISA: tools/README.md, round 4).  Here a pass issues ALL its loads in one go, branch-free, so that
// the compiler's own counted waits work: one round trip per pass.
// FAST 4 (round 6) = residual + its lo plane, NO lo plane out: the feed-forward's second linear in the split mode (its output has one consumer, proj_out's fp16 operand) --
// it ran the generic code since the hidden state after it stopped carrying a lo plane.  LO8: the lo planes are 8-bit (IgemmParams::lo8); FAST 2 / 4 take it as a template
// parameter, the generic code as a runtime branch.
template <bool GEGLU, int NT, int MT, int GROUP, class RowMap, int RSPLIT, bool F32, int LNM, int FAST = 0, bool LTAB = false, bool LO8 = false>
__device__ __forceinline__ void igemm_epilogue_impl(const IgemmParams& p, f32x4 (&acc)[NT][MT], const RowMap& rows, int n_base, int lane, char* wave_lds, char* ln_tab, const char* ln_rows) {
    static_assert(NT % GROUP == 0, "GROUP must divide NT");
    static_assert(MT % RSPLIT == 0, "RSPLIT must divide MT");
    static_assert(!(F32 && GEGLU), "the GEGLU layers add nothing after the product");
    constexpr int RT = MT / RSPLIT;                           // 16-row tiles per pass
    constexpr int COLS = GEGLU ? GROUP * 8 : GROUP * 16;      // output columns per pass
    constexpr int ROWB = F32 ? COLS * 4 + 16 : (COLS + 8) * 2;   // padded LDS row (bytes, multiple of 16)
    static_assert(RT * 16 * ROWB <= 11264, "the wave's patch");
    constexpr int CH = COLS / 8;                              // 8-column chunks per row (16 bytes of output each)
    const int g4 = (lane >> 4) * 4, i16 = lane & 15;
    const int Nout = GEGLU ? (p.N >> 1) : p.N;
    // GroupNorm statistics of the output (p.gn_stats): the final fp16 values of a pass are written back into the patch, and lanes
    // 0 .. COLS/4-1 then walk its rows with four columns each: v_dot2_f32_f16 against (1, 1) and against the value itself gives the sum and
    // the sum of squares of a channel PAIR per instruction (a group always holds whole pairs).
    const bool stats = !GEGLU && GROUP == NT && p.gn_stats != nullptr;
    float st_sum[2] = {0.f, 0.f}, st_sq[2] = {0.f, 0.f};
    // LayerNorm folded into this GEMM (p.ln_stats): the A operand was the RAW hidden state and the weights carry gamma, so the row's (mean, rstd) enter here:
    // value = rstd (acc - mean s[n]) + b'[n] = acc rstd + (b'[n] - (mean rstd) s[n]); the lane's MT rows are patch rows 16 j + (lane & 15)
    constexpr bool lnf = LNM == 1 && std::is_same<RowMap, LinearRows>::value;
    float ln_r[MT], ln_mr[MT];
    if constexpr (lnf && LTAB) {
        // the tile prologue (ln_tile_prologue) left (rstd, mean rstd) of the wave's rows and the b' / s table in LDS: no global load here
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(ln_rows + (j * 16 + i16) * 8);
            ln_r[j] = t[0]; ln_mr[j] = t[1];
        }
    } else if constexpr (lnf) {
        // (the generic tile, small shapes:) b' and s of the wave's NT * 16 weight rows through a wave-private LDS table (ln_tab, [2][NT * 16] floats), one round of
        // global loads together with the row statistics
        constexpr int WC = NT * 16;
        for (int c = lane * 4; c < WC; c += 256) {
            *reinterpret_cast<f32x4*>(ln_tab + c * 4) = *reinterpret_cast<const f32x4*>(p.ln_b + n_base + c);
            *reinterpret_cast<f32x4*>(ln_tab + (WC + c) * 4) = *reinterpret_cast<const f32x4*>(p.ln_s + n_base + c);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = rows(j * 16 + i16);
            float s1 = 0.f, s2 = 0.f;
            if (m >= 0) {
                const float* st = p.ln_stats + (size_t)m * p.ln_groups * 2;
                ln_row_moments(st, p.ln_groups, s1, s2);
            }
            const float mean = s1 * p.ln_inv_c;
            ln_r[j] = __builtin_amdgcn_rsqf(fmaxf(__builtin_fmaf(-mean, mean, s2 * p.ln_inv_c), 0.f) + p.ln_eps);
            ln_mr[j] = mean * ln_r[j];
        }
    }
    // row statistics of THIS layer's output for a LayerNorm folded into its consumer (p.row_stats; F32 path only): one (sum, sum of squares) per row and wave
    constexpr bool rstats = LNM == 2 && std::is_same<RowMap, LinearRows>::value && F32 && !GEGLU && GROUP == NT;
    // FAST: within a pass the loads run PD phase-2 iterations ahead of their use (iteration g = pass * NI + k): three iterations in flight, 24 registers with a
    // lo plane.  (All loads of a pass at once: 40 registers and spills; running ahead ACROSS the passes, the first loads of a pass under its phase 1: spills in the
    // row-statistics forms, and where it fitted the UNet forward did not move: 29.28 vs 29.29 ms.)  Branch-free, so that the compiler's own counted waits work.
    constexpr int NI = RT * 16 * CH / 64;                     // phase-2 iterations of a pass
    constexpr int NG = (NT / GROUP) * RSPLIT * NI;            // ... of the wave
    constexpr int PD = NI < 3 ? NI : 3;
    constexpr bool RLO = FAST == 2 || FAST == 4;              // the residual comes with a lo plane
    f16x8 pa[FAST ? PD : 1], pb[(RLO && !LO8) ? PD : 1];
    u32x2_t pb8[(RLO && LO8) ? PD : 1];
    auto prefetch = [&](int g) {
        if constexpr (FAST != 0) {
            static_assert(!FAST || (F32 && !GEGLU), "FAST is a form of the fp32-patch path");
            const int pass = g / NI, k = g - pass * NI, grp2 = pass / RSPLIT, rh2 = pass - grp2 * RSPLIT;
            const int n0p = n_base + grp2 * GROUP * 16;
            const int idx = lane + 64 * k;
            const int row = idx / CH, ch = idx - row * CH;
            const int m = rows(rh2 * RT * 16 + row);
            if constexpr (FAST == 3) pa[g % PD] = *reinterpret_cast<const f16x8*>(p.temb + (size_t)(m / p.HoWo) * p.temb_stride + n0p + ch * 8);
            else {
                const size_t off = (size_t)m * Nout + n0p + ch * 8;
                pa[g % PD] = *reinterpret_cast<const f16x8*>(p.res + off);
                if constexpr (RLO && !LO8) pb[g % PD] = *reinterpret_cast<const f16x8*>(p.res_lo + off);
                if constexpr (RLO && LO8) pb8[g % PD] = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(p.res_lo) + off);
            }
        }
    };
#pragma unroll
    for (int grp = 0; grp < NT / GROUP; ++grp)
#pragma unroll
    for (int rh = 0; rh < RSPLIT; ++rh) {
        // ---- phase 1: registers -> LDS patch [RT * 16 rows][COLS] fp16 (F32: fp32) --------------------------
        // LTAB: the bias values of the NEXT 16 columns are read from the LDS table while this step's are used (one wave per SIMD runs the epilogues of conv3_lw /
        // gemm_lw: nothing else hides an LDS round trip there; all of a pass's values at once cost 20 registers and spilled)
        constexpr int BSTEP = GEGLU ? 2 : 1;
        f16x4 bt_t = {0, 0, 0, 0}, bt_u = {0, 0, 0, 0};
        auto bias_read = [&](int ii2, f16x4& t2, f16x4& u2) {
            if constexpr (LTAB && !lnf) {
                const __attribute__((address_space(3))) char* lt = (const __attribute__((address_space(3))) char*)ln_tab;
                t2 = *reinterpret_cast<const __attribute__((address_space(3))) f16x4*>(lt + ((grp * GROUP + ii2) * 16 + g4) * 2);
                if (GEGLU) u2 = *reinterpret_cast<const __attribute__((address_space(3))) f16x4*>(lt + ((grp * GROUP + ii2) * 16 + g4 + 16) * 2);
            }
        };
        if constexpr (LTAB && !lnf) { if (p.bias) bias_read(0, bt_t, bt_u); }
        f32x4 ln_t = {0.f, 0.f, 0.f, 0.f}, ln_u = ln_t, ln_t2 = ln_t, ln_u2 = ln_t;
        auto ln_read = [&](int ii2) {
            if constexpr (LTAB && lnf) {
                constexpr int WC = NT * 16;
                const __attribute__((address_space(3))) char* lt = (const __attribute__((address_space(3))) char*)ln_tab;
                const int c = (grp * GROUP + ii2) * 16 + g4;
                ln_t = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lt + c * 4);
                ln_u = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lt + (WC + c) * 4);
                if (GEGLU) {
                    ln_t2 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lt + (c + 16) * 4);
                    ln_u2 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lt + (WC + c + 16) * 4);
                }
            }
        };
        if constexpr (LTAB && lnf) ln_read(0);
#pragma unroll
        for (int ii = 0; ii < GROUP; ii += (GEGLU ? 2 : 1)) {
            const int i = grp * GROUP + ii;
            const int n = n_base + i * 16 + g4;                // (GEGLU: row of the permuted weight, value half)
            float bv[4] = {0.f, 0.f, 0.f, 0.f}, bg[4] = {0.f, 0.f, 0.f, 0.f};
            float sv[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
            if (lnf) {                                          // folded LayerNorm: b' (fp32, holds the layer's own bias) and s = row sums of the folded weight
                f32x4 t, u, t2 = {0.f, 0.f, 0.f, 0.f}, u2 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (LTAB) {                           // (read one step ahead, like the bias values below)
                    t = ln_t; u = ln_u; t2 = ln_t2; u2 = ln_u2;
                    if (ii + BSTEP < GROUP) ln_read(ii + BSTEP);
                } else {
                    constexpr int WC = NT * 16;
                    const int c = i * 16 + g4;
                    t = *reinterpret_cast<const f32x4*>(ln_tab + c * 4); u = *reinterpret_cast<const f32x4*>(ln_tab + (WC + c) * 4);
                    if (GEGLU) { t2 = *reinterpret_cast<const f32x4*>(ln_tab + (c + 16) * 4); u2 = *reinterpret_cast<const f32x4*>(ln_tab + (WC + c + 16) * 4); }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { bv[r] = t[r]; sv[r] = u[r]; }
                if (GEGLU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bg[r] = t2[r]; sg[r] = u2[r]; }
                }
            } else if (p.bias) {
                // LTAB (a template parameter: gemm_w8 / gemm_lw / conv3_lw): the wave's bias values are in LDS (ln_tab), fetched under the k loop.  NOT a runtime
                // `ln_tab ? *lds : *global`: hipcc turns a select of pointers into flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) -- 180 of them per kernel, one in front
                // of every phase-1 LDS write, each also waiting for the previous pass's global stores to be acknowledged (it merges an if / else into the same).
                f16x4 t, u = {0, 0, 0, 0};
                if constexpr (LTAB) {
                    t = bt_t; u = bt_u;
                    if (ii + BSTEP < GROUP) bias_read(ii + BSTEP, bt_t, bt_u);
                } else {
                    t = *reinterpret_cast<const f16x4*>(p.bias + n);
                    if (GEGLU) u = *reinterpret_cast<const f16x4*>(p.bias + n + 16);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = (float)t[r];
                if (GEGLU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) bg[r] = (float)u[r];
                }
            }
#pragma unroll
            for (int jj = 0; jj < RT; ++jj) {
                const int j = rh * RT + jj;
                // val = folded LayerNorm ? acc rstd + (b' - mean rstd s) : acc + bias.  EXPLICIT fused multiply-adds: left to -ffp-contract the compiler fused some of the
                // unrolled (i, j) instances and not others, and a row's result depended on which 16-row tile of the wave it sat in (last-bit differences between the
                // same sample at two batch positions; tools/dbg_rows.py)
                f32x4 val;
                if (lnf) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] = __builtin_fmaf(acc[i][j][r], ln_r[j], __builtin_fmaf(-ln_mr[j], sv[r], bv[r]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) val[r] = acc[i][j][r] + bv[r];
                }
                if constexpr (F32) {
                    const f32x4 o = val;
                    *reinterpret_cast<f32x4*>(wave_lds + (jj * 16 + i16) * ROWB + (ii * 16 + g4) * 4) = o;
                } else {
                    f16x4 o;
                    if (GEGLU) {
                        const f32x4 ga = acc[i + 1 < NT ? i + 1 : i][j];
                        f32x4 gt;
                        if (lnf) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) gt[r] = __builtin_fmaf(ga[r], ln_r[j], __builtin_fmaf(-ln_mr[j], sg[r], bg[r]));
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) gt[r] = ga[r] + bg[r];
                        }
                        f32x2 g01, g23;
                        gelu_erf4(f32x2{gt[0], gt[1]}, f32x2{gt[2], gt[3]}, g01, g23);
                        o[0] = (f16)(val[0] * g01[0]); o[1] = (f16)(val[1] * g01[1]);
                        o[2] = (f16)(val[2] * g23[0]); o[3] = (f16)(val[3] * g23[1]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = (f16)val[r];
                    }
                    const int col = GEGLU ? (ii >> 1) * 16 + g4 : ii * 16 + g4;
                    *reinterpret_cast<f16x4*>(wave_lds + (jj * 16 + i16) * ROWB + col * 2) = o;
                }
            }
        }
        // ---- phase 2: LDS patch -> global, 16 bytes per lane, + temb + residual ------------------------------
        if constexpr (FAST != 0) {
            // (behind phase 1: the pass's accumulators are dead, their registers hold the loads)
#pragma unroll
            for (int k = 0; k < PD; ++k) prefetch((grp * RSPLIT + rh) * NI + k);
            __builtin_amdgcn_sched_barrier(0);                   // (all of them out before the first is waited for)
        }
        const int n0 = GEGLU ? ((n_base + grp * GROUP * 16) >> 1) : n_base + grp * GROUP * 16;
        static_assert((RT * 16 * CH) % 64 == 0, "patch items must fill whole wave passes");
#pragma unroll
        for (int k = 0; k < RT * 16 * CH / 64; ++k) {
            const int g = (grp * RSPLIT + rh) * NI + k;
            const int idx = lane + 64 * k;
            const int row = idx / CH, ch = idx - row * CH;
            const int m = rows(rh * RT * 16 + row);
            if constexpr (F32) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(wave_lds + row * ROWB + ch * 32), v1 = *reinterpret_cast<const f32x4*>(wave_lds + row * ROWB + ch * 32 + 16);
                f16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
                float rs1 = 0.f, rs2 = 0.f;
                if (FAST != 0 || m >= 0) {
                    const size_t off = (size_t)m * Nout + n0 + ch * 8;
                    float f[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    if constexpr (FAST != 0) {                // (the same additions in the same order as below, from the registers loaded in front of phase 1)
#pragma unroll
                        for (int r = 0; r < 8; ++r) f[r] += (float)pa[g % PD][r];
                        if constexpr (RLO && !LO8) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) f[r] += (float)pb[g % PD][r];
                        }
                        if constexpr (RLO && LO8) lo8_decode_add(f, pb8[g % PD]);
                        if (k + PD < NI) prefetch(g + PD);        // (into the registers just consumed)
                    } else {
                    if (p.temb) {
                        const f16x8 t = *reinterpret_cast<const f16x8*>(p.temb + (size_t)(m / p.HoWo) * p.temb_stride + n0 + ch * 8);
#pragma unroll
                        for (int r = 0; r < 8; ++r) f[r] += (float)t[r];
                    }
                    if (p.res) {
                        const f16x8 t = *reinterpret_cast<const f16x8*>(p.res + off);
#pragma unroll
                        for (int r = 0; r < 8; ++r) f[r] += (float)t[r];
                        if (p.res_lo) {                       // split-fp16 residual stream: value = hi + lo
                            if constexpr (LO8) {                // (the byte-plane family's kernels only ever see byte planes: launch_igemm_impl)
                                lo8_decode_add(f, *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(p.res_lo) + off));
                            } else {
                                const f16x8 t2 = *reinterpret_cast<const f16x8*>(p.res_lo + off);
#pragma unroll
                                for (int r = 0; r < 8; ++r) f[r] += (float)t2[r];
                            }
                        }
                    }
                    }
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (f16)f[r];
                    if (rstats) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { rs1 += f[r]; rs2 = __builtin_fmaf(f[r], f[r], rs2); }
                    }
                    *reinterpret_cast<f16x8*>(p.out + off) = o;
                    if (FAST == 2 || (FAST == 0 && p.out_lo)) {   // what the fp16 store dropped, as a second plane (exact subtraction, then one rounding: to fp16, or to e5m2 -- lo8)
                        float d8[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) d8[r] = f[r] - (float)o[r];
                        if constexpr (LO8) {
                            *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned char*>(p.out_lo) + off) = lo8_encode(d8);
                        } else {
                            f16x8 l;
#pragma unroll
                            for (int r = 0; r < 8; ++r) l[r] = (f16)d8[r];
                            *reinterpret_cast<f16x8*>(p.out_lo + off) = l;
                        }
                    }
                }
                // (the fp16 values of chunk ch go where the fp32 values of chunks ch / 2 were: every lane has read its fp32 slot by now -- LDS operations of a
                //  wave execute in order -- and later iterations read other rows)
                if (stats) *reinterpret_cast<f16x8*>(wave_lds + row * ROWB + ch * 16) = o;
                // (row statistics: the chunk's partial sums over the first bytes of the row's fp32 area, same argument; a row that straddles two iterations
                //  is read by the next one from byte 32 ch upwards and written here below byte 8 ch)
                if (rstats) *reinterpret_cast<f32x2*>(wave_lds + row * ROWB + ch * 8) = f32x2{rs1, rs2};
            } else {
                const f16x8 v = *reinterpret_cast<const f16x8*>(wave_lds + row * ROWB + ch * 16);
                if (m >= 0) {
                    const size_t off = (size_t)m * Nout + n0 + ch * 8;
                    *reinterpret_cast<f16x8*>(p.out + off) = v;
                } else if (stats) {
                    *reinterpret_cast<f16x8*>(wave_lds + row * ROWB + ch * 16) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
        }
        if constexpr (F32) {
            if (rstats && lane < RT * 16) {                      // one lane per patch row: add the row's CH partial sums, one 8-byte store per row
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int c2 = 0; c2 < CH / 2; ++c2) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(wave_lds + lane * ROWB + c2 * 16);
                    a1 += t[0] + t[2]; a2 += t[1] + t[3];
                }
                if (CH & 1) { const f32x2 t = *reinterpret_cast<const f32x2*>(wave_lds + lane * ROWB + (CH - 1) * 8); a1 += t[0]; a2 += t[1]; }
                const int m = rows(rh * RT * 16 + lane);
                if (m >= 0) *reinterpret_cast<f32x2*>(p.row_stats + ((size_t)m * (p.N / COLS) + n_base / COLS) * 2) = f32x2{a1, a2};
            }
        }
        if (stats && lane < COLS / 4) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll 8
            for (int r = 0; r < RT * 16; ++r) {
                union { u32x2 u; h2 h[2]; } v;
                v.u = *reinterpret_cast<const u32x2*>(wave_lds + r * ROWB + lane * 8);
                st_sum[0] = __builtin_amdgcn_fdot2(v.h[0], one, st_sum[0], false); st_sq[0] = __builtin_amdgcn_fdot2(v.h[0], v.h[0], st_sq[0], false);
                st_sum[1] = __builtin_amdgcn_fdot2(v.h[1], one, st_sum[1], false); st_sq[1] = __builtin_amdgcn_fdot2(v.h[1], v.h[1], st_sq[1], false);
            }
        }
    }
    if (stats && lane < COLS / 4) {
        const int slot = rows.stat_slot(p.HoWo);
        if (slot >= 0)
            *reinterpret_cast<f32x4*>(p.gn_stats + ((size_t)slot * (p.N >> 1) + (n_base >> 1) + lane * 2) * 2) = f32x4{st_sum[0], st_sq[0], st_sum[1], st_sq[1]};
    }
}

// the fp32-patch path: the FAST forms where the layer adds exactly what one of them covers (EFAST: the kernel wants them compiled), else the generic code.
// EPI (round 6) selects the FAMILY of forms a kernel instantiation carries -- a kernel template parameter like LNM, for the same reason: every form compiled into a
// kernel taxes its register allocation, used or not (with the byte-plane forms added to the round-5 kernels gemm_w8_kernel<false, 0> went from 248 to 256 VGPRs,
// <false, 2> and conv3_lw_kernel to 256 + scratch).  EPI 0: FAST 1 / 2 / 3 + the generic code, fp16 lo planes only -- the round-5 kernels, bit for bit.
// EPI 1: the transformer blocks' linear layers on a hidden state with BYTE lo planes (lo8): FAST 2 and FAST 4 in their byte forms + the generic code reading / writing
// bytes.  EPI 2: a residual with an fp16 lo plane and no lo plane out (the feed-forward's second linear when lo8 is off): FAST 4 + the generic code.
template <bool GEGLU, int NT, int MT, int GROUP, class RowMap, int RS2, int LNM, bool EFAST, int EPI = 0>
__device__ __forceinline__ void igemm_epilogue_f32(const IgemmParams& p, f32x4 (&acc)[NT][MT], const RowMap& rows, int n_base, int lane, char* wave_lds, char* ln_tab, const char* ln_rows) {
    if constexpr (EFAST && EPI != 0) {
        constexpr bool L8 = EPI == 1;
        if (rows.all_valid() && p.res && !p.temb && p.res_lo) {
            // (one FAST form per instantiation: the row-statistics producers -- to_out -- write both planes, the plain ones -- the feed-forward's second linear -- the hi plane)
            if constexpr (L8 && LNM == 2) {
                if (p.out_lo && (p.epi_fast & 1)) { igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 2, EFAST, true>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
            }
            if constexpr (LNM == 0) {
                if (!p.out_lo && (p.epi_fast & 2)) { igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 4, EFAST, L8>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
            }
        }
        igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 0, EFAST, L8>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
        return;
    } else if constexpr (EFAST) {
        if ((p.epi_fast & 1) && rows.all_valid()) {
            if (p.res && !p.temb) {
                if (p.res_lo && p.out_lo) { igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 2, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
                if (!p.res_lo && !p.out_lo) { igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 1, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
            } else if (p.temb && !p.res && !p.out_lo) { igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 3, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
        }
    }
    // (EPI 1 outside the EFAST kernels = igemm_kernel: the generic code on byte planes)
    igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RS2, true, LNM, 0, EFAST, EPI == 1>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
}

// EFAST: the kernel (gemm_w8 / gemm_lw / conv3_lw) hands its bias through an LDS table (LTAB) and wants the FAST forms of the fp32-patch path compiled
template <bool GEGLU, int NT, int MT, int GROUP, class RowMap, int RSPLIT = 1, int LNM = 0, bool EFAST = false, int EPI = 0>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x4 (&acc)[NT][MT], const RowMap& rows, int n_base, int lane, char* wave_lds, char* ln_tab = nullptr, const char* ln_rows = nullptr) {
    if constexpr (LNM == 2) {                           // row statistics come from the fp32 values: always the fp32-patch path
        igemm_epilogue_f32<GEGLU, NT, MT, GROUP, RowMap, RSPLIT * 2, 2, EFAST, EPI>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
    } else if constexpr (LNM == 1) {                    // a folded LayerNorm's consumer adds nothing after the product (launch_igemm_impl checks)
        igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RSPLIT, false, 1, 0, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
    } else if constexpr (EPI != 0) {                    // (these families always add something: launch_igemm_impl routes only such layers here)
        igemm_epilogue_f32<GEGLU, NT, MT, GROUP, RowMap, RSPLIT * 2, 0, EFAST, EPI>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
    } else {
        if constexpr (!GEGLU) {
            if (p.temb || p.res || p.out_lo) { igemm_epilogue_f32<GEGLU, NT, MT, GROUP, RowMap, RSPLIT * 2, 0, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows); return; }
        }
        igemm_epilogue_impl<GEGLU, NT, MT, GROUP, RowMap, RSPLIT, false, 0, 0, EFAST>(p, acc, rows, n_base, lane, wave_lds, ln_tab, ln_rows);
    }
}

// ------------------------------------------------------------------------------------------------
// Halo-resident 3x3 convolution (stride 1, pad 1): the generic implicit GEMM re-stages the input
// pixels of a tile once per tap (9x) and is bound by L2->LDS traffic, not by MFMA.  Here a workgroup
// owns a 2-D PATCH of 256 output pixels (16 x 16; whole 8-wide images, several per tile, when the image
// is smaller) x BN channels; per 64-channel chunk the (TH+2) x (TW+2) input HALO of the patch is
// staged into LDS ONCE (zero page for the padding ring) and all nine taps multiply out of it with
// shifted row addresses, while only the BN x 64 weight tile streams per tap.  A 16 x 16 patch needs
// 324 halo rows whatever the image size (64 x 4 row strips needed 396 and stopped working past W = 64),
// so the same kernel serves the UNet (8..64 wide) and the VAE decoder (64..512 wide).
//   * 8 wave64 (4 pixel groups x 2 channel groups), wave tile 64 x BN/2, one workgroup per CU;
//   * halo rows are 128 B, chunk index XOR-swizzled with (row & 7): conflict-free ds_read_b128 for
//     ANY run of 16 consecutive rows, which is what a shifted tap reads;
//   * k order is chunk-major / tap-minor; the next chunk's halo is prefetched in eight slices (taps
//     1..8) into the second halo buffer, the next tap's weights into the next of THREE weight buffers;
//   * wave stagger: see the comment at the loop.  The third weight buffer and the tap-1 start of the halo
//     prefetch exist for it: a buffer is only re-staged two barriers after its last fragment read.
// ------------------------------------------------------------------------------------------------
struct HaloParams {
    IgemmParams e;          // epilogue view (M, N, HoWo, bias, temb, res, out, tiles_n, nblk)
    const f16* x; const f16* w;
    int Cin, H, W, B, NC;   // INPUT geometry; NC = Cin / 64
    int Ho, Wo;             // output geometry (2 H x 2 W when the nearest-x2 upsample is fused)
    int tw_shift, trw_shift;   // patch: TW = 1 << tw_shift output columns, TRW = TH * TW = 1 << trw_shift pixels per image in a tile
    int PX, PP;             // patches per image row, patches per image (PP == 1: 256 / TRW whole images per tile)
    int HALO_W, HALO_IMG;   // halo row length (TW_in + 2) ; halo rows per image patch
    int NHALO, NQ;          // halo rows per tile ; DMA instructions (8 rows each) per halo
    int splits;             // split-K over channel chunks (gridDim.y); > 1 -> fp32 partials to `partial`
    float* partial;         // [splits][M][N]
    int sched;              // kernel schedule variant (template SCHED): 0 lock-step groups as in round 1; 1 + static priority for group B;
                            // 2 + group B's DMA issues among its MFMAs
};

constexpr int HALO_ROWS_MAX = 400;
#ifndef CS_HALO_NWB
#define CS_HALO_NWB 3
#endif

struct PatchRows {          // tile-local pixel -> output row
    int o_base, b0, y0, x0, B, Ho, Wo, tw_shift, trw_shift;
    __device__ __forceinline__ int operator()(int row) const {
        const int o = o_base + row;
        const int img = o >> trw_shift, rem = o & ((1 << trw_shift) - 1);
        const int fy = rem >> tw_shift, fx = rem & ((1 << tw_shift) - 1);
        const int b = b0 + img;
        return b < B ? (b * Ho + y0 + fy) * Wo + x0 + fx : -1;
    }
    __device__ __forceinline__ bool all_valid() const { return b0 + ((o_base + 63) >> trw_shift) < B; }
    // the wave's 64 pixels are a quarter of a 16 x 16 patch (or a whole 8 x 8 image / a part of a small image): any bijection of the
    // (patch, quarter) pairs of an image onto [0, HoWo / 64) serves as the block index
    __device__ __forceinline__ int stat_slot(int HoWo) const {
        const int img = o_base >> trw_shift, b = b0 + img;
        if (b >= B || trw_shift < 6) return -1;
        const int th_shift = trw_shift - tw_shift;
        const int patch = ((y0 >> th_shift) * (Wo >> tw_shift)) + (x0 >> tw_shift);
        const int sub = (o_base & ((1 << trw_shift) - 1)) >> 6;
        return b * (HoWo >> 6) + (patch << (trw_shift - 6)) + sub;
    }
};

// conv3_lw_kernel<.., SUB>: the tile's 256 pixels are INPUT pixels of a nearest-x2 upsample + 3x3 conv in its sub-pixel form; the tile computes output phase (oy, ox):
// input pixel (y, x) -> output pixel (2 y + oy, 2 x + ox).  (y0, x0) is the patch origin in the input image, Ho x Wo the output image.
struct SubRows {
    int o_base, b0, y0, x0, B, Ho, Wo, tw_shift, trw_shift, oy, ox, slot_off;
    __device__ __forceinline__ int operator()(int row) const {
        const int o = o_base + row;
        const int img = o >> trw_shift, rem = o & ((1 << trw_shift) - 1);
        const int fy = rem >> tw_shift, fx = rem & ((1 << tw_shift) - 1);
        const int b = b0 + img;
        return b < B ? (b * Ho + 2 * (y0 + fy) + oy) * Wo + 2 * (x0 + fx) + ox : -1;
    }
    __device__ __forceinline__ bool all_valid() const { return b0 + ((o_base + 63) >> trw_shift) < B; }
    // the wave's 64 pixels are one phase of a quarter input patch (or of a whole 8 x 8 input image): phase-major over the blocks of the input image
    __device__ __forceinline__ int stat_slot(int HoWo) const {
        const int img = o_base >> trw_shift, b = b0 + img;
        if (b >= B || trw_shift < 6) return -1;
        const int th_shift = trw_shift - tw_shift;
        const int patch = ((y0 >> th_shift) * ((Wo >> 1) >> tw_shift)) + (x0 >> tw_shift);
        const int sub = (o_base & ((1 << trw_shift) - 1)) >> 6;
        return b * (HoWo >> 6) + slot_off + (patch << (trw_shift - 6)) + sub;
    }
};

// KH = 2 (BN = 320): the inner step is HALF a tap (k = 32): wave tile 64 x 160 like the big GEMM (22 % fewer LDS fragment bytes per MFMA than
// 64 x 80, half the per-tile prologue / epilogue and half the halo traffic per FLOP); the weight tile of a step is [320 rows][32 k] = 64-byte
// rows (chunk index XOR (row >> 1) & 3: conflict-free ds_read_b128 under gfx950's lane grouping), the same 20 KB and 20 DMA instructions as
// the [160][64] tile of KH = 1, so buffers, rotation and the stagger are unchanged - a chunk is 18 steps instead of 9.
template <bool UP, int BN, int KH = 1, int SCHED = 2>
__global__ __launch_bounds__(512, 2) void conv3_halo_kernel(HaloParams p) {
    static_assert(KH == 1 || KH == 2, "k halves per tap");
    constexpr int NT = BN / 32, MT = 4;
    constexpr int WROW = 128 / KH;                          // bytes per weight row in LDS
    constexpr int NBQ = BN * WROW / 1024;                   // weight-tile DMA instructions (1 KiB each)
    constexpr int STEPS = 9 * KH;                           // inner steps per 64-channel chunk
    constexpr int A_BYTES = HALO_ROWS_MAX * 128, B_BYTES = BN * WROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lA = smem;                    // [2][A_BYTES]
    char* const lB = smem + 2 * A_BYTES;      // [3][B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    int tm, tn;
    tile_of(blockIdx.x, p.e.nblk, p.e.tiles_n, 1, p.e.pn, tm, tn);
    const int n_blk = tn * BN;
    // tile -> (first image, patch origin)
    int b0, y0, x0;
    if (p.PP == 1) { b0 = tm << (8 - p.trw_shift); y0 = 0; x0 = 0; }
    else {
        b0 = tm / p.PP;
        const int pr = tm - b0 * p.PP, py = pr / p.PX, px = pr - py * p.PX;
        y0 = py << (p.trw_shift - p.tw_shift); x0 = px << p.tw_shift;
    }
    const int iy_base = (UP ? (y0 >> 1) : y0) - 1, ix_base = (UP ? (x0 >> 1) : x0) - 1;    // input pixel of halo (0, 0)
    const int c_per = p.NC / p.splits, c_begin = blockIdx.y * c_per, c_end = c_begin + c_per;

    // ---- halo staging: slice s (0..7) of a chunk is DMA instruction q = s + 8 w of this wave (rows 8q .. 8q+7).  The
    // lane's halo row walks hr = 64 w + (lane>>3) + 8 s; its (image, y, x) coordinates advance incrementally
    // (HALO_W >= 10 > 8: at most one wrap per step), so no per-slice divisions and no per-slice register table.
    const int pch = lane & 7;
    int hi0, hy0, hx0;
    {
        const int hr = 64 * w + (lane >> 3);
        hi0 = hr / p.HALO_IMG; const int rem = hr - hi0 * p.HALO_IMG;
        hy0 = rem / p.HALO_W; hx0 = rem - hy0 * p.HALO_W;
    }
    const int halo_rows = p.HALO_IMG / p.HALO_W;
    const int a_sw = (pch ^ ((lane >> 3) & 7)) * 8;      // halo row index & 7 == (lane>>3): 8q is a multiple of 8
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + pch * 16;
    int hi = hi0, hy = hy0, hx = hx0;
    auto halo_pix = [&]() -> int {      // input pixel index of the current halo row, or -1 (padding / out of tile)
        const int b = b0 + hi, y = iy_base + hy, x = ix_base + hx;
        const bool ok = (b < p.B) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W) & (hi * p.HALO_IMG + hy * p.HALO_W + hx < p.NHALO);
        return ok ? (b * p.H + y) * p.W + x : -1;
    };
    auto halo_advance = [&]() {
        hx += 8;
        const bool wrapx = hx >= p.HALO_W;
        hx -= wrapx ? p.HALO_W : 0; hy += wrapx ? 1 : 0;
        const bool wrapy = hy >= halo_rows;
        hy = wrapy ? 0 : hy; hi += wrapy ? 1 : 0;
    };
    // weight tile: NBQ DMA instructions, wave w issues q = w, w + 8 (, w + 16 when that is < NBQ)
    const f16* b_src[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = w + 8 * j;
        if (KH == 1) {
            const int r = 8 * q + (lane >> 3);
            b_src[j] = p.w + (size_t)(n_blk + (q < NBQ ? r : 0)) * (9 * p.Cin) + (pch ^ ((r >> 1) & 7)) * 8;
        } else {                                            // 16 rows x 64 B per DMA instruction
            const int r = 16 * q + (lane >> 2);
            b_src[j] = p.w + (size_t)(n_blk + (q < NBQ ? r : 0)) * (9 * p.Cin) + ((lane & 3) ^ ((r >> 1) & 3)) * 8;
        }
    }

    auto stage_a = [&](int c, int sl, int buf) {         // one eighth of chunk c's halo
        if (sl + 8 * w < p.NQ) {
            const int pix = halo_pix();
            const uintptr_t real = (uintptr_t)(p.x + ((long)(pix < 0 ? 0 : pix) * p.Cin + c * BK + a_sw));
            const uintptr_t msk = (uintptr_t)0 - (uintptr_t)(pix >= 0);
            glds16((const void*)((real & msk) | ((uintptr_t)zero & ~msk)), lA + buf * A_BYTES + (sl + 8 * w) * 1024);
        }
    };
    auto stage_w = [&](int c, int st, int buf) {           // st: step inside the chunk (tap, or tap * 2 + k half)
        const size_t koff = (size_t)(st / KH) * p.Cin + c * BK + (st % KH) * 32;
        char* lb = lB + buf * B_BYTES;
        glds16(b_src[0] + koff, lb + w * 1024);
        glds16(b_src[1] + koff, lb + (w + 8) * 1024);
        if (NBQ > 16 && w + 16 < NBQ) glds16(b_src[2] + koff, lb + (w + 16) * 1024);
    };

    // ---- fragment addressing: output pixel -> (image offset, row in patch, column in patch) ---------------
    int fi[MT], fy[MT], fx[MT];          // (without the fused upsample only fi is kept: the halo row of tap (0, 0))
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int o = wm * 64 + j * 16 + (lane & 15);
        const int img = o >> p.trw_shift, rem = o & ((1 << p.trw_shift) - 1);
        fy[j] = rem >> p.tw_shift; fx[j] = rem & ((1 << p.tw_shift) - 1); fi[j] = img * p.HALO_IMG;
        if (!UP) fi[j] += fy[j] * p.HALO_W + fx[j];
    }
    const int g = lane >> 4;
    const int swz = (lane >> 1) & 7;
    const int wfrag0 = KH == 1 ? (lane & 15) * 128 + ((g) ^ swz) * 16 : (lane & 15) * 64 + (g ^ (swz & 3)) * 16;
    const int wfrag1 = (lane & 15) * 128 + ((4 + g) ^ swz) * 16;      // (KH = 1 only)

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int sl = 0; sl < 8; ++sl) { stage_a(c_begin, sl, 0); halo_advance(); }
    stage_w(c_begin, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Wave stagger: waves 0-3 (group A) and 4-7 (group B) share the four SIMDs pairwise.  Run in lock-step, all
    // eight waves read their fragments right after the barrier and all multiply afterwards, so the LDS pipe and
    // the MFMA pipe take turns (measured: the two times ADD).  Group B therefore multiplies tap t one iteration
    // late -- after the next barrier, while group A is reading tap t+1's fragments -- and reads its own
    // fragments while group A multiplies.  Same registers, same LDS traffic, same results.
    // Group B's fragment reads of iteration `it` may still be in the LDS queue when the barrier that ends `it` opens;
    // they have certainly returned when group B passes the NEXT barrier (its multiply in it+1 consumed them).  So a
    // buffer is re-staged no earlier than iteration it+2: weights rotate through three buffers, and the next chunk's
    // halo (into the buffer last read at tap 8 of the previous chunk) starts at tap 1, not tap 0.
    const bool groupB = w >= 4;
    f16x8 fa[2 / KH][MT], fw[2 / KH][NT];
    auto read_frags = [&](const char* ha, const char* tb, int st) {
        const int t = st / KH, kh = st % KH;       // tap, k half (KH = 2)
        const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            int hr;
            if (UP) {
                const int hyv = ((fy[j] + dy - 1) >> 1) + 1, hxv = ((fx[j] + dx - 1) >> 1) + 1;
                hr = fi[j] + hyv * p.HALO_W + hxv;
            } else {
                hr = fi[j] + dy * p.HALO_W + dx;
            }
            const int sw = (g ^ (hr & 7)) * 16;
            if (KH == 1) {
                fa[0][j] = *reinterpret_cast<const f16x8*>(ha + hr * 128 + sw);
                fa[2 / KH - 1][j] = *reinterpret_cast<const f16x8*>(ha + hr * 128 + (sw ^ 64));
            } else {
                fa[0][j] = *reinterpret_cast<const f16x8*>(ha + hr * 128 + (sw ^ (kh * 64)));
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            fw[0][i] = *reinterpret_cast<const f16x8*>(tb + i * (16 * WROW) + wfrag0);
            if (KH == 1) fw[2 / KH - 1][i] = *reinterpret_cast<const f16x8*>(tb + i * 2048 + wfrag1);
        }
    };
    auto multiply_part = [&](auto i0_tag, auto i1_tag) {      // weight tiles [i0, i1) of the wave tile
        constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
#pragma unroll
        for (int ks = 0; ks < 2 / KH; ++ks)
#pragma unroll
            for (int i = I0; i < I1; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[ks][i], fa[ks][j], acc[i][j], 0, 0, 0);
    };
    auto multiply = [&]() { multiply_part(std::integral_constant<int, 0>{}, std::integral_constant<int, NT>{}); };

    // Static priority for group B.  Without it the stagger barely pays: group A's MFMAs become ready a few hundred cycles into the step
    // (when its fragment reads return) and then share the SIMD's matrix pipe 1:1 with group B's, so B finishes its 40 MFMAs at the END of
    // the step instead of the middle, issues its fragment reads late, and both groups stand at the next barrier with B's reads still in
    // flight -- the pipe idles for an LDS latency every step (measured with the staging switched off: 238 us -> 194 us on the 64 x 64
    // 320 -> 320 layer once B wins the arbitration; pure MFMA issue of that loop: 177 us).  With priority B's MFMAs run first and
    // uncontested, its reads overlap A's MFMAs, and vice versa.
    // What bounds this loop was measured with throw-away builds that skip parts of it (profiles/r02_conv_bound.txt): pure MFMA issue
    // 177 us, + per-step barrier 184, + fragment reads 221, + staging 259 (round-1 schedule); no barrier at all 180.
    if (SCHED >= 1 && groupB) __builtin_amdgcn_s_setprio(2);
    int it = 0, ab = 0, wb = 0;
    for (int c = c_begin; c < c_end; ++c, ab ^= 1) {
        const char* ha = lA + ab * A_BYTES;
        hi = hi0; hy = hy0; hx = hx0;
#pragma unroll 1
        for (int t = 0; t < STEPS; ++t, ++it) {       // t: step inside the chunk (a tap, or half a tap when KH = 2)
            const int wnext = wb == (CS_HALO_NWB - 1) ? 0 : wb + 1;
            const bool do_w = t < STEPS - 1 || c + 1 < c_end;
            const bool do_a = t > 0 && t <= 8 && c + 1 < c_end;
            const int wc = t < STEPS - 1 ? c : c + 1, wt = t < STEPS - 1 ? t + 1 : 0;
            const char* tb = lB + wb * B_BYTES + (wn * (BN / 2)) * WROW;
            if (SCHED >= 2 && groupB) {
                // group B's DMA issues go BETWEEN its MFMAs: a 1 KiB LDS-DMA instruction costs the issuing wave 100-200 cycles, and issued
                // in front of the MFMAs (fine for group A, whose MFMAs wait for its fragment reads anyway) they left the matrix pipe
                // empty at the start of every step
                constexpr int C1 = NT / 3, C2 = 2 * NT / 3;
                if (it > 0) multiply_part(std::integral_constant<int, 0>{}, std::integral_constant<int, C1>{});
                __builtin_amdgcn_sched_barrier(0);
                if (do_w) stage_w(wc, wt, wnext);
                __builtin_amdgcn_sched_barrier(0);
                if (it > 0) multiply_part(std::integral_constant<int, C1>{}, std::integral_constant<int, C2>{});
                __builtin_amdgcn_sched_barrier(0);
                if (do_a) { stage_a(c + 1, t - 1, ab ^ 1); halo_advance(); }
                __builtin_amdgcn_sched_barrier(0);
                if (it > 0) multiply_part(std::integral_constant<int, C2>{}, std::integral_constant<int, NT>{});
                __builtin_amdgcn_sched_barrier(0);
            } else {
                if (do_w) stage_w(wc, wt, wnext);
                if (do_a) { stage_a(c + 1, t - 1, ab ^ 1); halo_advance(); }
                if (SCHED < 2 && groupB && it > 0) multiply();     // group B: step it-1, fragments read before the previous barrier
            }
            read_frags(ha, tb, t);
            if (!groupB) multiply();                           // group A: this step
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            wb = wnext;
        }
    }
    if (groupB) multiply();                        // drain: the last tap of group B
    const PatchRows rows{wm * 64, b0, y0, x0, p.B, p.Ho, p.Wo, p.tw_shift, p.trw_shift};
    if (p.splits == 1) {
        __syncthreads();                           // group B's last reads are consumed; the stage buffers become the epilogue patches
        igemm_epilogue<false, NT, MT, NT, PatchRows, KH>(p.e, acc, rows, n_blk + wn * (BN / 2), lane, smem + w * 11264);      // (no FAST forms: the BN = 320 instantiation spills already)
    } else {
        // split-K: raw fp32 partial sums, 16-byte stores (4 consecutive channels per lane)
        float* dst = p.partial + (size_t)blockIdx.y * p.e.M * p.e.N;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = rows(j * 16 + (lane & 15));
            if (m < 0) continue;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                *reinterpret_cast<f32x4*>(dst + (size_t)m * p.e.N + n_blk + wn * (BN / 2) + i * 16 + g * 4) = acc[i][j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Halo conv with DEDICATED LOADER WAVES (round 3).  What the 8-wave kernel above pays for staging is not bytes but ISSUE: one 1 KiB
// LDS-DMA instruction costs the wave that issues it 25-40 cycles of MFMA issue whatever its form (global_load_lds with per-lane
// 64-bit addresses, buffer_load ... lds with a scalar offset: tools/probe/dma_issue.hip, profiles/r03_probe_dma_issue.txt), and
// next to nothing when ANOTHER wave of the SIMD issues it (80 MFMAs + 1 barrier per step: 1369 cycles with loader waves against
// 1570-1670 with 5-8 pieces issued between the MFMAs; floor 1280).  So: 8 waves, waves 0-3 (one per SIMD) multiply and never
// touch global memory, waves 4-7 (their SIMD partners) do nothing but stage.
//   * tile 256 pixels (the 16 x 16 / multi-image patch of the kernel above) x BN channels, wave tile 64 pixels x BN channels,
//     k step = one tap x 64 channels = 4 * BN/16 * 2 MFMAs per wave (80 at BN = 160);
//   * LDS: two halo buffers (chunk c / c + 1) + THREE weight stages (the third removes the lgkmcnt(0) in front of the step barrier);
//   * compute waves: the pipelined step of gemm_big_kernel -- weight fragments through a ring LA items ahead, the step's one
//     barrier at item QB with every fragment of the stage in registers, the next step's first fragments read under the last MFMAs;
//   * loader waves: after barrier K(g-1) (all reads of stage g - 1 are done) they issue stage g + 1's weights into that buffer
//     and one slice of the next chunk's halo, wait for the weights with a COUNTED vmcnt (the halo slice may stay in flight for
//     another step), and join barrier K(g).  All their per-piece addresses are computed once per tile.
// ------------------------------------------------------------------------------------------------
// padding rows of conv3_lw_kernel: the chunk offset c * 128 B (c < LW_ZERO_CHUNKS) is added to every source, the zero source included, so the region spans
// LW_ZERO_CHUNKS chunks + one 128-byte row + slack.  launch_igemm_impl sends Cin > 64 * LW_ZERO_CHUNKS (SD1.5 / VAE: <= 2560) to the halo kernels.
constexpr int LW_ZERO_CHUNKS = 64;
__device__ __attribute__((aligned(256))) unsigned g_zero_region[(LW_ZERO_CHUNKS * 128 + 256) / 4];

// ---- hand-counted LDS reads for the one-wave-per-SIMD kernels: hipcc neither sees nor waits for these -----------------------------------
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
template <int OFF>
__device__ __forceinline__ void lds_read(f16x8& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit unsigned offset");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait(f16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
// in-place MFMA: with the builtin, hipcc gave every accumulator update of the fully unrolled 9-tap body a fresh register (dst != srcC) and spilled 880 bytes per lane
__device__ __forceinline__ void mfma_inplace(f32x4& c, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }
// conv3_lw_kernel: number of LDS reads issued AFTER the youngest read item q multiplies, counted up to and including item q's own reads (LDS returns in
// order, so s_waitcnt lgkmcnt(that) is exactly "my operands have landed").  Reads in front of item p's MFMAs, in order: one weight fragment (for item p + LA),
// then (p < MT) one k-half-1 activation fragment, then (p = QB, QB + 1) two next-step activation fragments.
// FAST addressing of conv3_lw_kernel: halo row (relative to the wave's first one) read by output row j of the wave under tap row dy
constexpr int lw_rowi(bool up, int j, int dy) { return up ? (j + dy + 1) / 2 : dy + j; }        // (((j + dy - 1) >> 1) + 1 == (j + dy + 1) / 2 for j + dy >= 0)
constexpr int lw_rowf(int fast, bool up, int j, int dy) { return fast == 2 ? 2 * j + dy : lw_rowi(up, j, dy); }     // (FAST 2: output tile j = image rows 2 j, 2 j + 1)
constexpr int lw_reads_of(int p, int MT, int QB) { return 1 + (p < MT ? 1 : 0) + ((p == QB || p == QB + 1) ? 2 : 0); }
constexpr int lw_wait_count(int q, int NQ, int LA, int MT, int QB) {
    // the weight fragment of item q was the FIRST read of item q - LA (mod NQ)
    int n = 0;
    for (int d = LA; d >= 0; --d) {                      // items q - LA .. q
        const int pidx = ((q - d) % NQ + NQ) % NQ;
        n += lw_reads_of(pidx, MT, QB);
    }
    int after_w = n - 1;                                  // everything after that first read
    // items 0 and 1 also need all four next-step activation fragments, the last of which is the LAST read of item QB + 1 (previous step)
    int after_a = after_w;
    if (q < 2) {
        int m = 0;
        for (int pp = QB + 2; pp < NQ; ++pp) m += lw_reads_of(pp, MT, QB);
        for (int pp = 0; pp <= q; ++pp) m += lw_reads_of(pp, MT, QB);
        after_a = m;
    }
    return after_a < after_w ? after_a : after_w;
}

// TRACE: a separate instantiation with in-kernel cycle stamps (cs_set_tuning("debug", 16384), tools/conv_lw_trace.py).  It must be a compile-time
// variant: s_memtime is a scalar-memory instruction and shares lgkmcnt with the LDS reads, so even a never-taken stamp makes hipcc wait lgkmcnt(0).
// FAST (no fused upsample, 16 x 16 patches: HALO_W = 18, 324 halo rows, 41 pieces): every LDS address of the compute waves is a per-tile register plus an
// IMMEDIATE, so a steady-state step issues no vector-ALU instruction at all.  A lone MFMA-issuing wave pays ~10 cycles of matrix-pipe time per VALU
// instruction placed in its stream (tools/probe/mfma_stream.hip, profiles/r03_probe_mfma_stream.txt: 2 v_add per 4-MFMA item = +420 cycles per 80-MFMA
// step) and the generic path's ~40 address instructions per step were 450 of its 1775 cycles.  What makes it possible:
//   * the halo rows are swizzled by their COLUMN in the halo (hx & 7) instead of their row index: the 16 lanes of a fragment read rows of one halo line, so
//     the swizzle term of lane fx under tap column dx is (fx + dx) & 7 -- three per-lane values for the whole tile, whatever the tap row or output row;
//     (conflict-free like the row-index form: within a ds_read_b128 lane group the (row parity, 16-byte slot) pairs stay distinct because HALO_W is even);
//   * taps and halo-buffer parity are compile-time (the chunk loop runs one of two 9-step bodies), the weight buffer of tap t is t % 3 (9 % 3 == 0), and
//     (output row j + tap row dy) * 18 * 128 + parity * 41 KiB < 64 Ki fits the 16-bit offset field of ds_read.
// FAST = 2: four whole 8 x 8 images per tile (the UNet's 8 x 8 level): wave w multiplies image w, a fragment covers two image rows (lane bit 3), HALO_W = 10, 400 halo rows,
// 50 pieces; the same column swizzle is conflict-free for it (simulated over the four 16-lane service groups of ds_read_b128 for every tap and output tile).
// SUB (round 6): nearest-x2 upsample + 3x3 conv in its SUB-PIXEL form.  Output pixel (2 y + py, 2 x + px) of the upsampled conv reads the 2 x 2 input neighbourhood
// rows {y - 1 + py, y + py} x columns {x - 1 + px, x + px} with the 3 x 3 filter's taps summed per neighbour (cs_conv_up_fold_pack): 16 multiplies per input pixel and
// channel pair instead of 36.  The tile is a plain (no UP) FAST tile over INPUT pixels whose k loop runs the 4 taps (py + a, px + b) of its phase out of the same
// 18 x 18 (FAST 2: 10 x 10) halo; p.w = [4 phases][N][4 Cin]; the phase is tm & 3; the rows scatter to the phase's output pixels (SubRows).  4 steps per chunk: the
// weight buffer of a tap is no longer a compile-time constant (4 % 3 != 0), the three address registers rotate once per chunk instead.
template <bool UP, int BN, bool TRACE = false, int FAST = 0, bool SUB = false>
__global__ __launch_bounds__(512, 2) void conv3_lw_kernel(HaloParams p) {
    static_assert(!SUB || (!UP && FAST != 0 && !TRACE), "the sub-pixel form is a FAST tile over input pixels");
    constexpr int TAPS = SUB ? 4 : 9, TL = TAPS - 1;
    unsigned long long tw_entry = 0;
    if (TRACE) tw_entry = __builtin_amdgcn_s_memrealtime();
    // (FAST with the fused nearest-x2 upsample: the 16 x 16 output patch reads an 8 x 8 input patch, HALO_W = 10, 100 halo rows, 13 pieces; output column fx
    //  under tap column dx reads halo column ((fx + dx - 1) >> 1) + 1 -- again three per-lane values -- and output row 4 w + j under tap row dy reads halo row
    //  2 w + ((j + dy - 1) >> 1) + 1: the (j, dy) part is a compile-time immediate)
    constexpr int NT = BN / 16, MT = 4;
    constexpr int B_BYTES = BN * 128, NWB = 3;
    constexpr int NBQ = BN / 8;                              // weight DMA pieces per stage (8 rows of 128 B each)
    static_assert(NBQ % 4 == 0, "every loader wave issues the same number of weight pieces");
    constexpr int WPL = NBQ / 4;                             // ... per loader wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(FAST != 2 || !UP, "the multi-image form has no fused upsample");
    const int A_BYTES = FAST ? (FAST == 2 ? 50 : UP ? 13 : 41) * 1024 : p.NQ * 1024;          // halo buffer: NQ pieces of 8 rows (<= HALO_ROWS_MAX * 128)
    char* const lA = smem;                    // [2][A_BYTES]
    char* const lB = smem + 2 * A_BYTES;      // [NWB][B_BYTES]: stage g lives in buffer g % 3

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_of(blockIdx.x, p.e.nblk, p.e.tiles_n, 1, p.e.pn, tm, tn);
    const int n_blk = tn * BN;
    int phase = 0;
    if constexpr (SUB) { phase = __builtin_amdgcn_readfirstlane(tm & 3); tm >>= 2; }                   // (the four phases of a patch are neighbours in the grid: they read the same halo)
    int b0, y0, x0;
    if (p.PP == 1) { b0 = tm << (8 - p.trw_shift); y0 = 0; x0 = 0; }
    else {
        b0 = tm / p.PP;
        const int pr = tm - b0 * p.PP, py = pr / p.PX, px = pr - py * p.PX;
        y0 = py << (p.trw_shift - p.tw_shift); x0 = px << p.tw_shift;
    }
    const int c_per = p.NC / p.splits, c_begin = blockIdx.y * c_per, c_end = c_begin + c_per;
    const int nsteps = c_per * TAPS;
    const bool btab = p.e.bias != nullptr && p.splits == 1;          // (uniform: every wave of the workgroup takes the extra barrier or none does)

    if (w >= 4) {
        // =============================== loader waves ===============================
        const int l = w - 4;
        // the tile's bias values: fetched NOW into registers of loader wave 0, handed to the compute waves through LDS behind the k loop (a bias load inside the
        // epilogue sits behind every CU's store burst; same finding as for the GEMMs, bias_tile_prologue)
        f16x8 bias_reg = {0, 0, 0, 0, 0, 0, 0, 0};
        if (btab && l == 0 && lane < BN / 8) bias_reg = *reinterpret_cast<const f16x8*>(p.e.bias + n_blk + lane * 8);
        const int pch = lane & 7, lr = lane >> 3;
        const int iy_base = (UP ? (y0 >> 1) : y0) - 1, ix_base = (UP ? (x0 >> 1) : x0) - 1;    // input pixel of halo (0, 0)
        // halo piece of slot k (0..15): q = (k & 7) * 4 + l + (k >> 3) * 32 -- slice k & 7, second round for halos of more than 32 pieces
        const char* hsrc[16];
        const float inv_img = 1.0f / (float)p.HALO_IMG, inv_w = 1.0f / (float)p.HALO_W;
        const char* zero = reinterpret_cast<const char*>(g_zero_region) + pch * 16;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int q = (k & 7) * 4 + l + (k >> 3) * 32;
            const int hr = 8 * q + lr;
            const int hi = (int)(((float)hr + 0.5f) * inv_img), rem = hr - hi * p.HALO_IMG;
            const int hy = (int)(((float)rem + 0.5f) * inv_w), hx = rem - hy * p.HALO_W;
            const int b = b0 + hi, y = iy_base + hy, x = ix_base + hx;
            const bool ok = (q < p.NQ) & (hr < p.NHALO) & (b < p.B) & (y >= 0) & (y < p.H) & (x >= 0) & (x < p.W);
            const long pix = ((long)b * p.H + y) * p.W + x;
            // source chunk of LDS slot pch: swizzled by the halo row index (hr & 7 == lr: 8 q is a multiple of 8) or, FAST, by the halo column
            hsrc[k] = ok ? reinterpret_cast<const char*>(p.x + pix * p.Cin + (pch ^ (FAST ? (hx & 7) : lr)) * 8) : zero;
        }
        const f16* wsrc[WPL];
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const int q = l + 4 * j, r = 8 * (q < NBQ ? q : 0) + lr;
            wsrc[j] = p.w + (SUB ? (size_t)phase * p.e.N * (TAPS * p.Cin) : (size_t)0) + (size_t)(n_blk + r) * (TAPS * p.Cin) + (pch ^ ((r >> 1) & 7)) * 8;
        }
        auto issue_w = [&](int c, int t, int buf) {
            const size_t koff = (size_t)t * p.Cin + c * BK;
#pragma unroll
            for (int j = 0; j < WPL; ++j) glds16(wsrc[j] + koff, lB + buf * B_BYTES + (l + 4 * j) * 1024);
        };
        // slice s (0..7) of chunk c: this wave's pieces q = 4 s + l and 4 s + l + 32; returns how many it issued
        auto issue_h = [&](int c, auto s_tag, int buf) -> int {
            constexpr int S = decltype(s_tag)::value;
            const int q0 = S * 4 + l, q1 = q0 + 32;
            int n = 0;
            if (q0 < p.NQ) { glds16(hsrc[S] + (size_t)c * 128, lA + buf * A_BYTES + q0 * 1024); ++n; }
            if (q1 < p.NQ) { glds16(hsrc[S + 8] + (size_t)c * 128, lA + buf * A_BYTES + q1 * 1024); ++n; }
            return n;
        };
        // stage 0: the whole halo of the first chunk + the first tap's weights
        issue_h(c_begin, std::integral_constant<int, 0>{}, 0); issue_h(c_begin, std::integral_constant<int, 1>{}, 0);
        issue_h(c_begin, std::integral_constant<int, 2>{}, 0); issue_h(c_begin, std::integral_constant<int, 3>{}, 0);
        issue_h(c_begin, std::integral_constant<int, 4>{}, 0); issue_h(c_begin, std::integral_constant<int, 5>{}, 0);
        issue_h(c_begin, std::integral_constant<int, 6>{}, 0); issue_h(c_begin, std::integral_constant<int, 7>{}, 0);
        issue_w(c_begin, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                           // K(-1): stage 0 has landed
        int g = 0, wb1 = 1;                                                     // wb1 = (g + 1) % 3
        // (timing experiments, debug bit 16384: cycles wave 4 spends waiting for its DMA / at the step barriers -> trace slots 9, 10)
        const bool trl = TRACE && l == 0 && lane == 0 && blockIdx.x < CS_TRACE_SLOTS && blockIdx.y == 0;
        unsigned long long tl_wait = 0, tl_bar = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const int abn = (c + 1 - c_begin) & 1;
            const bool halo_next = c + 1 < c_end;
            auto step = [&](auto t_tag) {
                constexpr int T = decltype(t_tag)::value;
                const bool last = T == TL && !halo_next;                         // the very last step: nothing to stage, but the compute waves' step is branch-free and has its barrier
                // Stage g + 1 goes into buffer (g + 1) % 3, last read as stage g - 2.  Those reads were ISSUED before K(g - 2) and every compute
                // wave has since waited for younger ones (LDS returns in order), so they are complete behind K(g - 1): the third buffer is what
                // lets the compute waves pass their barrier without an lgkmcnt(0) (an exposed LDS latency per step with one computing wave per SIMD).
                if (!last) issue_w(T == TL ? c + 1 : c, T == TL ? 0 : T + 1, wb1);
                int nh = 0;
                if constexpr (!SUB) { if constexpr (T < 8) { if (halo_next) nh = issue_h(c + 1, std::integral_constant<int, T>{}, abn); } }
                else if constexpr (T < 3) {                                      // (SUB: the eight slices ride the first three of the chunk's four steps)
                    if (halo_next) {
                        nh = issue_h(c + 1, std::integral_constant<int, 3 * T>{}, abn) + issue_h(c + 1, std::integral_constant<int, 3 * T + 1>{}, abn);
                        if constexpr (3 * T + 2 < 8) nh += issue_h(c + 1, std::integral_constant<int, 3 * T + 2>{}, abn);
                    }
                }
                // the weights are needed behind the next barrier; the halo slice just issued is not (vmcnt counts in issue order; step (c, 8) issues
                // no halo, so the whole halo of chunk c + 1 has landed when that step's barrier opens)
                unsigned long long s0 = 0, s1 = 0;
                if (TRACE) s0 = __builtin_readcyclecounter();
                if (nh == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (nh == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else if (!SUB || nh == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else if (nh == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (nh == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (nh == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                if (TRACE) s1 = __builtin_readcyclecounter();
                __builtin_amdgcn_s_barrier();                                    // K(g)
                if (TRACE) { const unsigned long long s2 = __builtin_readcyclecounter(); tl_wait += s1 - s0; tl_bar += s2 - s1; }
                ++g; wb1 = wb1 == NWB - 1 ? 0 : wb1 + 1;
            };
            step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
            if constexpr (!SUB) {
                step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
                step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
            }
        }
        if (trl) { g_trace[blockIdx.x * CS_TRACE_W + 9] = tl_wait; g_trace[blockIdx.x * CS_TRACE_W + 10] = tl_bar; }
        __builtin_amdgcn_s_barrier();                                           // E: the stage buffers become the epilogue patches
        if (btab) {                                                             // bias table behind the four compute waves' patches, then E2
            if (l == 0 && lane < BN / 8) *reinterpret_cast<f16x8*>(smem + 4 * 11264 + lane * 16) = bias_reg;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                       // E2
        }
        return;
    }

    // =============================== compute waves ===============================
    if (p.e.debug & 4096) __builtin_amdgcn_s_setprio(1);       // (timing experiment: static priority for the multiplying wave of each SIMD over its loader partner)
    // One computing wave per SIMD: nothing but this wave's own lookahead hides an LDS latency (~200 cycles with four waves reading and the DMA writing), so the
    // step is branch-free, every LDS read is an inline-asm ds_read_b128 and every wait a hand-counted lgkmcnt.  (hipcc's own waits on this loop were lgkmcnt(0) at
    // the loop head -- the back edge merges two histories -- and a 4, 3, 2, 1, 0 ladder behind the conditional reads of the last items: two exposed latencies per
    // step, 2050 cycles per step against 1280 of MFMA issue; profiles/r03_conv_lw_trace.txt.)
    //   item q = (k half ks, weight tile i), 4 MFMAs.  Reads issued in front of item q's MFMAs, in this order:
    //     weight fragment q + LA of this stage (q + LA < NQ) or q + LA - NQ of the next one  -> ring slot (q + LA) % RS, the slot item q - 1 has just used
    //     q < MT:            the k-half-1 activation fragment q of this step
    //     q = QB, QB + 1:    two activation fragments (k half 0) of the next step
    //   The barrier sits in front of item QB = NQ - LA: every weight fragment of this stage has been issued by then and the reads behind it go to the next stage.
    //   The prologue issues exactly what a step's items QB .. NQ - 1 issue, so the counts hold from the first step on; the last step's "next" reads fetch stale
    //   bytes of existing buffers and are never used.
    const int wm = w;
    int fi[MT], fy[MT], fx[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int o = wm * 64 + j * 16 + (lane & 15);
        const int img = o >> p.trw_shift, rem = o & ((1 << p.trw_shift) - 1);
        fy[j] = rem >> p.tw_shift; fx[j] = rem & ((1 << p.tw_shift) - 1); fi[j] = img * p.HALO_IMG;
        if (!UP) fi[j] += fy[j] * p.HALO_W + fx[j];
    }
    const int gq = lane >> 4;
    const int swz = (lane >> 1) & 7;
    const unsigned lA_base = lds_addr(lA), lB_base = lds_addr(lB);
    const unsigned wfrag0 = (lane & 15) * 128 + (gq ^ swz) * 16, wfrag1 = (lane & 15) * 128 + ((4 + gq) ^ swz) * 16;
    // byte offset (inside a halo buffer) of the k-half-0 fragment of output tile j under tap (dy, dx); the other k half is ^ 64
    auto a_off = [&](int j, int dy, int dx) {
        int hr;
        if (UP) hr = fi[j] + (((fy[j] + dy - 1) >> 1) + 1) * p.HALO_W + ((fx[j] + dx - 1) >> 1) + 1;
        else hr = fi[j] + dy * p.HALO_W + dx;
        return (unsigned)(hr * 128 + ((gq ^ (hr & 7)) * 16));
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NQ = 2 * NT, RS = NT / 2, LA = RS - 1, QB = NQ - LA;
    static_assert(NT % 2 == 0 && NQ % RS == 0 && LA >= 2 && MT == 4 && QB + 1 < NQ && QB > NT + MT, "ring / fragment placement");
    f16x8 fa[2][MT], fw[RS];
    unsigned aoff[MT], anext[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) aoff[j] = a_off(j, 0, 0);

    unsigned long long tc_t0 = 0, tc_r0 = 0;
    if constexpr (FAST) {
        // per-tile address registers: VA[dx][ks] = halo row of (output row 4 w, tap row 0) at column fx + dx, k half ks, in the CURRENT chunk's halo buffer
        // (moved to the other buffer once per chunk: 7 v_add per 720 MFMAs); VN = VA[0][0] in the other buffer; WB[wb][ks] = weight tile 0 of buffer wb
        constexpr int HW_F = (UP || FAST == 2) ? 10 : 18, AB_F = (FAST == 2 ? 50 : UP ? 13 : 41) * 1024, ROWB = HW_F * 128;
        const int fxl = FAST == 2 ? (lane & 7) : (lane & 15);
        // first halo row of this lane's fragments: the wave's four output rows of the 16 x 16 patch, or (FAST 2) image wm of the tile, its row 0 or 1 by lane bit 3
        const unsigned rowb = FAST == 2 ? lA_base + (unsigned)(wm * 100 + ((lane >> 3) & 1) * HW_F) * 128 : lA_base + (unsigned)((UP ? 2 : 4) * wm * HW_F) * 128;
        unsigned VA[3][2], VN, WB[NWB][2];
        if constexpr (!SUB) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int hcol = UP ? ((fxl + d - 1) >> 1) + 1 : fxl + d;                 // halo column read by this lane under tap column d
                const unsigned sw = (unsigned)((gq ^ (hcol & 7)) << 4);
                VA[d][0] = rowb + hcol * 128 + sw; VA[d][1] = rowb + hcol * 128 + (sw ^ 64u);
            }
        }
#pragma unroll
        for (int b3 = 0; b3 < NWB; ++b3) { WB[b3][0] = lB_base + b3 * B_BYTES + wfrag0; WB[b3][1] = lB_base + b3 * B_BYTES + wfrag1; }
        __builtin_amdgcn_s_barrier();                                           // K(-1): stage 0 has landed
        // the k loop for one (compile-time) first tap (DY0, DX0): (0, 0), or the phase (py, px) of the sub-pixel form
        auto kloop = [&](auto dy0_tag, auto dx0_tag) {
        constexpr int DY0 = decltype(dy0_tag)::value, DX0 = decltype(dx0_tag)::value;
        // SUB: a phase reads two halo columns per lane (DX0, DX0 + 1), its address registers are made here (all three columns of all four phases alive across the
        // branch cost the FAST 2 form a spill)
        unsigned VL[2][2];
        if constexpr (SUB) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int hcol = fxl + d + DX0;
                const unsigned sw = (unsigned)((gq ^ (hcol & 7)) << 4);
                VL[d][0] = rowb + hcol * 128 + sw; VL[d][1] = rowb + hcol * 128 + (sw ^ 64u);
            }
        }
#define CS_VA(dx, ks) (SUB ? VL[(dx) - DX0 < 0 ? 0 : (dx) - DX0 > 1 ? 1 : (dx) - DX0][ks] : VA[(dx) > 2 ? 2 : (dx)][ks])
        VN = CS_VA(DX0, 0) + AB_F;
        // prologue = the read sequence of items QB .. NQ - 1 with "next" = stage 0 (the first tap, halo buffer 0, weight buffer 0)
        lds_read<0>(fw[0], WB[0][0]); lds_read<lw_rowf(FAST, UP, 0, DY0) * ROWB>(fa[0][0], CS_VA(DX0, 0)); lds_read<lw_rowf(FAST, UP, 1, DY0) * ROWB>(fa[0][1], CS_VA(DX0, 0));
        lds_read<2048>(fw[1], WB[0][0]); lds_read<lw_rowf(FAST, UP, 2, DY0) * ROWB>(fa[0][2], CS_VA(DX0, 0)); lds_read<lw_rowf(FAST, UP, 3, DY0) * ROWB>(fa[0][3], CS_VA(DX0, 0));
        lds_read<2 * 2048>(fw[2], WB[0][0]);
        if constexpr (LA == 4) lds_read<3 * 2048>(fw[3], WB[0][0]);
        static_assert(LA == 3 || LA == 4, "prologue reads");
        if (TRACE) { tc_t0 = __builtin_readcyclecounter(); tc_r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        int delta = AB_F;
        for (int c = c_begin; c < c_end; ++c) {
            static_for<TAPS>([&](auto tc) {                                     // tap T of this chunk; the next step is tap T + 1 or the first tap of the next chunk (other halo buffer)
                constexpr int T = decltype(tc)::value;
                constexpr int TN = T == TL ? 0 : T + 1;
                constexpr int DY = SUB ? DY0 + T / 2 : T / 3, DX = SUB ? DX0 + T % 2 : T % 3, DYN = SUB ? DY0 + TN / 2 : TN / 3, DXN = SUB ? DX0 + TN % 2 : TN % 3;
                constexpr int WBC = T % 3, WBN = (T + 1) % 3;                   // (9 taps: (8 + 1) % 3 == 0, the first tap's buffer; SUB: the registers rotate per chunk)
                static_for<NQ>([&](auto qc) {
                    constexpr int q = decltype(qc)::value, ks = q / NT, i = q - ks * NT;
                    if constexpr (q == QB) __builtin_amdgcn_s_barrier();       // K(g): stage g + 1 has landed
                    {
                        constexpr int r = q + LA;
                        if constexpr (r < NT) lds_read<r * 2048>(fw[r % RS], WB[WBC][0]);
                        else if constexpr (r < NQ) lds_read<(r - NT) * 2048>(fw[r % RS], WB[WBC][1]);
                        else lds_read<(r - NQ) * 2048>(fw[r % RS], WB[WBN][0]);
                    }
                    if constexpr (q < MT) lds_read<lw_rowf(FAST, UP, q, DY) * ROWB>(fa[1][q], CS_VA(DX, 1));
                    if constexpr (q == QB || q == QB + 1) {
                        constexpr int j0 = (q - QB) * 2;
                        if constexpr (T == TL) {
                            lds_read<lw_rowf(FAST, UP, j0, DY0) * ROWB>(fa[0][j0], VN); lds_read<lw_rowf(FAST, UP, j0 + 1, DY0) * ROWB>(fa[0][j0 + 1], VN);
                        } else {
                            lds_read<lw_rowf(FAST, UP, j0, DYN) * ROWB>(fa[0][j0], CS_VA(DXN, 0)); lds_read<lw_rowf(FAST, UP, j0 + 1, DYN) * ROWB>(fa[0][j0 + 1], CS_VA(DXN, 0));
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(lw_wait_count(q, NQ, LA, MT, QB)));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < MT; ++j) mfma_inplace(acc[i][j], fw[q % RS], fa[ks][j]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            // the next chunk lives in the other halo buffer
            if constexpr (SUB) {
#pragma unroll
                for (int d = 0; d < 2; ++d) { VL[d][0] += delta; VL[d][1] += delta; }
            } else {
#pragma unroll
                for (int d = 0; d < 3; ++d) { VA[d][0] += delta; VA[d][1] += delta; }
            }
            VN -= delta; delta = -delta;
            if constexpr (SUB) {                                                // stage 4 (c + 1) + t lives in weight buffer (c + 1 + t) % 3: what was buffer [1] is the next chunk's [0]
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2) { const unsigned t0 = WB[0][k2]; WB[0][k2] = WB[1][k2]; WB[1][k2] = WB[2][k2]; WB[2][k2] = t0; }
            }
        }
        };
#undef CS_VA
        if constexpr (!SUB) kloop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        else {
            // uniform branch on the tile's phase: four copies of the loop, one executed
            if (phase == 0) kloop(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            else if (phase == 1) kloop(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            else if (phase == 2) kloop(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            else kloop(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        }
    } else {
    __builtin_amdgcn_s_barrier();                                               // K(-1): stage 0 has landed
    // prologue = the read sequence of items QB .. NQ - 1 with "next" = stage 0
    {
        const unsigned w0 = lB_base + wfrag0, ha0 = lA_base;
        lds_read<0>(fw[0], w0); lds_read<0>(fa[0][0], ha0 + aoff[0]); lds_read<0>(fa[0][1], ha0 + aoff[1]);
        lds_read<2048>(fw[1], w0); lds_read<0>(fa[0][2], ha0 + aoff[2]); lds_read<0>(fa[0][3], ha0 + aoff[3]);
#pragma unroll
        for (int r = 2; r < LA; ++r) {
            if (r == 2) lds_read<2 * 2048>(fw[2 % RS], w0);
            if (r == 3) lds_read<3 * 2048>(fw[3 % RS], w0);
        }
        static_assert(LA <= 4, "prologue reads");
    }

    if (TRACE) { tc_t0 = __builtin_readcyclecounter(); tc_r0 = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    int ab = 0, t = 0, dy = 0, dx = 0, wb = 0;                                   // wb = g % 3
    for (int g = 0; g < nsteps; ++g) {
        // next step's tap / halo buffer
        int tn = t + 1, dxn = dx + 1, dyn = dy, abn = ab;
        if (dxn == 3) { dxn = 0; ++dyn; }
        if (tn == 9) { tn = 0; dxn = 0; dyn = 0; abn ^= 1; }
        const int wbn = wb == NWB - 1 ? 0 : wb + 1;
        const unsigned ha1 = lA_base + ab * A_BYTES, han = lA_base + abn * A_BYTES;
        const unsigned wc0 = lB_base + wb * B_BYTES + wfrag0, wc1 = lB_base + wb * B_BYTES + wfrag1, wn0 = lB_base + wbn * B_BYTES + wfrag0;
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value, ks = q / NT, i = q - ks * NT;
            if constexpr (q == QB) __builtin_amdgcn_s_barrier();               // K(g): stage g + 1 has landed (no lgkmcnt(0): the loader's comment on the third buffer)
            // ---- reads -----------------------------------------------------------------------------------------------
            {
                constexpr int r = q + LA;
                if constexpr (r < NT) lds_read<r * 2048>(fw[r % RS], wc0);
                else if constexpr (r < NQ) lds_read<(r - NT) * 2048>(fw[r % RS], wc1);
                else lds_read<(r - NQ) * 2048>(fw[r % RS], wn0);
            }
            if constexpr (q < MT) lds_read<0>(fa[1][q], ha1 + (aoff[q] ^ 64u));
            if constexpr (q == QB || q == QB + 1) {
                lds_read<0>(fa[0][(q - QB) * 2], han + anext[(q - QB) * 2]);
                lds_read<0>(fa[0][(q - QB) * 2 + 1], han + anext[(q - QB) * 2 + 1]);
            }
            // ---- wait for what this item multiplies --------------------------------------------------------------------
            lds_wait<lw_wait_count(q, NQ, LA, MT, QB)>(fw[q % RS]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MT; ++j)
                mfma_inplace(acc[i][j], fw[q % RS], fa[ks][j]);
            if constexpr (q < MT) anext[q] = a_off(q, dyn, dxn);                 // (address arithmetic of the next step, one tile per item)
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int j = 0; j < MT; ++j) aoff[j] = anext[j];
        t = tn; dx = dxn; dy = dyn; ab = abn; wb = wbn;
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");          // (the last step's look-ahead reads; the last MFMAs' results: hipcc pads nothing behind an asm MFMA)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));            // (the epilogue's reads of the accumulators stay behind the padding: hipcc may move consumers of an asm result up between volatile statements)
    if (TRACE && w == 0 && lane == 0 && blockIdx.x < CS_TRACE_SLOTS && blockIdx.y == 0) {
        g_trace[blockIdx.x * CS_TRACE_W + 6] = __builtin_readcyclecounter() - tc_t0;
        g_trace[blockIdx.x * CS_TRACE_W + 8] = __builtin_amdgcn_s_memrealtime() - tc_r0;
        g_trace[blockIdx.x * CS_TRACE_W + 5] = (unsigned long long)nsteps;
        g_trace[blockIdx.x * CS_TRACE_W + 0] = tw_entry;                                        // 100 MHz stamps: entry | k loop start | k loop end (exit: slot 3, below)
        g_trace[blockIdx.x * CS_TRACE_W + 1] = tc_r0;
        g_trace[blockIdx.x * CS_TRACE_W + 2] = __builtin_amdgcn_s_memrealtime();
        g_trace[blockIdx.x * CS_TRACE_W + 4] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32) | __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    }
    __builtin_amdgcn_s_barrier();                                               // E
    if (btab) __builtin_amdgcn_s_barrier();                                     // E2: loader wave 0 has put the tile's bias values into LDS
    if constexpr (SUB) {
        // (launch_igemm_impl: never split-K) the tile's rows are INPUT pixels; phase (py, px) writes output pixels (2 y + py, 2 x + px)
        const SubRows rows{wm * 64, b0, y0, x0, p.B, p.Ho, p.Wo, p.tw_shift, p.trw_shift, phase >> 1, phase & 1, phase * ((p.Ho * p.Wo) >> 8)};
        igemm_epilogue<false, NT, MT, NT, SubRows, 2, 0, true>(p.e, acc, rows, n_blk, lane, smem + w * 11264, btab ? smem + 4 * 11264 : nullptr);
        return;
    }
    const PatchRows rows{wm * 64, b0, y0, x0, p.B, p.Ho, p.Wo, p.tw_shift, p.trw_shift};
    if (p.splits == 1) {
        igemm_epilogue<false, NT, MT, NT, PatchRows, 2, 0, true>(p.e, acc, rows, n_blk, lane, smem + w * 11264, btab ? smem + 4 * 11264 : nullptr);
    } else {
        float* dst = p.partial + (size_t)blockIdx.y * p.e.M * p.e.N;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = rows(j * 16 + (lane & 15));
            if (m < 0) continue;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                *reinterpret_cast<f32x4*>(dst + (size_t)m * p.e.N + n_blk + i * 16 + gq * 4) = acc[i][j];
        }
    }
    if (TRACE && w == 0 && lane == 0 && blockIdx.x < CS_TRACE_SLOTS && blockIdx.y == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                        // (the stores have left the CU)
        g_trace[blockIdx.x * CS_TRACE_W + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------
// Linear / 1x1 layers whose 256 x 320 tiling cannot fill the chip (the 1280-wide layers at 16 x 16: M = 8192 -> 128 tiles), in the loader-wave form of
// conv3_lw_kernel (round 3): tile 256 rows x 160 columns, waves 0-3 multiply (wave tile 64 x 160: 80 MFMAs, 28 hand-counted LDS reads, no vector ALU per
// step), waves 4-7 stage [256 + 160 rows][64 k] = 52 KB per step into one of THREE stage buffers (156 of the CU's 160 KB), two steps ahead.
// With ONE tile per CU and the operands in L2 / Infinity Cache this form is latency-free; as a general GEMM it loses to the 256 x 320 tile (twice the
// L2 -> LDS bytes per FLOP: profiles/r03_ab_gemm_lw.txt), so launch_igemm_impl uses it only where gemm_big_kernel<false, 160> used to run.
// ------------------------------------------------------------------------------------------------
template <int LNM = 0, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_lw_kernel(IgemmParams p) {
    constexpr int BMX = 256, BN = 160, NT = BN / 16, MT = 4, NWB = 3;
    constexpr int A_BYTES = BMX * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int APL = BMX / 8 / 4, WPL = BN / 8 / 4;        // DMA pieces (8 rows x 128 B) per loader wave and stage: 8 + 5
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    tile_of(blockIdx.x, p.nblk, p.tiles_n, p.gm, p.pn, tm, tn);
    const int m_blk = tm * BMX, n_blk = tn * BN;
    const int KT = p.KT;
    char* const lnx = smem + NWB * STAGE;                       // LNM == 1: 256 x (rstd, mean rstd) + one (b' | s) table of the tile's 160 weight rows, behind the stages

    if (w >= 4) {
        // =============================== loader waves ===============================
        const int l = w - 4;

        const int pch = lane & 7, lr = lane >> 3;
        // rows past M are clamped to the last row: their products are computed and never stored
        const f16* asrc0[APL]; const f16* asrc1[APL]; const f16* wsrc[WPL];
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const int r = 8 * (l + 4 * j) + lr;
            const long m = min(m_blk + r, p.M - 1);
            const int ch = (pch ^ ((r >> 1) & 7)) * 8;
            asrc0[j] = p.a0 + m * p.c0 + ch;
            asrc1[j] = p.c1 ? p.a1 + m * p.c1 + ch : asrc0[j];
        }
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const int r = 8 * (l + 4 * j) + lr;
            wsrc[j] = p.w + (size_t)(n_blk + r) * p.Ktot + (pch ^ ((r >> 1) & 7)) * 8;
        }
        // lo planes of a split-fp16 A operand: same layout as the hi planes, so a k step >= KTh adds the (uniform) distance between the planes
        const long dlo0 = p.a0_lo ? p.a0_lo - p.a0 : 0, dlo1 = (p.a1_lo && p.c1) ? p.a1_lo - p.a1 : dlo0;
        const int KTh = p.KTh;
        auto issue = [&](int kt, int buf) {
            char* st = smem + buf * STAGE;
            const bool lo = kt >= KTh;
            const int cc = (lo ? kt - KTh : kt) * BK;
            if (cc < p.c0) {
                const long o = cc + (lo ? dlo0 : 0);
#pragma unroll
                for (int j = 0; j < APL; ++j) glds16(asrc0[j] + o, st + (l + 4 * j) * 1024);
            } else {
                const long o = (cc - p.c0) + (lo ? dlo1 : 0);
#pragma unroll
                for (int j = 0; j < APL; ++j) glds16(asrc1[j] + o, st + (l + 4 * j) * 1024);
            }
#pragma unroll
            for (int j = 0; j < WPL; ++j) glds16(wsrc[j] + cc, st + A_BYTES + (l + 4 * j) * 1024);
        };
        // two stages ahead (an activation tile may come from the Infinity Cache: ~1 us under load, more than one step).  Stage g + 2 goes out behind
        // K(g - 1) into buffer (g + 2) % 3 = (g - 1) % 3 -- the compute waves wait lgkmcnt(0) in front of their barrier here, so those reads are complete --
        // and K(g) needs stage g + 1: everything but the 13 pieces just issued.
        issue(0, 0);
        if (KT > 1) issue(1, 1);
        // folded LayerNorm: the tile's row statistics and b' / s table, fetched behind the first two stages (one memory round trip for all of it)
        if constexpr (LNM == 1) ln_tile_prologue(p, lnx, tid - 256, 256, m_blk, BMX, n_blk, BN, 1);
        else bias_tile_prologue(p, lnx, tid - 256, n_blk, BN);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                           // K(-1): stages 0 and 1 have landed
        int wb2 = 2;
        static_assert(APL + WPL == 13, "counted wait below");
        for (int g = 0; g < KT; ++g) {
            if (g + 2 < KT) { issue(g + 2, wb2); asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                        // K(g)
            wb2 = wb2 == NWB - 1 ? 0 : wb2 + 1;
        }
        __builtin_amdgcn_s_barrier();                                           // E: the stage buffers become the epilogue patches
        return;
    }

    // =============================== compute waves ===============================
    const int wm = w;
    const int gq = lane >> 4, swz = (lane >> 1) & 7;
    const unsigned sbase = lds_addr(smem);
    const unsigned wfrag0 = (lane & 15) * 128 + (gq ^ swz) * 16, wfrag1 = (lane & 15) * 128 + ((4 + gq) ^ swz) * 16;
    unsigned SA[NWB][2], SW[NWB][2];                            // fragment bases: rows 64 wm .. of stage buffer b / weight tile 0 of stage buffer b; k half 0 / 1
#pragma unroll
    for (int b3 = 0; b3 < NWB; ++b3) {
        SA[b3][0] = sbase + b3 * STAGE + wm * 8192 + wfrag0; SA[b3][1] = sbase + b3 * STAGE + wm * 8192 + wfrag1;
        SW[b3][0] = sbase + b3 * STAGE + A_BYTES + wfrag0; SW[b3][1] = sbase + b3 * STAGE + A_BYTES + wfrag1;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NQ = 2 * NT, RS = NT / 2, LA = RS - 1, QB = NQ - LA;
    static_assert(LA == 4 && MT == 4, "read schedule of conv3_lw_kernel (lw_wait_count)");
    f16x8 fa[2][MT], fw[RS];

    __builtin_amdgcn_s_barrier();                                               // K(-1)
    // prologue = the read sequence of items QB .. NQ - 1 with "next" = stage 0
    lds_read<0>(fw[0], SW[0][0]); lds_read<0>(fa[0][0], SA[0][0]); lds_read<2048>(fa[0][1], SA[0][0]);
    lds_read<2048>(fw[1], SW[0][0]); lds_read<2 * 2048>(fa[0][2], SA[0][0]); lds_read<3 * 2048>(fa[0][3], SA[0][0]);
    lds_read<2 * 2048>(fw[2], SW[0][0]); lds_read<3 * 2048>(fw[3], SW[0][0]);
    auto step = [&](auto b_tag) {                                               // stage in buffer B, the next one in (B + 1) % 3
        constexpr int B = decltype(b_tag)::value, BNX = (B + 1) % NWB;
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value, ks = q / NT, i = q - ks * NT;
            if constexpr (q == QB) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's reads of stage g are complete: its buffer is refilled right behind the barrier
                __builtin_amdgcn_s_barrier();                                    // K(g): stage g + 1 has landed
            }
            {
                constexpr int r = q + LA;
                if constexpr (r < NT) lds_read<r * 2048>(fw[r % RS], SW[B][0]);
                else if constexpr (r < NQ) lds_read<(r - NT) * 2048>(fw[r % RS], SW[B][1]);
                else lds_read<(r - NQ) * 2048>(fw[r % RS], SW[BNX][0]);
            }
            if constexpr (q < MT) lds_read<q * 2048>(fa[1][q], SA[B][1]);
            if constexpr (q == QB || q == QB + 1) {
                constexpr int j0 = (q - QB) * 2;
                lds_read<j0 * 2048>(fa[0][j0], SA[BNX][0]); lds_read<(j0 + 1) * 2048>(fa[0][j0 + 1], SA[BNX][0]);
            }
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(lw_wait_count(q, NQ, LA, MT, QB)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MT; ++j) mfma_inplace(acc[i][j], fw[q % RS], fa[ks][j]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int kt = 0;;) {
        step(std::integral_constant<int, 0>{}); if (++kt == KT) break;
        step(std::integral_constant<int, 1>{}); if (++kt == KT) break;
        step(std::integral_constant<int, 2>{}); if (++kt == KT) break;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");          // (the last step's look-ahead reads; the last MFMAs' results)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));            // (the epilogue's reads of the accumulators stay behind the padding: hipcc may move consumers of an asm result up between volatile statements)
    __builtin_amdgcn_s_barrier();                                               // E
    igemm_epilogue<false, NT, MT, NT, LinearRows, 2, LNM, true, EPI>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk, lane, smem + w * 11264, LNM == 1 ? lnx + BMX * 8 : lnx, LNM == 1 ? lnx + wm * 64 * 8 : nullptr);
}

// ------------------------------------------------------------------------------------------------
// The 256 x 320 x 64 GEMM tile (8 waves, all multiplying, wave tile 64 x 160, two 72 KB stages) with the k loop written like conv3_lw_kernel's
// (round 3).  gemm_big_kernel's loop carried ~180 vector-ALU instructions per wave and step -- the zero-page select and the 64-bit address of its
// nine LDS-DMA pieces -- plus hipcc's own waits (lgkmcnt(0) at the loop head).  Here:
//   * LDS-DMA as buffer_load ... lds: per-piece 32-bit row offsets computed once per tile, the k offset in an SGPR (soffset): no vector ALU;
//     rows past M are clamped to the last row (their products are never stored) instead of being redirected to a zero page;
//   * every fragment read is an inline-asm ds_read_b128 at (per-tile register + immediate), every wait a hand-counted lgkmcnt, MFMAs in place;
//   * piece schedule: behind barrier K(g) (item 16 of step g) buffer g & 1 is free (every wave has waited lgkmcnt(0) in front of the barrier):
//     pieces 0..3 of stage g + 2 go out with items 16..19, pieces 4..8 with items 0..4 of step g + 1; vmcnt(0) in front of K(g + 1).
// Activations / weights must be addressable with 32-bit byte offsets (M * K * 2 < 4 GiB), else the caller falls back to gemm_big_kernel.
// Measured against it (round 3, same-box A/Bs under profiles/): the loader-wave form on a 256 x 160 tile as a GENERAL GEMM (r03_ab_gemm_lw.txt: bit-identical,
// +20..40 % time on the multi-round K <= 2560 shapes -- twice the L2 -> LDS bytes per FLOP and that path is the bound; kept for the one-round shapes only) and the 256 x 320 tile on four 512-register waves
// with the accumulators in AGPRs (r03_ab_gemm_w4.txt: bit-identical, +10..30 % -- half the waves for a store-bound epilogue, no k-loop gain).
// ------------------------------------------------------------------------------------------------
// (Round 4 tried this kernel as a PERSISTENT one -- one workgroup per CU walking the tiles, stage 0 of the next tile issued in front of the current tile's epilogue:
// bit-identical, 0 .. -4 % per launch, UNet forward unchanged (profiles/r04_ab_gemm_persist.txt), removed.  Two things ate the prologue it was meant to hide: hipcc
// puts s_waitcnt vmcnt(0) in front of the epilogue's first LDS access while an LDS-DMA it knows of is in flight (it cannot tell the patch from the stage), and the wait
// for the prefetched stage at the top of the next tile also waits for the epilogue's stores -- vmcnt counts loads, stores and LDS-DMA together, in issue order.)
template <bool GEGLU, int LNM = 0, int EPI = 0>
__global__ __launch_bounds__(512, 2) void gemm_w8_kernel(IgemmParams p) {
    constexpr int BMX = 256, BNX = 320, NT = BNX / 32, MT = 4;
    constexpr int A_BYTES = BMX * 128, B_BYTES = BNX * 128, STAGE = A_BYTES + B_BYTES;      // 32 KB + 40 KB
    constexpr int NPC = 9;                                       // DMA pieces per wave and stage: 4 of the activations + 5 of the weights
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    int tm, tn;
    tile_of(blockIdx.x, p.nblk, p.tiles_n, p.gm, p.pn, tm, tn);
    const int m_blk = tm * BMX, n_blk = tn * BNX;
    const int KT = p.KT;
    char* const lnx = smem + 2 * STAGE;                         // LNM == 1: 256 x (rstd, mean rstd) + two (b' | s) tables (column halves wn = 0 / 1), behind the stages

    // ---- staging: wave w owns activation pieces 4 w .. 4 w + 3 (rows 32 w ..) and weight pieces w, w + 8, .. w + 32 ----------------------------
    const int pch = lane & 7, lr = lane >> 3;
    unsigned aoff0[4], aoff1[4], woff[5];                        // byte offsets from a0 / a1 / w (row * row length + swizzled chunk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (w * 4 + j) + lr;
        const unsigned m = (unsigned)min(m_blk + r, p.M - 1);
        const unsigned ch = (unsigned)((pch ^ ((r >> 1) & 7)) * 16);
        aoff0[j] = m * (unsigned)(p.c0 * 2) + ch; aoff1[j] = m * (unsigned)(p.c1 * 2) + ch;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int r = 8 * (w + 8 * j) + lr;
        woff[j] = (unsigned)(n_blk + r) * (unsigned)(p.Ktot * 2) + (unsigned)((pch ^ ((r >> 1) & 7)) * 16);
    }
    const __amdgpu_buffer_rsrc_t ra0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a0, 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.c1 ? p.a1 : p.a0), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0xffffffff, 0x00020000);
    // lo planes of a split-fp16 A operand (same row offsets as the hi planes): k steps >= KTh
    const __amdgpu_buffer_rsrc_t ra0l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a0_lo ? p.a0_lo : p.a0), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra1l = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a1_lo ? p.a1_lo : (p.c1 ? p.a1 : p.a0)), 0, 0xffffffff, 0x00020000);
    const int KTh = p.KTh;
    // piece n (0..8) of stage kt into stage buffer `buf`
    auto piece = [&](auto n_tag, int kt, int buf) {
        constexpr int n = decltype(n_tag)::value;
        const bool lo = kt >= KTh;                              // (scalar: kt and KTh are wave-uniform)
        const int cc = (lo ? kt - KTh : kt) * BK;
        if constexpr (n < 4) {
            lptr_t dst = (lptr_t)(smem + buf * STAGE + (w * 4 + n) * 1024);
            if (!lo) {
                if (cc < p.c0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0, dst, 16, aoff0[n], cc * 2, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1, dst, 16, aoff1[n], (cc - p.c0) * 2, 0, 0);
            } else {
                if (cc < p.c0) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra0l, dst, 16, aoff0[n], cc * 2, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(ra1l, dst, 16, aoff1[n], (cc - p.c0) * 2, 0, 0);
            }
        } else {
            lptr_t dst = (lptr_t)(smem + buf * STAGE + A_BYTES + (w + 8 * (n - 4)) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, woff[n - 4], cc * 2, 0, 0);
        }
    };

    // ---- fragment addressing: per-tile registers + immediates ----------------------------------------------------------------------------------
    const int gq = lane >> 4, swz = (lane >> 1) & 7;
    const unsigned sbase = lds_addr(smem);
    const unsigned wfrag0 = (lane & 15) * 128 + (gq ^ swz) * 16, wfrag1 = (lane & 15) * 128 + ((4 + gq) ^ swz) * 16;
    unsigned SA[2][2], SW[2][2];
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2) {
        SA[b2][0] = sbase + b2 * STAGE + wm * 8192 + wfrag0; SA[b2][1] = sbase + b2 * STAGE + wm * 8192 + wfrag1;
        SW[b2][0] = sbase + b2 * STAGE + A_BYTES + wn * (BNX / 2) * 128 + wfrag0; SW[b2][1] = sbase + b2 * STAGE + A_BYTES + wn * (BNX / 2) * 128 + wfrag1;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int NQ = 2 * NT, RS = NT / 2, LA = RS - 1, QB = NQ - LA;
    static_assert(LA == 4 && MT == 4 && NPC == 9, "read / piece schedule");
    f16x8 fa[2][MT], fw[RS];

    // prologue: stage 0, then what items 16..19 of a step issue (pieces 0..3 of the next stage, the next step's first fragments)
    static_for<NPC>([&](auto nc) { piece(nc, 0, 0); });
    // folded LayerNorm: the tile's row statistics and b' / s tables, fetched behind stage 0 (one memory round trip for all of it; published by the barrier below)
    if constexpr (LNM == 1) ln_tile_prologue(p, lnx, tid, 512, m_blk, BMX, n_blk, BNX / 2, 2);
    else bias_tile_prologue(p, lnx, tid, n_blk, BNX);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (KT > 1) static_for<4>([&](auto nc) { piece(nc, 1, 1); });
    lds_read<0>(fw[0], SW[0][0]); lds_read<0>(fa[0][0], SA[0][0]); lds_read<2048>(fa[0][1], SA[0][0]);
    lds_read<2048>(fw[1], SW[0][0]); lds_read<2 * 2048>(fa[0][2], SA[0][0]); lds_read<3 * 2048>(fa[0][3], SA[0][0]);
    lds_read<2 * 2048>(fw[2], SW[0][0]); lds_read<3 * 2048>(fw[3], SW[0][0]);

    int kt = 0;
    auto step = [&](auto b_tag) {                                               // stage kt in buffer B
        constexpr int B = decltype(b_tag)::value, BO = B ^ 1;
        static_for<NQ>([&](auto qc) {
            constexpr int q = decltype(qc)::value, ks = q / NT, i = q - ks * NT;
            if constexpr (q == QB) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // stage kt + 1 (this wave's pieces) has landed; this wave's reads of stage kt are done
                __builtin_amdgcn_s_barrier();                                    // K(kt)
            }
            {
                constexpr int r = q + LA;
                if constexpr (r < NT) lds_read<r * 2048>(fw[r % RS], SW[B][0]);
                else if constexpr (r < NQ) lds_read<(r - NT) * 2048>(fw[r % RS], SW[B][1]);
                else lds_read<(r - NQ) * 2048>(fw[r % RS], SW[BO][0]);
            }
            if constexpr (q < MT) lds_read<q * 2048>(fa[1][q], SA[B][1]);
            if constexpr (q == QB || q == QB + 1) {
                constexpr int j0 = (q - QB) * 2;
                lds_read<j0 * 2048>(fa[0][j0], SA[BO][0]); lds_read<(j0 + 1) * 2048>(fa[0][j0 + 1], SA[BO][0]);
            }
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(lw_wait_count(q, NQ, LA, MT, QB)));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < MT; ++j) mfma_inplace(acc[i][j], fw[q % RS], fa[ks][j]);
            __builtin_amdgcn_sched_barrier(0);
            // staging: pieces 4..8 of stage kt + 1 (buffer BO) with items 0..4, pieces 0..3 of stage kt + 2 (buffer B, free behind K(kt)) with items 16..19
            if constexpr (q < 5) { if (kt + 1 < KT) piece(std::integral_constant<int, 4 + q>{}, kt + 1, BO); }
            if constexpr (q >= QB) { if (kt + 2 < KT) piece(std::integral_constant<int, q - QB>{}, kt + 2, B); }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (;;) {
        step(std::integral_constant<int, 0>{}); if (++kt == KT) break;
        step(std::integral_constant<int, 1>{}); if (++kt == KT) break;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));            // (the epilogue's reads of the accumulators stay behind the padding: hipcc may move consumers of an asm result up between volatile statements)
    __builtin_amdgcn_s_barrier();                                               // the stage buffers become the epilogue patches
    char* const ln_tab = LNM == 1 ? lnx + BMX * 8 + wn * (2 * (BNX / 2) * 4) : lnx + wn * (BNX / 2) * 2;      // (LNM != 1: the wave's half of the bias table)
    const char* const ln_rows = LNM == 1 ? lnx + wm * 64 * 8 : nullptr;
    if constexpr (GEGLU) igemm_epilogue<GEGLU, NT, MT, NT, LinearRows, 1, LNM, true>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk + wn * (BNX / 2), lane, smem + w * 11264, ln_tab, ln_rows);
    else igemm_epilogue<GEGLU, NT, MT, NT, LinearRows, 2, LNM, true, EPI>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk + wn * (BNX / 2), lane, smem + w * 11264, ln_tab, ln_rows);
}

// row statistics of a [M][C] tensor (value = x + x_lo when x_lo != null): stats[M][1][2] = (sum, sum of squares) per row.  The fallback of IgemmArgs::row_stats
// for the kernels whose epilogue cannot leave them (split-K forms), and cs_op_row_stats.  One wave per row.
__global__ __launch_bounds__(256) void row_stats_kernel(const f16* __restrict__ x, const f16* __restrict__ x_lo, int M, int C, float* __restrict__ stats, int lo8) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(x + (size_t)row * C + c);
        float fv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) fv[k] = (float)t[k];
        if (x_lo && lo8) lo8_decode_add(fv, *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(x_lo) + (size_t)row * C + c));
        else if (x_lo) {
            const f16x8 t2 = *reinterpret_cast<const f16x8*>(x_lo + (size_t)row * C + c);
#pragma unroll
            for (int k = 0; k < 8; ++k) fv[k] += (float)t2[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float f = fv[k]; s1 += f; s2 = __builtin_fmaf(f, f, s2); }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { stats[2 * (size_t)row] = s1; stats[2 * (size_t)row + 1] = s2; }
}

// out = sum_s partial[s] + bias + temb + res  (8 channels per thread): one (row m, 8-channel chunk n) item; returns the fp16 values stored to the hi plane
__device__ __forceinline__ f16x8 splitk_reduce_item(const IgemmParams& p, const float* __restrict__ partial, int splits, int m, int n) {
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int s = 0; s < splits; ++s) {
        const float* src = partial + ((size_t)s * p.M + m) * p.N + n;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += a[k]; v[4 + k] += b[k]; }
    }
    if (p.ln_stats) {                                      // LayerNorm folded into this GEMM (see igemm_epilogue_impl)
        float s1 = 0.f, s2 = 0.f;
        const float* st = p.ln_stats + (size_t)m * p.ln_groups * 2;
        ln_row_moments(st, p.ln_groups, s1, s2);
        const float mean = s1 * p.ln_inv_c, rstd = __builtin_amdgcn_rsqf(fmaxf(__builtin_fmaf(-mean, mean, s2 * p.ln_inv_c), 0.f) + p.ln_eps), mr = mean * rstd;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], rstd, __builtin_fmaf(-mr, p.ln_s[n + k], p.ln_b[n + k]));
    } else if (p.bias) { const f16x8 t = *reinterpret_cast<const f16x8*>(p.bias + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)t[k]; }
    if (p.temb) { const f16x8 t = *reinterpret_cast<const f16x8*>(p.temb + (size_t)(m / p.HoWo) * p.temb_stride + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)t[k]; }
    if (p.res) { const f16x8 t = *reinterpret_cast<const f16x8*>(p.res + (size_t)m * p.N + n);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += (float)t[k]; }
    if (p.res && p.res_lo) {
        if (p.lo8) lo8_decode_add(v, *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const unsigned char*>(p.res_lo) + (size_t)m * p.N + n));
        else { const f16x8 t = *reinterpret_cast<const f16x8*>(p.res_lo + (size_t)m * p.N + n);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += (float)t[k]; }
    }
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)v[k];
    *reinterpret_cast<f16x8*>(p.out + (size_t)m * p.N + n) = o;
    if (p.out_lo) {
        float d8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d8[k] = v[k] - (float)o[k];
        if (p.lo8) *reinterpret_cast<u32x2_t*>(reinterpret_cast<unsigned char*>(p.out_lo) + (size_t)m * p.N + n) = lo8_encode(d8);
        else {
            f16x8 l;
#pragma unroll
            for (int k = 0; k < 8; ++k) l[k] = (f16)d8[k];
            *reinterpret_cast<f16x8*>(p.out_lo + (size_t)m * p.N + n) = l;
        }
    }
    return o;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(IgemmParams p, const float* __restrict__ partial, int splits) {
    const int NV = p.N >> 3;
    const long total = (long)p.M * NV;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i / NV), n = (int)(i - (long)m * NV) * 8;
        splitk_reduce_item(p, partial, splits, m, n);
    }
}

// Round 5, the 8 x 8 level (64 pixels per sample = ONE statistics slot): the reduce also leaves the GroupNorm statistics of its output, in the layout the conv
// epilogues write (gn_stats[B][1][N / 2][2] = (sum, sum of squares) of channel pairs over the sample's 64 rows, from the fp16 values of the hi plane).  The split-K
// layers had a gn_stats_kernel launch behind every reduce (16 per UNet forward, 10 us each).  A workgroup owns 64 rows (one sample) x 32 channels: thread = (row, chunk of
// 8 channels); rows are reduced by wave shuffles (a wave holds 16 rows x 4 chunks) and across the four waves through LDS.  Deterministic (fixed order).
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(IgemmParams p, const float* __restrict__ partial, int splits, float* __restrict__ gn_stats) {
    __shared__ float red[4][4][16];                       // [wave][chunk][8 sums | 8 sums of squares]
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row = t >> 2, chunk = t & 3;
    const int b = blockIdx.y, n = blockIdx.x * 32 + chunk * 8, m = b * 64 + row;
    const f16x8 o = splitk_reduce_item(p, partial, splits, m, n);
    float sm[8], sq[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float f = (float)o[k]; sm[k] = f; sq[k] = f * f; }
#pragma unroll
    for (int off = 4; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm[k] += __shfl_xor(sm[k], off, 64); sq[k] += __shfl_xor(sq[k], off, 64); }
    if (lane < 4) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[w][lane][k] = sm[k]; red[w][lane][8 + k] = sq[k]; }
    }
    __syncthreads();
    if (t < 16) {                                         // 16 channel pairs of the workgroup's 32 channels: pair t = chunk t >> 2, channels 2 (t & 3), 2 (t & 3) + 1
        const int c = t >> 2, k = 2 * (t & 3);
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) { a += red[ww][c][k] + red[ww][c][k + 1]; q += red[ww][c][8 + k] + red[ww][c][8 + k + 1]; }
        *reinterpret_cast<f32x2*>(gn_stats + ((size_t)b * (p.N >> 1) + (blockIdx.x * 16 + t)) * 2) = f32x2{a, q};
    }
}

// the reduce behind a split-K launch.  With p.gn_stats set (statistics wanted and allowed: launch_igemm_impl) and one 64-row statistics slot per sample it also
// leaves the GroupNorm statistics: reduce_leaves_stats(p) tells the caller that no statistics pass is needed.
static bool reduce_leaves_stats(const IgemmParams& p) { return p.gn_stats && p.HoWo == 64 && p.M % 64 == 0 && p.N % 32 == 0 && !(p.debug & 64); }      // (debug bit 64: A/B against the statistics pass)
static int launch_splitk_reduce(const IgemmParams& p, const float* partial, int splits, hipStream_t s) {
    if (reduce_leaves_stats(p)) {
        hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3(p.N / 32, p.M / 64), dim3(256), 0, s, p, partial, splits, p.gn_stats);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    const long total = (long)p.M * (p.N / 8);
    int grid2 = (int)((total + 255) / 256); if (grid2 > 2048) grid2 = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid2), dim3(256), 0, s, p, partial, splits);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// ------------------------------------------------------------------------------------------------
// Large-tile GEMM for the linear / 1x1 layers: 256 x 320 x 64 tile, 8 wave64 (4 x 2), wave tile
// 64 x 160.  Every SD1.5 layer width (320 ... 10240) is a multiple of 320, and the tile stages
// 72 KB per 10.5 MFLOP k-step = 142 FLOP per LDS-DMA byte, 2x the 128 x {128,160} tile: these
// layers were bound by L2->LDS traffic (the activation panel was re-staged N/160 times), not MFMA.
// Two LDS stages (144 KB), one workgroup per CU, 9 DMA issues per wave per 80 MFMAs.
// ------------------------------------------------------------------------------------------------
template <bool GEGLU, int BNX>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(IgemmParams p) {
    // BNX = 320: wave tile 64 x 160 (every SD1.5 width >= 320 with enough rows); BNX = 160: wave tile 64 x 80 for the 1280-wide layers
    // at 16 x 16 (M = 8192: 32 x 8 = 256 tiles = one per CU, where 256 x 320 tiles would leave half the chip idle)
    constexpr int BMX = 256, NT = BNX / 32, NH = NT / (NT % 2 == 0 ? 2 : 1), MT = 4, NBP = BNX / 8;
    constexpr int A_BYTES = BMX * BK * 2, B_BYTES = BNX * BK * 2, STAGE = A_BYTES + B_BYTES;
    constexpr int BPW = (NBP + 7) / 8;                       // weight-tile DMA pieces per wave (40 -> 5; 20 -> 3 for waves 0-3, 2 for 4-7)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    int tm, tn;
    tile_of(blockIdx.x, p.nblk, p.tiles_n, p.gm, p.pn, tm, tn);
    const int m_blk = tm * BMX, n_blk = tn * BNX;

    const int pch = lane & 7;
    int a_row[4], a_chunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (w * 4 + j) + (lane >> 3);
        const int m = m_blk + r;
        a_chunk[j] = (pch ^ ((r >> 1) & 7)) * 8;
        a_row[j] = (m < p.M && !(p.debug & 65536)) ? m : -1;                                           // (timing experiment: every activation piece from the zero page)
    }
    const f16* b_src[BPW];
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
        const int q = w + 8 * j;                             // piece index; pieces >= NBP are never issued
        const int r = 8 * (q < NBP ? q : 0) + (lane >> 3);
        b_src[j] = p.w + (size_t)(n_blk + r) * p.Ktot + (pch ^ ((r >> 1) & 7)) * 8;
    }
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + pch * 16;

    auto stage = [&](int kt, int buf) {
        const int cc = kt * BK;
        const f16* src; int cs, coff;
        if (cc < p.c0) { src = p.a0; cs = p.c0; coff = cc; } else { src = p.a1; cs = p.c1; coff = cc - p.c0; }
        char* la = smem + buf * STAGE + (w * 4) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = a_row[j] >= 0;
            const uintptr_t real = (uintptr_t)(src + ((long)a_row[j] * cs + coff + a_chunk[j]));
            const uintptr_t msk = (uintptr_t)0 - (uintptr_t)ok;
            glds16((const void*)((real & msk) | ((uintptr_t)zero & ~msk)), la + j * 1024);
        }
        char* lb = smem + buf * STAGE + A_BYTES;
#pragma unroll
        for (int j = 0; j < BPW; ++j)
            if (NBP % 8 == 0 || w + 8 * j < NBP) glds16(b_src[j] + (size_t)kt * BK, lb + (w + 8 * j) * 1024);
    };

    // one DMA piece of a stage: n = 0..3 the wave's activation pieces (HBM / L2), 4.. its weight pieces (L2)
    auto stage_piece = [&](int kt, int buf, int n) {
        if (n < 4) {
            const int cc = kt * BK;
            const f16* src; int cs, coff;
            if (cc < p.c0) { src = p.a0; cs = p.c0; coff = cc; } else { src = p.a1; cs = p.c1; coff = cc - p.c0; }
            const bool ok = a_row[n] >= 0;
            const uintptr_t real = (uintptr_t)(src + ((long)a_row[n] * cs + coff + a_chunk[n]));
            const uintptr_t msk = (uintptr_t)0 - (uintptr_t)ok;
            glds16((const void*)((real & msk) | ((uintptr_t)zero & ~msk)), smem + buf * STAGE + (w * 4 + n) * 1024);
        } else {
            const int j = n - 4;
            if (NBP % 8 == 0 || w + 8 * j < NBP) glds16(b_src[j] + (size_t)kt * BK, smem + buf * STAGE + A_BYTES + (w + 8 * j) * 1024);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int swz = (lane >> 1) & 7;
    const int frag_off0 = (lane & 15) * 128 + (((lane >> 4)) ^ swz) * 16;
    const int frag_off1 = (lane & 15) * 128 + ((4 + (lane >> 4)) ^ swz) * 16;

    if ((p.debug & 4096) && w >= 4) __builtin_amdgcn_s_setprio(2);      // (experiment: static priority for waves 4-7)
    // (experiment, debug bit 8192 + count in bits 16..: the first-round workgroups on every other CU start late, so that half the chip is in
    //  its store phase while the other half is in its k loop instead of all CUs storing at once)
    if ((p.debug & 8192) && blockIdx.x < 256 && ((blockIdx.x >> 3) & 1))
        for (int i = 0; i < (p.debug >> 16); ++i) __builtin_amdgcn_s_sleep(127);
    trace_stamp(p.debug, blockIdx.x, 0);
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    trace_stamp(p.debug, blockIdx.x, 1);

    const int KTX = (p.debug & 2) ? 0 : p.KT;
    {
        // PIPELINED form (the FLUX GEMM's structure, csrc/gemm2.hip, on this tile): the per-step barrier sits at three quarters of the step.
        // Before it a wave has ALL of the stage's fragments in registers (the last NT/2 weight fragments are read ahead into a five-slot
        // ring), so after it (i) the buffer just released is re-filled in place with stage kt + 2 while (ii) the remaining MFMAs of step kt
        // issue and (iii) the first fragments of step kt + 1 are read from the other buffer underneath them: no wave stands behind the
        // barrier with an empty matrix pipe waiting for an LDS read.  A stage's DMA pieces are issued one per item: the first NPOST after the
        // barrier of step kt - 2 + ... (activations, HBM), the rest in the first items of the step before its use (weights, L2).
        constexpr int NQ = 2 * NT, NP = 4 + BPW, NPOST = NT / 2, QB = NQ - NPOST, APP = (MT + NPOST - 1) / NPOST;
        static_assert(NQ % 5 == 0 && NP - NPOST <= QB - 1 && NPOST >= 2, "ring / piece placement");
        const bool staging = !(p.debug & 32768);
        auto rd_w = [&](const char* tbx, int q) { const int ks = q / NT, i = q - ks * NT; return *reinterpret_cast<const f16x8*>(tbx + i * 2048 + (ks ? frag_off1 : frag_off0)); };
        // weight fragments are read LA items ahead of their MFMAs into the five-slot ring (slot = item index % 5, continuous across steps)
        constexpr int LA = 2;                                          // (3 and 4 measured the same: profiles/r02_ab_gemm_pipe.txt)
        static_assert(LA >= 1 && LA <= 4, "lookahead");
        auto upto = [](int q) { return q + LA < NQ - 2 ? q + LA : NQ - 2; };      // highest fragment of this stage issued before item q's MFMAs (q < QB)
        f16x8 fa[2][MT], fw[5];
        if (p.KT > 1 && staging) {
#pragma unroll
            for (int n = 0; n < NPOST; ++n) stage_piece(1, 1, n);
        }
        {
            const char* ta0 = smem + (wm * 64) * 128;
            const char* tb0 = smem + A_BYTES + (wn * (BNX / 2)) * 128;
#pragma unroll
            for (int j = 0; j < MT; ++j) fa[0][j] = *reinterpret_cast<const f16x8*>(ta0 + j * 2048 + frag_off0);
#pragma unroll
            for (int r = 0; r < LA; ++r) fw[r] = rd_w(tb0, r);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int kt = 0; kt < KTX; ++kt) {
            const int buf = kt & 1;
            const char* ta = smem + buf * STAGE + (wm * 64) * 128;
            const char* tb = smem + buf * STAGE + A_BYTES + (wn * (BNX / 2)) * 128;
            const char* tan = smem + (buf ^ 1) * STAGE + (wm * 64) * 128;
            const char* tbn = smem + (buf ^ 1) * STAGE + A_BYTES + (wn * (BNX / 2)) * 128;
            const bool more = kt + 1 < p.KT, more2 = kt + 2 < p.KT;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ks = q / NT, i = q - ks * NT;
                if (q == QB && more) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // stage kt + 1 has landed; this wave's reads of stage kt are done
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (q < QB) {                          // (at item QB - 1: the rest of this stage; its last fragment shares that item's ring slot -> after its MFMAs)
                    const int lo = (q == 0 ? LA - 1 : upto(q - 1)) + 1, hi = q == QB - 1 ? NQ - 2 : upto(q);
#pragma unroll
                    for (int r = lo; r <= hi; ++r) fw[r % 5] = rd_w(tb, r);
                } else if (q + LA >= NQ && more) fw[(q + LA) % 5] = rd_w(tbn, q + LA - NQ);
                if (ks == 0 && i >= NT - MT) fa[1][i - (NT - MT)] = *reinterpret_cast<const f16x8*>(ta + (i - (NT - MT)) * 2048 + frag_off1);
                if (q >= QB && more) {
#pragma unroll
                    for (int t = 0; t < APP; ++t) {
                        const int j = (q - QB) * APP + t;
                        if (j < MT) fa[0][j] = *reinterpret_cast<const f16x8*>(tan + j * 2048 + frag_off0);
                    }
                }
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[q % 5], fa[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q == QB - 1) { fw[(NQ - 1) % 5] = rd_w(tb, NQ - 1); __builtin_amdgcn_sched_barrier(0); }
                if (q >= QB) {
                    if (staging && more2 && q - QB < NP) stage_piece(kt + 2, buf, q - QB);
                    __builtin_amdgcn_sched_barrier(0);
                } else if (q < NP - NPOST) {
                    if (staging && more) stage_piece(kt + 1, buf ^ 1, NPOST + q);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                            // the stage buffers become the epilogue patches
    }
    trace_stamp(p.debug, blockIdx.x, 2);
    if (p.debug & 1) { if (acc[0][0][0] == 123.456f) p.out[0] = (f16)1; return; }
    if constexpr (GEGLU) igemm_epilogue<GEGLU, NT, MT, NT>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk + wn * (BNX / 2), lane, smem + w * 11264);
    else igemm_epilogue<GEGLU, NT, MT, NT, LinearRows, (BNX == 320 ? 2 : 1)>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk + wn * (BNX / 2), lane, smem + w * 11264);
    trace_stamp(p.debug, blockIdx.x, 3);
    if (p.debug & 16384) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); trace_stamp(p.debug, blockIdx.x, 4); }
}

template <int BN, bool CONV3, bool GEGLU, int LNM = 0, int EPI = 0>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
    constexpr int NT = BN / 32;          // 16-wide n tiles per wave (wave tile = 64 x BN/2)
    constexpr int MT = 4;
    constexpr int NBI = BN / 32;         // B-tile DMA instructions per wave (8 rows each)
    constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lA = smem;                    // [2][A_BYTES]
    char* const lB = smem + 2 * A_BYTES;      // [2][B_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware, bijective tile id (blocks b and b+8 share an XCD)
    int tm, tn;
    tile_of(blockIdx.x, p.nblk, p.tiles_n, 1, p.pn, tm, tn);
    const int m_blk = tm * BM, n_blk = tn * BN;

    // ---- per-thread staging state: 4 A rows + NBI B rows, one 16-byte chunk each ----------------
    const int pch = lane & 7;                      // physical chunk slot this lane fills
    int a_row_base[4], a_y[4], a_x[4];             // CONV3: pixel base / top-left coords ; else: row offset
    int a_chunk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * (w * 4 + j) + (lane >> 3);
        const int m = m_blk + r;
        a_chunk[j] = pch ^ ((r >> 1) & 7);
        if (CONV3) {
            if (m < p.M) {
                const int b = m / p.HoWo, rem = m - b * p.HoWo, yo = rem / p.Wo, xo = rem - yo * p.Wo;
                a_row_base[j] = b * p.Hi * p.Wi;
                a_y[j] = yo * p.stride - p.pad_lo;
                a_x[j] = xo * p.stride - p.pad_lo;
            } else {
                a_row_base[j] = 0; a_y[j] = -(1 << 20); a_x[j] = 0;
            }
        } else {
            a_row_base[j] = (m < p.M) ? m : -1;
            a_y[j] = a_x[j] = 0;
        }
    }
    const f16* b_src[NBI];
#pragma unroll
    for (int j = 0; j < NBI; ++j) {
        const int q = w * NBI + j;
        const int r = 8 * q + (lane >> 3);
        const int c = pch ^ ((r >> 1) & 7);
        b_src[j] = p.w + (size_t)(n_blk + r) * p.Ktot + c * 8;
    }
    const int Hlim = p.upsample ? 2 * p.Hi : p.Hi, Wlim = p.upsample ? 2 * p.Wi : p.Wi;
    const char* zero = reinterpret_cast<const char*>(g_zero_page) + pch * 16;

    auto stage = [&](int kt, int buf) {
        const int tap = kt / p.cpt;                 // (1x1 with a split-fp16 A operand: "tap" 1 = the lo planes, k steps [KTh, 2 KTh), KTh == cpt)
        const int cc = (kt - tap * p.cpt) * BK;
        const f16* src; int cs, coff;
        if (!CONV3 && tap) {
            if (cc < p.c0) { src = p.a0_lo; cs = p.c0; coff = cc; } else { src = p.a1_lo; cs = p.c1; coff = cc - p.c0; }
        } else if (cc < p.c0) { src = p.a0; cs = p.c0; coff = cc; } else { src = p.a1; cs = p.c1; coff = cc - p.c0; }
        const int dy = CONV3 ? tap / 3 : 0, dx = CONV3 ? tap - 3 * dy : 0;
        char* la = lA + buf * A_BYTES + (w * 4) * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // branch-free select between the gathered pixel and the zero page (an exec-masked
            // branch here would serialise the four DMA issues)
            uintptr_t real; bool ok;
            if (CONV3) {
                int yi = a_y[j] + dy, xi = a_x[j] + dx;
                ok = (yi >= 0) & (yi < Hlim) & (xi >= 0) & (xi < Wlim);
                if (p.upsample) { yi >>= 1; xi >>= 1; }
                real = (uintptr_t)(src + ((long)(a_row_base[j] + yi * p.Wi + xi) * cs + coff + a_chunk[j] * 8));
            } else {
                ok = a_row_base[j] >= 0;
                real = (uintptr_t)(src + ((long)a_row_base[j] * cs + coff + a_chunk[j] * 8));
            }
            const uintptr_t msk = (uintptr_t)0 - (uintptr_t)ok;
            const void* g = (const void*)((real & msk) | ((uintptr_t)zero & ~msk));
            glds16(g, la + j * 1024);
        }
        char* lb = lB + buf * B_BYTES + (w * NBI) * 1024;
#pragma unroll
        for (int j = 0; j < NBI; ++j) glds16(b_src[j] + (size_t)(CONV3 ? kt * BK : cc), lb + j * 1024);      // (1x1: the weight columns wrap with the lo planes)
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane fragment read offset inside a tile: row (lane&15), chunk ((ks*4 + lane>>4) ^ swz)
    const int swz = (lane >> 1) & 7;
    const int frag_off0 = (lane & 15) * 128 + (((lane >> 4)) ^ swz) * 16;        // ks = 0
    const int frag_off1 = (lane & 15) * 128 + ((4 + (lane >> 4)) ^ swz) * 16;    // ks = 1

    // split-K (gridDim.y > 1): this workgroup covers k-steps [kt0, kt1) and leaves raw fp32 partial sums for splitk_reduce_kernel
    const int ksplit = gridDim.y, kper = p.KT / ksplit, kt0 = blockIdx.y * kper, kt1 = kt0 + kper;
    stage(kt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) stage(kt + 1, buf ^ 1);
        const char* ta = lA + buf * A_BYTES + (wm * 64) * 128;
        const char* tb = lB + buf * B_BYTES + (wn * (BN / 2)) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = ks ? frag_off1 : frag_off0;
            f16x8 fa[MT], fw[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const f16x8*>(ta + i * 2048 + fo);
#pragma unroll
            for (int i = 0; i < NT; ++i) fw[i] = *reinterpret_cast<const f16x8*>(tb + i * 2048 + fo);
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (ksplit > 1) {
        float* dst = p.partial + (size_t)blockIdx.y * p.M * p.N;
        const int g = lane >> 4;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m_blk + wm * 64 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                *reinterpret_cast<f32x4*>(dst + (size_t)m * p.N + n_blk + wn * (BN / 2) + i * 16 + g * 4) = acc[i][j];
        }
        return;
    }
    // (no FAST forms of the fp32-patch epilogue here: their registers cost this kernel a wave of occupancy per SIMD, 4 -> 3 / 3 -> 2)
    igemm_epilogue<GEGLU, NT, MT, NT, LinearRows, 1, LNM, false, EPI>(p, acc, LinearRows{m_blk + wm * 64, p.M}, n_blk + wn * (BN / 2), lane, smem + w * 11264, LNM == 1 ? smem + 4 * 11264 + w * 1280 : nullptr);
}

template <int BN, bool CONV3, bool GEGLU, int LNM = 0, int EPI = 0>
int launch_variant(const IgemmParams& p, hipStream_t s, int splits = 1) {
    constexpr size_t lds = 2 * (BM * BK * 2 + BN * BK * 2);
    static bool configured = false;
    auto kfn = igemm_kernel<BN, CONV3, GEGLU, LNM, EPI>;
    if (!configured) {
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(kfn, dim3(p.nblk, splits), dim3(256), lds, s, p);
    CS_CHECK_LAUNCH();
    if (splits > 1) return launch_splitk_reduce(p, (const float*)p.partial, splits, s);
    return CS_OK;
}

}  // namespace

// extra dynamic LDS of the folded-LayerNorm consumer instantiations (ln_tile_prologue): 256 rows x 8 B + the (b' | s) tables
constexpr size_t LN_LDS_W8 = 256 * 8 + 2 * 2 * 160 * 4, LN_LDS_LW = 256 * 8 + 2 * 160 * 4;

// tile_of's pn for a launch of tiles_m x tiles_n tiles that reads a_bytes of activations and w_bytes of weights once each algorithmically: per-XCD L2s mean the
// contiguous order fetches a + 8 w, an (8 / pn) x pn grid pn a + (8 / pn) w.  Switch only for a clear gain (the contiguous order has the banded walk, IgemmParams::gm).
static int choose_xcd_grid(int tiles_m, int tiles_n, double a_bytes, double w_bytes) {
    if (!tune().xcd_grid) return 0;
    int best = 0; double best_cost = 0.85 * (a_bytes + 8.0 * w_bytes);
    for (int pn = 2; pn <= 8; pn *= 2) {
        const int px = 8 / pn;
        if (tiles_n % pn || tiles_m % px) continue;
        const double cost = pn * a_bytes + px * w_bytes;
        if (cost < best_cost) { best_cost = cost; best = pn; }
    }
    return best;
}

double igemm_flops(const IgemmArgs& a) {
    const double M = (double)a.B * a.Ho * a.Wo;
    return 2.0 * M * a.N * (double)a.taps * (a.c0 + a.c1);
}

// one halo-conv instantiation per (upsample, tile width, k halves)
template <bool UP, int BN, int KH, int SCHED>
static int launch_halo_sched(const HaloParams& h, dim3 grid, size_t lds, hipStream_t s) {
    static bool configured = false;
    if (!configured) {
        constexpr size_t max_lds = 2 * (HALO_ROWS_MAX * 128) + 3 * (160 * 128);
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_halo_kernel<UP, BN, KH, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds));
        configured = true;
    }
    hipLaunchKernelGGL((conv3_halo_kernel<UP, BN, KH, SCHED>), grid, dim3(512), lds, s, h);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
// the k32-step kernels (KH = 2: BN 320 / 256) run schedule 2 (priority + interleaved DMA for the staggered group: -6 ... -11 %, profiles/r02_ab_conv_sched.txt), the
// k64-step kernels (BN 160 / 128) the lock-step schedule 0 (schedule 2 cost them 3 %); the other combinations were measured in round 2 and are not compiled any more
template <bool UP, int BN, int KH>
static int launch_halo(const HaloParams& h, dim3 grid, size_t lds, hipStream_t s) { return launch_halo_sched<UP, BN, KH, KH == 2 ? 2 : 0>(h, grid, lds, s); }

struct LaunchInfo { bool gn_done = false; int row_groups = 0; };    // what the chosen kernel's epilogue left: GroupNorm statistics / row statistics in N / row_groups-column groups
static int launch_igemm_impl(const IgemmArgs& a, hipStream_t s, LaunchInfo* li) {
    const int cin = a.c0 + a.c1;
    if (!a.a0 || !a.w || !a.out) CS_FAIL(CS_E_ARG, "igemm: a0, w, out required");
    if (a.taps != 1 && a.taps != 9) CS_FAIL(CS_E_ARG, "igemm: taps must be 1 or 9");
    if (cin % BK || a.c0 % BK || (a.c1 && !a.a1)) CS_FAIL(CS_E_SHAPE, "igemm: channels (%d,%d) must be multiples of %d", a.c0, a.c1, BK);
    if (a.B <= 0 || a.Ho <= 0 || a.Wo <= 0) return (a.B < 0) ? CS_E_SHAPE : CS_OK;
    if (a.taps == 1 && (a.stride != 1 || a.upsample || a.Hi != a.Ho || a.Wi != a.Wo)) CS_FAIL(CS_E_ARG, "igemm: 1x1 needs stride 1, no upsample");
    if (a.geglu && (a.N % 256) && (a.N % 320)) CS_FAIL(CS_E_SHAPE, "igemm: GEGLU needs N %% 256 == 0 or N %% 320 == 0 (N=%d)", a.N);
    IgemmParams p;
    p.a0 = a.a0; p.a1 = a.a1; p.c0 = a.c0; p.c1 = a.c1;
    p.Hi = a.Hi; p.Wi = a.Wi; p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo;
    p.stride = a.stride; p.upsample = a.upsample; p.pad_lo = a.pad_after_only ? 0 : 1;
    if (a.pad_after_only && !(a.taps == 9 && a.stride == 2)) CS_FAIL(CS_E_ARG, "igemm: pad_after_only is the stride-2 3x3 form");
    p.M = a.B * a.Ho * a.Wo; p.N = a.N; p.cpt = cin / BK; p.KT = a.taps * p.cpt; p.Ktot = a.taps * cin;
    p.w = a.w; p.bias = a.bias; p.temb = a.temb; p.temb_stride = a.temb_stride; p.res = a.res; p.out = a.out;
    p.res_lo = a.res ? a.res_lo : nullptr; p.out_lo = a.out_lo; p.lo8 = (a.lo8 && (p.res_lo || p.out_lo)) ? 1 : 0;
    const bool conv3_early = a.taps == 9;
    p.row_stats = a.row_stats;
    p.ln_stats = a.ln_stats; p.ln_groups = a.ln_groups; p.ln_inv_c = 1.0f / (float)cin; p.ln_eps = a.ln_eps; p.ln_s = a.ln_s; p.ln_b = a.ln_b;
    if (a.ln_stats) {
        if (a.taps != 1 || a.c1 || a.temb || a.res || a.out_lo || a.row_stats || a.gn_stats)
            CS_FAIL(CS_E_ARG, "igemm: a folded LayerNorm (ln_stats) is the epilogue of a plain linear layer (one source, no temb / residual / lo plane / statistics)");
        if (!a.ln_s || !a.ln_b || a.ln_groups < 1) CS_FAIL(CS_E_ARG, "igemm: ln_stats needs ln_s, ln_b and ln_groups >= 1");
    }
    if (a.row_stats && (a.geglu || a.gn_stats)) CS_FAIL(CS_E_ARG, "igemm: row_stats excludes GEGLU and gn_stats");
    if (a.row_stats && !a.row_stats_groups) CS_FAIL(CS_E_ARG, "igemm: row_stats needs row_stats_groups (the layout the chosen kernel wrote is returned there)");
    p.a0_lo = a.a0_lo; p.a1_lo = a.c1 ? a.a1_lo : nullptr; p.KTh = p.KT;
    const int lnm = a.ln_stats ? 1 : (a.row_stats && !conv3_early ? 2 : 0);      // epilogue instantiation: folded-LayerNorm consumer / row-statistics producer / neither
    const bool split_a = a.a0_lo != nullptr;
    if (split_a) {
        if (a.taps != 1 || a.geglu) CS_FAIL(CS_E_ARG, "igemm: a split-fp16 A operand (a0_lo) is built for 1x1 / linear layers without GEGLU");
        if (a.c1 && !a.a1_lo) CS_FAIL(CS_E_ARG, "igemm: a0_lo without a1_lo");
        p.KT = 2 * p.KTh;                                  // Ktot (row length of w) stays cin: the lo k steps re-read the same weight columns
    }
    if (a.geglu && a.out_lo) CS_FAIL(CS_E_ARG, "igemm: the GEGLU epilogue writes no lo plane");
    if (p.lo8 && (a.taps != 1 || a.temb || a.geglu || a.ln_stats)) CS_FAIL(CS_E_ARG, "igemm: 8-bit lo planes (lo8) are built for plain 1x1 / linear layers (the transformer hidden state)");
    p.epi_fast = tune().epi_fast;
    // the epilogue family (igemm_epilogue_f32, EPI): 1 = byte lo planes, or a residual with an fp16 lo plane and no lo plane out (FAST 4) -- separate kernel instantiations
    const int epi = (a.taps == 1 && !a.geglu && !a.ln_stats && !a.temb) ? (p.lo8 ? 1 : (p.res && p.res_lo && !p.out_lo && (p.epi_fast & 2)) ? 2 : 0) : 0;
    p.debug = tune().debug; p.partial = nullptr;
    // GroupNorm statistics of the output: by the epilogue where the chosen kernel runs one (not the split-K forms), else by the caller below
    const bool stats_ok = a.gn_stats && !a.geglu && p.HoWo % 64 == 0 && tune().gn_fuse != 0;
    p.gn_stats = stats_ok ? a.gn_stats : nullptr;
    p.gm = 1; p.pn = 0;
    const double a_bytes = 2.0 * a.B * a.Hi * a.Wi * cin, w_bytes = 2.0 * a.N * a.taps * cin;
    const int tiles_m = (p.M + BM - 1) / BM;
    int bn;
    if (a.N % 128 == 0) bn = 128;
    else if (a.N % 160 == 0) bn = 160;
    else CS_FAIL(CS_E_SHAPE, "igemm: N=%d must be a multiple of 128 or 160", a.N);
    p.tiles_n = a.N / bn; p.nblk = tiles_m * p.tiles_n;
    const bool conv3 = a.taps == 9;
    const int use_halo = tune().halo;   // 0 = never, 1 = when it pays, 2 = whenever the shape allows (tests)
    const int Wo = a.upsample ? 2 * a.Wi : a.Wi, Ho = a.upsample ? 2 * a.Hi : a.Hi;
    int hbn = a.N % 160 == 0 ? 160 : (a.N % 128 == 0 ? 128 : 0);
    // 256 x 320 tiles (k32 inner step) when they still fill the chip: SD1.5's 320- and 640-wide layers at 64 x 64 / 32 x 32 (tune().halo 3 forces,
    // 4 forbids: tests / A-B)
    bool wide = false;
    // patch geometry: TW = min(Wo, 16) in {8, 16}; TH = min(Ho, 256 / TW); both powers of two dividing the image
    const int TW = Wo >= 16 ? 16 : Wo, THmax = TW ? 256 / TW : 0, TH = Ho < THmax ? Ho : THmax;
    const bool pow2 = TW == 8 || TW == 16;
    const bool geom_ok = pow2 && TH >= 2 && (TH & (TH - 1)) == 0 && Wo % TW == 0 && Ho % TH == 0 && Ho == a.Ho && Wo == a.Wo &&
                         (!a.upsample || (TH % 2 == 0 && TW % 2 == 0 && TW / 2 + 2 >= 10));
    // round 6: nearest-x2 upsample + 3x3 conv in the sub-pixel form (conv3_lw_kernel<.., SUB>): tiles over INPUT pixels, four phases, 4 of 9 taps each with the
    // filter's taps pre-summed (a.w_up_sub from conv_up_fold_pack_host: one more fp16 rounding of the weights, which is why the caller decides -- tune().up_fold)
    if (conv3 && a.upsample && a.w_up_sub && a.stride == 1 && a.c1 == 0 && (a.N % 160 == 0 || a.N % 128 == 0) && !a.geglu && !a.res && !a.temb && tune().conv_lw != 0 &&
        cin / BK <= LW_ZERO_CHUNKS && Ho == a.Ho && Wo == a.Wo && !(tune().debug & 16384)) {
        const int sbn = a.N % 160 == 0 ? 160 : 128;                      // (the VAE decoder's 256 / 512-wide upsamplers: 128-column tiles)
        const bool f1 = a.Wi % 16 == 0 && a.Hi % 16 == 0, f2 = a.Wi == 8 && a.Hi == 8 && sbn == 160;
        if (f1 || f2) {
            const int TWi = f1 ? 16 : 8, TRW = f1 ? 256 : 64, IPT = 256 / TRW;
            const int PX = a.Wi / TWi, PY = a.Hi / TWi, PP = PX * PY;
            const int tiles_lo = PP == 1 ? (a.B + IPT - 1) / IPT : a.B * PP, tiles_m = 4 * tiles_lo, tiles_n = a.N / sbn;
            HaloParams h;
            h.HALO_W = TWi + 2; h.HALO_IMG = (TWi + 2) * (TWi + 2); h.NHALO = IPT * h.HALO_IMG; h.NQ = (h.NHALO + 7) / 8;
            h.e = p; h.e.w = a.w_up_sub; h.e.tiles_n = tiles_n; h.e.nblk = tiles_m * tiles_n; h.e.row_stats = nullptr; h.e.pn = choose_xcd_grid(tiles_m, tiles_n, a_bytes, w_bytes * 16.0 / 9.0);
            h.x = a.a0; h.w = a.w_up_sub; h.Cin = cin; h.H = a.Hi; h.W = a.Wi; h.B = a.B; h.NC = cin / BK; h.Ho = Ho; h.Wo = Wo;
            h.tw_shift = f1 ? 4 : 3; h.trw_shift = f1 ? 8 : 6; h.PX = PX; h.PP = PP; h.splits = 1; h.partial = nullptr; h.sched = 0;
            typedef void (*lw_fn)(HaloParams);
            static const lw_fn sub[3] = {conv3_lw_kernel<false, 160, false, 1, true>, conv3_lw_kernel<false, 160, false, 2, true>, conv3_lw_kernel<false, 128, false, 1, true>};
            static bool configured_sub = false;
            if (!configured_sub) {
                for (lw_fn f : sub)
                    CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (HALO_ROWS_MAX * 128) + 3 * (160 * 128))));
                configured_sub = true;
            }
            const size_t llw = 2 * ((size_t)h.NQ * 1024) + 3 * ((size_t)sbn * 128);
            hipLaunchKernelGGL(sub[sbn == 128 ? 2 : f1 ? 0 : 1], dim3(h.e.nblk, 1), dim3(512), llw, s, h);
            CS_CHECK_LAUNCH();
            li->gn_done = stats_ok;
            return CS_OK;
        }
    }
    if (use_halo && conv3 && a.stride == 1 && a.c1 == 0 && hbn && !a.geglu && geom_ok) {
        const int TRW = TH * TW, IPT = 256 / TRW;                       // output pixels per image in a tile; images per tile
        const int PX = Wo / TW, PY = Ho / TH, PP = PX * PY;
        const int tiles_m = PP == 1 ? (a.B + IPT - 1) / IPT : a.B * PP;
        if (a.N % 320 == 0 && tune().halo != 4 && (tiles_m * (a.N / 320) >= 192 || tune().halo == 3)) { hbn = 320; wide = true; }
        else if (a.N % 256 == 0 && tune().halo != 4 && (tiles_m * (a.N / 256) >= 192 || tune().halo == 3)) { hbn = 256; wide = true; }   // VAE: 256 / 512 channels
        // round 3: loader-wave kernel (256 pixels x 160 channels per workgroup) wherever the channel count allows it
        // (conv3_lw_kernel's padded halo rows read g_zero_region + chunk offset: the region covers LW_ZERO_CHUNKS 64-channel chunks, wider inputs take the halo kernels)
        const bool lw_cin_ok = cin / BK <= LW_ZERO_CHUNKS;
        const bool lw160 = tune().conv_lw != 0 && a.N % 160 == 0 && lw_cin_ok;
        const bool lw128 = tune().conv_lw != 0 && tune().conv_lw != 3 && !lw160 && a.N % 128 == 0 && lw_cin_ok;        // the VAE's widths 128 / 256 / 512 (conv_lw = 3: BN 160 only)
        const bool lw = lw160 || lw128;
        if (lw) { hbn = lw160 ? 160 : 128; wide = false; }
        const int tiles_n = a.N / hbn;
        const int NC = cin / BK;
        int splits = 1;
        if (tiles_m * tiles_n < 160 && a.splitk_ws) {          // small images: split the channel chunks to fill the chip
            while (splits < 8 && tiles_m * tiles_n * splits < 192 && NC % (splits * 2) == 0 &&
                   (size_t)(splits * 2) * p.M * a.N * sizeof(float) <= a.splitk_ws_bytes) splits *= 2;
        }
        const bool pays = tiles_m * tiles_n * splits >= 160;
        const int tin_w = a.upsample ? TW / 2 : TW, tin_h = a.upsample ? TH / 2 : TH;
        HaloParams h;
        h.HALO_W = tin_w + 2; h.HALO_IMG = (tin_h + 2) * (tin_w + 2); h.NHALO = IPT * h.HALO_IMG; h.NQ = (h.NHALO + 7) / 8;
        if ((pays || use_halo == 2 || use_halo == 3) && h.NHALO <= HALO_ROWS_MAX && h.NQ <= 64 && (PP == 1 || IPT == 1)) {
            h.e = p; h.e.tiles_n = tiles_n; h.e.nblk = tiles_m * tiles_n; h.e.row_stats = nullptr; h.e.pn = choose_xcd_grid(tiles_m, tiles_n, a_bytes, w_bytes);       // (row statistics of a conv output: the pass in launch_igemm)
            h.x = a.a0; h.w = a.w; h.Cin = cin; h.H = a.Hi; h.W = a.Wi; h.B = a.B; h.NC = NC; h.Ho = Ho; h.Wo = Wo;
            h.tw_shift = TW == 16 ? 4 : 3; h.trw_shift = 0; while ((1 << h.trw_shift) < TRW) ++h.trw_shift;
            h.PX = PX; h.PP = PP;
            // schedule: the k32-step kernels (BN 320 / 256) gain 6-11 % from priority + interleaved DMA (A/B on one box, tools/ab_convsched.sh);
            // the k64-step kernels (BN 160 / 128: 16 x 16 and 8 x 8 images) measured 0.207 -> 0.213 ms with it and keep the lock-step schedule
            h.splits = splits; h.partial = a.splitk_ws; h.sched = wide ? 2 : 0;
            const size_t l = 2 * (HALO_ROWS_MAX * 128) + 3 * ((size_t)hbn * (wide ? 64 : 128));
            const dim3 grid(h.e.nblk, splits);
            int rc;
            if (lw) {
                constexpr size_t llw_max = 2 * (HALO_ROWS_MAX * 128) + 3 * (160 * 128);      // = 160 KiB exactly: the whole LDS of a CU
                const size_t llw = 2 * ((size_t)h.NQ * 1024) + 3 * ((size_t)hbn * 128);
                // FAST: plain conv on 16 x 16 patches (every UNet level down to 16 x 16, the VAE): immediate-offset LDS addressing
                const bool fast = TW == 16 && TH == 16 && tune().conv_lw != 2 && (a.upsample ? (h.HALO_W == 10 && h.NQ == 13) : (h.HALO_W == 18 && h.NQ == 41));
                const bool fast8 = hbn == 160 && TW == 8 && TH == 8 && !a.upsample && IPT == 4 && PP == 1 && h.HALO_W == 10 && h.NQ == 50 && tune().conv_lw != 2 &&
                                   !((tune().debug & 16384) != 0);
                typedef void (*lw_fn)(HaloParams);
                static const lw_fn variants[11] = {conv3_lw_kernel<false, 160, false, true>, conv3_lw_kernel<false, 160>, conv3_lw_kernel<true, 160>,
                                                   conv3_lw_kernel<false, 160, true, true>, conv3_lw_kernel<false, 160, true, false>, conv3_lw_kernel<true, 160, false, true>,
                                                   conv3_lw_kernel<false, 128, false, true>, conv3_lw_kernel<false, 128>, conv3_lw_kernel<true, 128>, conv3_lw_kernel<true, 128, false, true>,
                                                   conv3_lw_kernel<false, 160, false, 2>};
                static bool configured_lw = false;
                if (!configured_lw) {
                    for (lw_fn f : variants)
                        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)llw_max));
                    configured_lw = true;
                }
                auto launch = [&](lw_fn kfn) -> int { hipLaunchKernelGGL(kfn, grid, dim3(512), llw, s, h); return CS_OK; };
                const bool trace = (tune().debug & 16384) && !a.upsample;    // timing experiments: the stamped instantiations (plain conv only)
                if (fast8) rc = launch(variants[10]);
                else if (hbn == 128) rc = launch(variants[a.upsample ? (fast ? 9 : 8) : (fast ? 6 : 7)]);
                else if (trace) rc = launch(variants[fast ? 3 : 4]);
                else if (a.upsample) rc = launch(variants[fast ? 5 : 2]);
                else rc = launch(variants[fast ? 0 : 1]);
                if (rc != CS_OK) return rc;
                rc = CS_OK;
            } else if (wide && hbn == 320) rc = a.upsample ? launch_halo<true, 320, 2>(h, grid, l, s) : launch_halo<false, 320, 2>(h, grid, l, s);
            else if (wide) rc = a.upsample ? launch_halo<true, 256, 2>(h, grid, l, s) : launch_halo<false, 256, 2>(h, grid, l, s);
            else if (hbn == 160) rc = a.upsample ? launch_halo<true, 160, 1>(h, grid, l, s) : launch_halo<false, 160, 1>(h, grid, l, s);
            else rc = a.upsample ? launch_halo<true, 128, 1>(h, grid, l, s) : launch_halo<false, 128, 1>(h, grid, l, s);
            if (rc != CS_OK) return rc;
            CS_CHECK_LAUNCH();
            li->gn_done = stats_ok && (splits == 1 || reduce_leaves_stats(h.e));
            if (splits > 1) return launch_splitk_reduce(h.e, (const float*)a.splitk_ws, splits, s);
            return CS_OK;
        }
    }
    if (a.geglu && conv3) CS_FAIL(CS_E_ARG, "igemm: GEGLU epilogue only for linear layers");
    if (tune().biggemm && !conv3 && a.N % 320 == 0) {
        const int tiles_m = (p.M + 255) / 256, tn = a.N / 320;
        if (tiles_m * tn >= 192 || tune().biggemm == 2) {
            p.tiles_n = tn; p.nblk = tiles_m * tn; p.pn = choose_xcd_grid(tiles_m, tn, a_bytes, w_bytes); li->gn_done = stats_ok; li->row_groups = a.N / 160;      // gemm_w8 / gemm_big<., 320>: 64 x 160 wave tiles
            p.gm = tune().gemm_gm >= 0 ? (tune().gemm_gm > 1 ? tune().gemm_gm : 1) : (tn >= 12 ? 4 : 1);
            constexpr size_t lds = 2 * (256 * BK * 2 + 320 * BK * 2);
            static bool configured = false;
            if (!configured) {
                CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<false, 320>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<true, 320>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                configured = true;
            }
            // round 3: the hand-scheduled k loop (gemm_w8_kernel) whenever 32-bit byte offsets reach every operand row
            const bool off32 = (double)p.M * (a.c0 > a.c1 ? a.c0 : a.c1) * 2 < 4.0e9 && (double)a.N * p.Ktot * 2 < 4.0e9;
            if ((split_a || lnm || epi == 1) && !(tune().gemm_w8 && off32 && !tune().debug)) goto generic_tiles;       // gemm_big_kernel has no lo-plane staging, no LayerNorm epilogues, no byte planes
            if (tune().gemm_w8 && off32 && !tune().debug) {
                typedef void (*w8_fn)(IgemmParams);
                static const w8_fn w8[8] = {gemm_w8_kernel<false, 0>, gemm_w8_kernel<true, 0>, gemm_w8_kernel<false, 1>, gemm_w8_kernel<true, 1>, gemm_w8_kernel<false, 2>,
                                            gemm_w8_kernel<false, 0, 1>, gemm_w8_kernel<false, 2, 1>, gemm_w8_kernel<false, 0, 2>};
                static bool configured_w8 = false;
                if (!configured_w8) {
                    for (w8_fn f : w8) CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + LN_LDS_W8)));
                    configured_w8 = true;
                }
                hipLaunchKernelGGL(w8[epi == 1 ? (lnm == 2 ? 6 : 5) : (epi == 2 && lnm == 0) ? 7 : lnm == 2 ? 4 : 2 * lnm + (a.geglu ? 1 : 0)], dim3(p.nblk), dim3(512), lds + LN_LDS_W8, s, p);
                CS_CHECK_LAUNCH();
                return CS_OK;
            }
            if (a.geglu) hipLaunchKernelGGL((gemm_big_kernel<true, 320>), dim3(p.nblk), dim3(512), lds, s, p);
            else hipLaunchKernelGGL((gemm_big_kernel<false, 320>), dim3(p.nblk), dim3(512), lds, s, p);
            CS_CHECK_LAUNCH();
            return CS_OK;
        }
    }
    if (tune().biggemm && !conv3 && !a.geglu && a.N % 160 == 0) {       // too few 256 x 320 tiles: 256 x 160 tiles, same 8-wave structure
        const int tiles_m = (p.M + 255) / 256, tn = a.N / 160;
        if (tiles_m * tn >= 192 || tune().biggemm == 3) {
            p.tiles_n = tn; p.nblk = tiles_m * tn; p.pn = choose_xcd_grid(tiles_m, tn, a_bytes, w_bytes); li->gn_done = stats_ok;
            li->row_groups = (tune().gemm_lw && !tune().debug) ? a.N / 160 : a.N / 80;       // gemm_lw: 64 x 160 wave tiles; gemm_big<., 160>: 64 x 80
            p.gm = tune().gemm_gm >= 0 ? (tune().gemm_gm > 1 ? tune().gemm_gm : 1) : (tn >= 12 ? 4 : 1);
            constexpr size_t lds = 2 * (256 * BK * 2 + 160 * BK * 2);
            static bool configured = false;
            if (!configured) {
                CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_kernel<false, 160>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                configured = true;
            }
            if ((split_a || lnm || epi == 1) && !(tune().gemm_lw && !tune().debug)) goto generic_tiles;
            if (tune().gemm_lw && !tune().debug) {                    // round 3: the loader-wave form (three 52 KB stages)
                constexpr size_t lds_lw = 3 * (256 * BK * 2 + 160 * BK * 2);
                typedef void (*lw_fn)(IgemmParams);
                static const lw_fn lwk[6] = {gemm_lw_kernel<0>, gemm_lw_kernel<1>, gemm_lw_kernel<2>, gemm_lw_kernel<0, 1>, gemm_lw_kernel<2, 1>, gemm_lw_kernel<0, 2>};
                static bool configured_lw = false;
                if (!configured_lw) {
                    for (lw_fn f : lwk) CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_lw + LN_LDS_LW)));
                    configured_lw = true;
                }
                hipLaunchKernelGGL(lwk[epi == 1 ? (lnm == 2 ? 4 : 3) : (epi == 2 && lnm == 0) ? 5 : lnm], dim3(p.nblk), dim3(512), lds_lw + LN_LDS_LW, s, p);
                CS_CHECK_LAUNCH();
                return CS_OK;
            }
            hipLaunchKernelGGL((gemm_big_kernel<false, 160>), dim3(p.nblk), dim3(512), lds, s, p);
            CS_CHECK_LAUNCH();
            return CS_OK;
        }
    }
generic_tiles:
    p.tiles_n = a.N / bn; p.nblk = tiles_m * p.tiles_n; p.gm = 1; p.pn = choose_xcd_grid(tiles_m, p.tiles_n, a_bytes, w_bytes); li->gn_done = false; li->row_groups = 0;
    if (a.geglu) {
        return lnm == 1 ? launch_variant<128, false, true, 1>(p, s) : launch_variant<128, false, true>(p, s);
    }
    // few tiles and a long k loop (stride-2 convs into the 16 x 16 / 8 x 8 levels, the 8 x 8 linears): split K to fill the chip
    int splits = 1;
    if (a.splitk_ws && !a.geglu && p.nblk < 192) {
        while (splits < 8 && p.nblk * splits < 256 && p.KT % (splits * 2) == 0 && p.KT / (splits * 2) >= 8 &&
               (size_t)(splits * 2) * p.M * a.N * sizeof(float) <= a.splitk_ws_bytes) splits *= 2;
        if (splits > 1) p.partial = a.splitk_ws;
    }
    li->gn_done = stats_ok && (splits == 1 || reduce_leaves_stats(p));
    li->row_groups = (splits == 1 && !conv3) ? a.N / (bn / 2) : 0;          // igemm_kernel: 64 x bn / 2 wave tiles; the split-K reduce leaves no row statistics
    if (conv3) return bn == 128 ? launch_variant<128, true, false>(p, s, splits) : launch_variant<160, true, false>(p, s, splits);
    const int lk = splits > 1 ? 0 : lnm;            // (split-K: raw partial sums leave the main kernel; the reduce kernel applies a folded LayerNorm itself)
    if (epi == 1 && splits == 1) {                  // byte planes (split-K: the reduce kernel handles the planes, the main kernel writes raw partial sums)
        if (bn == 128) return lk == 2 ? launch_variant<128, false, false, 2, 1>(p, s, splits) : launch_variant<128, false, false, 0, 1>(p, s, splits);
        return lk == 2 ? launch_variant<160, false, false, 2, 1>(p, s, splits) : launch_variant<160, false, false, 0, 1>(p, s, splits);
    }
    if (bn == 128) return lk == 1 ? launch_variant<128, false, false, 1>(p, s, splits) : lk == 2 ? launch_variant<128, false, false, 2>(p, s, splits) : launch_variant<128, false, false>(p, s, splits);
    return lk == 1 ? launch_variant<160, false, false, 1>(p, s, splits) : lk == 2 ? launch_variant<160, false, false, 2>(p, s, splits) : launch_variant<160, false, false>(p, s, splits);
}

int launch_igemm(const IgemmArgs& a, hipStream_t s) {
    LaunchInfo li;
    const int rc = launch_igemm_impl(a, s, &li);
    if (rc != CS_OK) return rc;
    if (a.row_stats) {                              // row statistics for a LayerNorm folded into the consumer: by the epilogue, else a pass over the output
        if (li.row_groups > 0) *a.row_stats_groups = li.row_groups;
        else {
            const int M = a.B * a.Ho * a.Wo;
            hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, (const f16*)a.out, (const f16*)a.out_lo, M, a.N, a.row_stats, a.lo8);
            CS_CHECK_LAUNCH();
            *a.row_stats_groups = 1;
        }
    }
    if (!a.gn_stats || li.gn_done) return rc;
    // the chosen kernel has no statistics epilogue (split-K forms) or the knob is off: a statistics pass over the output, same layout
    if (a.geglu || (a.Ho * a.Wo) % 64) CS_FAIL(CS_E_ARG, "igemm: gn_stats needs Ho * Wo %% 64 == 0 and no GEGLU");
    return launch_gn_stats64(a.out, a.B, a.Ho * a.Wo, a.N, a.gn_stats, s);
}

// cs_unet_calibrate_ln_fold: *dst += sum over rows of mean^2 / (var + eps) from the row statistics rs[M][G][2] a producer left (IgemmArgs::row_stats) -- how many sigma
// the rows of a hidden state sit away from zero.  A threshold decision: the order of the float atomics does not matter.
__global__ __launch_bounds__(256) void ln_dc_ratio_kernel(const float* __restrict__ rs, int M, int G, float inv_c, float eps, float* __restrict__ dst) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        float s1, s2;
        ln_row_moments(rs + (size_t)m * G * 2, G, s1, s2);
        const float mean = s1 * inv_c, var = fmaxf(s2 * inv_c - mean * mean, 0.f);
        acc += mean * mean / (var + eps);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
}

int launch_ln_dc_ratio(const float* rs, int M, int G, int C, float eps, float* dst, hipStream_t s) {
    if (!rs || !dst || M <= 0 || G < 1 || C <= 0) CS_FAIL(CS_E_ARG, "ln_dc_ratio: bad arguments");
    int blocks = (M + 255) / 256; if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ln_dc_ratio_kernel, dim3(blocks), dim3(256), 0, s, rs, M, G, 1.0f / (float)C, eps, dst);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_row_stats(const f16* x, const f16* x_lo, int M, int C, float* stats, hipStream_t s, int lo8) {
    if (!x || !stats || C % 8) CS_FAIL(CS_E_ARG, "row_stats: x, stats required; C %% 8 == 0");
    if (M <= 0) return M < 0 ? CS_E_SHAPE : CS_OK;
    hipLaunchKernelGGL(row_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, x_lo, M, C, stats, lo8);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

// timing experiments only: copy the per-workgroup stamps of the last gemm_big_kernel launches run with debug bit 16384 to host memory
int debug_trace_read(void* dst, size_t bytes) {
    if (bytes > sizeof(unsigned long long) * CS_TRACE_SLOTS * CS_TRACE_W) bytes = sizeof(unsigned long long) * CS_TRACE_SLOTS * CS_TRACE_W;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? CS_OK : CS_E_HIP;
}
