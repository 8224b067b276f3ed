// Fused cross-attention sub-block of the SD1.5 transformer block at C = 320 (8 heads x 40), the 64 x 64 level:
//
//     h_out = h + to_out( softmax(scale * to_q(LayerNorm2(h)) K^T) V ) + b_out        (K, V: the cached projections of the <= 80 text keys)
//
// replaces four kernels of the unfused executor (ln_kernel, gemm_big to_q, attn_kernel with 77 keys, gemm_big to_out + residual) whose
// only job besides ~54 MFLOP per 128 rows is to move three [M, 320] intermediates through HBM (LayerNorm output, q, attention output:
// 6 x 84 MB per block at batch 32).  Here a workgroup owns 128 token rows and keeps all three in LDS:
//
//   phase 0  rows -> registers -> LayerNorm (fp32 statistics, two-pass variance) -> fp16 tile XT in LDS, in the k-block layout the
//            GEMM fragment reads want: [5 blocks of 64 columns][128 rows][128 B], 16-byte chunk index XOR (row >> 1) & 7;
//            the V rows of the sample are fetched into registers in the same phase and written to LDS after phase 1 ([96 key rows][672 B]);
//   phase 1  q = XT Wq^T : 5 k64-steps, Wq streamed by LDS-DMA through two 40 KB stages (the first one issued before phase 0), 8 waves as
//            2 (rows) x 4 (columns), wave tile 64 x 80; accumulators -> fp16 (rounded as stored q, then pre-scaled by scale * log2 e and
//            rounded again, the unfused kernels' two rounding points) -> written over XT;
//   phase 2  (K stays in global / L2 too: a key row's 8 consecutive channels are one 16-byte load in MFMA A layout);
//   phase 3  each wave takes 16 query rows and loops over the 8 heads: S^T = K Q^T (10 MFMAs), masked softmax over <= 80 keys in
//            registers, O^T = V^T P^T (9 MFMAs, V^T through the transposing LDS read), normalised O -> fp16 -> written over the
//            head's q columns of XT (dead by then);
//   phase 3' K fragments of the NEXT head are loaded while the current head computes;
//   phase 4  out = XT Wo^T as phase 1, + bias -> fp16 -> LDS patch -> + residual (h re-read, L2-hot) -> fp16 -> 640-byte row stores.
//
// LDS: XT 80 KB + two weight stages 80 KB = 160 KB exactly (one workgroup per CU); the V tile reuses the stages, the epilogue patch XT and the first 2 KB of stage 0.
// The arithmetic class (fp16 storage points, fp32 accumulation) is the unfused path's; the softmax here subtracts the true row maximum
// (the unfused kernel's max-free steady state is for thousands of keys).
#include "ops.h"


namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}
typedef __fp16 hf4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef __attribute__((address_space(3))) hf4* lds_hf4_ptr;
__device__ __forceinline__ u32x2 tr_read(const char* lds_addr) {
    hf4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_hf4_ptr)lds_addr);
    union { hf4 a; u32x2 b; } u; u.a = r;
    return u.b;
}
__device__ __forceinline__ unsigned pk(float a, float b) { union { f16x2 v; unsigned u; } x; x.v = f16x2{(f16)a, (f16)b}; return x.u; }
__device__ __forceinline__ f16x8 frag_of(u32x4 v) { union { u32x4 u; f16x8 f; } x; x.u = v; return x.f; }

struct XattnParams {
    const f16* h;        // [M_in rows][320] block input (residual stream)
    f16* out;            // [M rows][320]; may alias h
    const f16* h_lo; f16* out_lo;   // SPLIT: lo planes of the split-fp16 residual stream (XattnArgs::h_lo / out_lo)
    float* row_stats;    // optional [M][2]: (sum, sum of squares) of every output row (XattnArgs::row_stats)
    const f16* ln_g; const f16* ln_b; float ln_eps;
    const f16* wq;       // [320][320]
    const f16* wo; const f16* bo;
    const f16* kv;       // [B][Nk][640]: k = cols 0..319, v = cols 320..639
    int M, HW, Nk;       // rows of this launch, rows per sample, keys (<= 80)
    float c;             // scale * log2(e)
    int debug;           // timing experiments only (results wrong): bit 0 skip the attention phase, 1 skip the to_out GEMM, 2 skip the to_q GEMM,
                         // 3 skip the LayerNorm phase's loads, 4 skip the epilogue's residual loads and stores
};

constexpr int C = 320, DH = 40, NH = 8, TM = 128;
constexpr int XT_BYTES = 5 * TM * 128;            // 81920
constexpr int WST = 320 * 128;                    // 40960 per weight stage
constexpr int VS = 672, VROWS = 96;               // V tile row stride (odd multiple of 32 B), rows incl. zero padding
constexpr int PROW = 656;                         // epilogue patch row stride (bytes)

__device__ __forceinline__ int xt_addr(int row, int col) {      // byte address of element (row, col) in XT (col multiple of 4 for 8-byte access)
    return (col >> 6) * (TM * 128) + row * 128 + ((((col & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + ((col & 7) << 1);
}

// max / sum over the lanes {l, l ^ 16, l ^ 32, l ^ 48} with the gfx950 permlane swaps (pure VALU, no LDS round trip)
__device__ __forceinline__ float group_max(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float group_sum(float v) {
    unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// out-tile GEMM: acc[i][j] (n-tile i of the wave's 80 columns, m-tile j of its 64 rows) = XT(rows) . W^T, K = 320: five k64 steps, the
// [320 rows][64 k] weight slab of a step staged by LDS-DMA into one of two 40 KB stages (chunk index XOR (row >> 1) & 7), one barrier per
// step.  (Weights straight from global into registers - no stage, no barrier - were measured too: 49 us per GEMM against 20 us staged;
// a lane group's 16-byte pieces of 16 different rows make poor vector-memory requests.)
__device__ __forceinline__ const f16* w_piece_src(const f16* __restrict__ w, int w8, int lane, int j) {
    const int r = 8 * (w8 + 8 * j) + (lane >> 3);
    return w + (size_t)r * C + ((lane & 7) ^ ((r >> 1) & 7)) * 8;
}
__device__ __forceinline__ void stage_w(const f16* __restrict__ w, char* WS, int w8, int lane, int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 5; ++j) glds16(w_piece_src(w, w8, lane, j) + kt * 64, WS + buf * WST + (w8 + 8 * j) * 1024);
}
template <bool PRESTAGED>
__device__ __forceinline__ void gemm_320(const f16* __restrict__ w, char* smem, int w8, int lane, int wm, int wn, f32x4 (&acc)[5][4]) {
    char* const XT = smem;
    char* const WS = smem + XT_BYTES;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swz = (lane >> 1) & 7;
    if (!PRESTAGED) stage_w(w, WS, w8, lane, 0, 0);   // (PRESTAGED: the caller issued stage 0 before its own work)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < 5; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < 5) stage_w(w, WS, w8, lane, kt + 1, buf ^ 1);
        const char* ta = XT + kt * (TM * 128) + (wm * 64) * 128;
        const char* tb = WS + buf * WST + (wn * 80) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fo = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ swz) << 4);
            f16x8 fa[4], fw[5];
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const f16x8*>(ta + j * 2048 + fo);
#pragma unroll
            for (int i = 0; i < 5; ++i) fw[i] = *reinterpret_cast<const f16x8*>(tb + i * 2048 + fo);
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// SPLIT: the residual stream is split-fp16 (value = h + h_lo).  The LayerNorm reads the hi plane (its output is an fp16 MFMA operand either way: a
// branch-level rounding); the residual add of the epilogue takes hi + lo and writes both planes, which is where the stream's precision lives.
template <int SPLIT>
__global__ __launch_bounds__(512, 2) void xattn_block_kernel(XattnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const VT = smem + XT_BYTES;               // V tile [96][672 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int g = lane >> 4, i16 = lane & 15;
    const int m_blk = blockIdx.x * TM;
    const int b = m_blk / p.HW;                      // the tile lies inside one sample (HW % 128 == 0)

    // ---------------- phase 0: LayerNorm of the wave's 16 rows -> XT; the V rows of the sample are fetched alongside ------------------
    stage_w(p.wq, VT, w, lane, 0, 0);               // Wq's first k-step lands while the LayerNorm runs (VT = the weight stages for now)
    // V of sample b: 96 rows x 42 chunks (40 data + 2 pad), 8 per thread: loads now, LDS writes after the to_q GEMM has released the stages
    u32x4 vt[8];
    {
        const f16* vsrc = p.kv + (size_t)b * p.Nk * (2 * C) + C;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int id = tid + 512 * k, row = id / 42, ch = id - row * 42;
            vt[k] = u32x4{0, 0, 0, 0};
            if (id < VROWS * 42 && row < p.Nk && ch < 40) vt[k] = *reinterpret_cast<const u32x4*>(vsrc + (size_t)row * (2 * C) + ch * 8);
        }
    }
    {
        const bool act = lane < 40;                  // 40 chunks of 8 channels per row
        f16x8 raw[16];                                // all 16 rows of the wave in flight at once: one memory round trip
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int m = m_blk + 16 * w + u;
            raw[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (act && m < p.M && !(p.debug & 8)) raw[u] = *reinterpret_cast<const f16x8*>(p.h + (size_t)m * C + lane * 8);
        }
        float gam[8], bet[8];
        if (act) {
            const f16x8 gv = *reinterpret_cast<const f16x8*>(p.ln_g + lane * 8), bv = *reinterpret_cast<const f16x8*>(p.ln_b + lane * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { gam[k] = (float)gv[k]; bet[k] = (float)bv[k]; }
        }
        // the 16 rows' reductions run as 16 independent butterflies (a row-at-a-time loop serialised 16 x 2 x 6 dependent cross-lane steps)
        float mean[16], rstd[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)raw[u][k];
            mean[u] = s;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 16; ++u) mean[u] += __shfl_xor(mean[u], o, 64);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            mean[u] *= (1.0f / C);
            float q = 0.f;
            if (act) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = (float)raw[u][k] - mean[u]; q += d * d; }
            }
            rstd[u] = q;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 16; ++u) rstd[u] += __shfl_xor(rstd[u], o, 64);
        if (act) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float rs = rsqrtf(rstd[u] * (1.0f / C) + p.ln_eps);
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pk(((float)raw[u][2 * k] - mean[u]) * rs * gam[2 * k] + bet[2 * k],
                              ((float)raw[u][2 * k + 1] - mean[u]) * rs * gam[2 * k + 1] + bet[2 * k + 1]);
                *reinterpret_cast<u32x4*>(XT + xt_addr(16 * w + u, lane * 8)) = o;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 1: q = LN(h) Wq^T -> XT (fp16, pre-scaled) ------------------------------------------------
    f32x4 acc[5][4];
    if (!(p.debug & 4)) gemm_320<true>(p.wq, smem, w, lane, wm, wn, acc);       // ends with a barrier: XT and the stages are free
    else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); for (int i = 0; i < 5; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{1.f, 1.f, 1.f, 1.f}; }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = wn * 80 + i * 16 + 4 * g, m = wm * 64 + j * 16 + i16;
            float qv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) qv[r] = (float)(f16)acc[i][j][r] * p.c;      // q as the unfused path stores it, then scaled
            *reinterpret_cast<u32x2*>(XT + xt_addr(m, n)) = u32x2{pk(qv[0], qv[1]), pk(qv[2], qv[3])};
        }
    // ---------------- phase 2: the V rows fetched in phase 0 -> LDS ([96][672 B], rows >= Nk and the pad chunks zero) ---------------
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int id = tid + 512 * k, row = id / 42, ch = id - row * 42;
        if (id < VROWS * 42) *reinterpret_cast<u32x4*>(VT + row * VS + ch * 16) = vt[k];
    }
    __syncthreads();

    // ---------------- phase 3: attention, 16 queries per wave, heads in sequence ------------------------------------------------
    if (!(p.debug & 1)) {
        const f16* kbase = p.kv + (size_t)b * p.Nk * (2 * C);
        const int qrow = 16 * w + i16;
        auto load_k = [&](int hd, u32x4 (&kf)[5][2]) {
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                const int key = kt * 16 + i16;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4 t = {0, 0, 0, 0};
                    if (key < p.Nk && (ks == 0 || g == 0))
                        t = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * (2 * C) + hd * DH + ks * 32 + 8 * g);
                    kf[kt][ks] = t;
                }
            }
        };
        u32x4 kf[5][2];
        load_k(0, kf);
#pragma unroll
        for (int hd = 0; hd < NH; ++hd) {
            // Q fragments of this head (B operand: lane = query column, k = 8 g + j)
            f16x8 qf[2];
            {
                const int c0 = hd * DH + 8 * g;                               // ks = 0: channels 0..31 of the head
                qf[0] = *reinterpret_cast<const f16x8*>(XT + xt_addr(qrow, c0));
                u32x4 z = {0, 0, 0, 0};
                if (g == 0) z = *reinterpret_cast<const u32x4*>(XT + xt_addr(qrow, hd * DH + 32));
                qf[1] = frag_of(z);
            }
            f32x4 s[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(kf[kt][0]), qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(kf[kt][1]), qf[1], s[kt], 0, 0, 0);
            }
            if (hd + 1 < NH) load_k(hd + 1, kf);                              // next head's K while this head's softmax / P V run
            // masked softmax over the keys of query i16 (lane holds keys kt*16 + 4g + r)
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt * 16 + 4 * g + r >= p.Nk) s[kt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = group_max(mx);
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; l += e; }
            l = group_sum(l);
            // P fragments (B operand of the second product; MFMA k index 8g + j <-> key t2*32 + (j < 4 ? 4g + j : 16 + 4g + j - 4))
            f16x8 pf[3];
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2) {
                u32x4 f;
                f[0] = pk(s[2 * t2][0], s[2 * t2][1]); f[1] = pk(s[2 * t2][2], s[2 * t2][3]);
                if (2 * t2 + 1 < 5) { f[2] = pk(s[2 * t2 + 1][0], s[2 * t2 + 1][1]); f[3] = pk(s[2 * t2 + 1][2], s[2 * t2 + 1][3]); }
                else { f[2] = 0; f[3] = 0; }
                pf[t2] = frag_of(f);
            }
            f32x4 o[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) o[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const char* addr = VT + (32 * t2 + 4 * g + (i16 >> 2)) * VS + (hd * DH + a * 16 + 4 * (i16 & 3)) * 2;
                    const u32x2 lo = tr_read(addr);
                    const u32x2 hi = tr_read(addr + 16 * VS);
                    o[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(u32x4{lo[0], lo[1], hi[0], hi[1]}), pf[t2], o[a], 0, 0, 0);
                }
            // O^T: lane holds channels a*16 + 4g + r of query i16 -> fp16 over the head's (consumed) q columns
            const float inv = 1.0f / l;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int d = a * 16 + 4 * g;
                if (d < DH)
                    *reinterpret_cast<u32x2*>(XT + xt_addr(qrow, hd * DH + d)) = u32x2{pk(o[a][0] * inv, o[a][1] * inv), pk(o[a][2] * inv, o[a][3] * inv)};
            }
        }
    }
    __syncthreads();

    // ---------------- phase 4: out = O Wo^T + bias + h ------------------------------------------------------------------------------------
    if (!(p.debug & 2)) gemm_320<false>(p.wo, smem, w, lane, wm, wn, acc);     // ends with a barrier: XT becomes the epilogue patch
    {
        char* const patch = smem;                    // [128 rows][PROW] = 83,968 B: all of XT and the first 2 KB of weight stage 0 (free: gemm_320 ends with a barrier)
        static_assert(TM * PROW <= XT_BYTES + WST, "the epilogue patch may reach into weight stage 0 but not beyond it");
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = wn * 80 + i * 16 + 4 * g;
            const f16x4 bv = *reinterpret_cast<const f16x4*>(p.bo + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = wm * 64 + j * 16 + i16;
                *reinterpret_cast<u32x2*>(patch + m * PROW + n * 2) =
                    u32x2{pk(acc[i][j][0] + (float)bv[0], acc[i][j][1] + (float)bv[1]), pk(acc[i][j][2] + (float)bv[2], acc[i][j][3] + (float)bv[3])};
            }
        }
        __syncthreads();
        // 128 rows x 40 chunks = 5120 items, 10 per thread: all residual loads of the thread first, then the patch reads, then the stores
        f16x8 res[10], resl[SPLIT == 1 ? 10 : 1];
        u32x2 resl8[SPLIT == 2 ? 10 : 1];                 // lo8: the lo planes are one e5m2 byte per element (IgemmArgs::lo8)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int id = tid + 512 * k, row = id / 40, ch = id - row * 40, m = m_blk + row;
            res[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M && !(p.debug & 16)) res[k] = *reinterpret_cast<const f16x8*>(p.h + (size_t)m * C + ch * 8);
            if constexpr (SPLIT == 1) {
                resl[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < p.M) resl[k] = *reinterpret_cast<const f16x8*>(p.h_lo + (size_t)m * C + ch * 8);
            }
            if constexpr (SPLIT == 2) {
                resl8[k] = u32x2{0, 0};
                if (m < p.M) resl8[k] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.h_lo) + (size_t)m * C + ch * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int id = tid + 512 * k, row = id / 40, ch = id - row * 40, m = m_blk + row;
            const f16x8 v = *reinterpret_cast<const f16x8*>(patch + row * PROW + ch * 16);
            f16x8 o;
            float s1 = 0.f, s2 = 0.f;
            if constexpr (SPLIT == 1) {
                f16x8 l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] + (float)res[k][e] + (float)resl[k][e];
                    o[e] = (f16)f; l[e] = (f16)(f - (float)o[e]);
                    s1 += f; s2 = __builtin_fmaf(f, f, s2);
                }
                if (m < p.M) *reinterpret_cast<f16x8*>(p.out_lo + (size_t)m * C + ch * 8) = l;
            } else if constexpr (SPLIT == 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 l01 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][0], false), l23 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][0], true);
                const f2 l45 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][1], false), l67 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][1], true);
                const float lv[8] = {l01[0], l01[1], l23[0], l23[1], l45[0], l45[1], l67[0], l67[1]};
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] + (float)res[k][e] + lv[e];
                    o[e] = (f16)f; d[e] = f - (float)o[e];
                    s1 += f; s2 = __builtin_fmaf(f, f, s2);
                }
                int w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[0], d[1], 0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[2], d[3], w0, true);
                int w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[4], d[5], 0, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[6], d[7], w1, true);
                if (m < p.M) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.out_lo) + (size_t)m * C + ch * 8) = u32x2{(unsigned)w0, (unsigned)w1};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e] + (float)res[k][e]; o[e] = (f16)f; s1 += f; s2 = __builtin_fmaf(f, f, s2); }
            }
            if (m < p.M && !(p.debug & 16)) *reinterpret_cast<f16x8*>(p.out + (size_t)m * C + ch * 8) = o;
            // row statistics for norm3 folded into the GEGLU GEMM: the item's partial sums go to weight stage 1 (free since the to_out GEMM's last barrier; the
            // patch reaches 2 KB into stage 0 only), [128 rows][40 chunks] float2 = 40,960 B = one stage exactly
            if (p.row_stats) *reinterpret_cast<float2*>(smem + XT_BYTES + WST + (row * 40 + ch) * 8) = float2{s1, s2};
        }
        if (p.row_stats) {                               // (uniform branch)
            __syncthreads();
            if (tid < TM && m_blk + tid < p.M) {
                float a1 = 0.f, a2 = 0.f;
                const char* src = smem + XT_BYTES + WST + tid * 320;
#pragma unroll
                for (int c2 = 0; c2 < 20; ++c2) { const f32x4 t = *reinterpret_cast<const f32x4*>(src + c2 * 16); a1 += t[0] + t[2]; a2 += t[1] + t[3]; }
                *reinterpret_cast<float2*>(p.row_stats + (size_t)(m_blk + tid) * 2) = float2{a1, a2};
            }
        }
    }
}


// =====================================================================================================================================================
// Round 5: the same sub-block on 64-row tiles, TWO workgroups per CU.
//
// xattn_block_kernel holds 160 KB of LDS, so a CU runs its four 128-row tiles strictly one after another and inside a tile every phase is exposed: all eight
// waves load rows, then all normalise, then all multiply, then all sit in the attention phase's dependent chains, then all store (profiles/r04_xattn_ablation.txt:
// 232 us per launch against a memory floor of 76 us and ~70 us of matrix work).  Here a workgroup is FOUR waves on 64 rows with the per-wave work unchanged
// (LayerNorm of 16 rows, a 64 x 80 GEMM wave tile, 16 queries x 8 heads, 10 epilogue items per thread), and its LDS is 80 KB:
//     XT 40 KB + two k32 weight stages of 20 KB ([320 weight rows][64 B], 16-byte chunk index XOR (row >> 2) & 3: conflict-free ds_read_b128 fragments);
//     the V tile goes through the stage area in two halves of four heads ([96 keys][352 B] = 33 KB each);
//     the epilogue patch is XT + the first KB of stage 0, the row-statistics scratch stage 1.
// Two such workgroups share a CU and drift apart, so one's loads / stores run under the other's MFMA and softmax phases.
constexpr int TM2 = 64;
constexpr int XT2_BYTES = 5 * TM2 * 128;          // 40960
constexpr int WST2 = 320 * 64;                    // 20480 per k32 weight stage
constexpr int VS2 = 352, VCH2 = 22;               // V half-tile row stride (odd multiple of 32 B) = 20 data chunks (4 heads x 40 channels) + 2 zero chunks

__device__ __forceinline__ int xt2_addr(int row, int col) {
    return (col >> 6) * (TM2 * 128) + row * 128 + ((((col & 63) >> 3) ^ ((row >> 1) & 7)) << 4) + ((col & 7) << 1);
}
// one k32 slab [320 rows][32 k] of w -> stage `buf`: 20 pieces of 16 rows, 5 per wave; lane -> (row = lane >> 2, LDS chunk = lane & 3), source chunk = LDS chunk ^ (row >> 2) & 3
__device__ __forceinline__ void stage_w32(const f16* __restrict__ w, char* WS, int w4, int lane, int kt, int buf) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int piece = w4 + 4 * j, r = 16 * piece + (lane >> 2);
        const f16* src = w + (size_t)r * C + kt * 32 + (((lane & 3) ^ ((lane >> 4) & 3)) << 3);
        glds16(src, WS + buf * WST2 + piece * 1024);
    }
}
template <bool PRESTAGED>
__device__ __forceinline__ void gemm_320_k32(const f16* __restrict__ w, char* smem, int w4, int lane, f32x4 (&acc)[5][4]) {
    char* const XT = smem;
    char* const WS = smem + XT2_BYTES;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int swz = (lane >> 1) & 7;
    if (!PRESTAGED) stage_w32(w, WS, w4, lane, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int wfo = (lane & 15) * 64 + ((((lane >> 4)) ^ ((lane >> 2) & 3)) << 4);       // weight fragment: row i16 of a 16-row piece, chunk g ^ (row >> 2) & 3
    for (int kt = 0; kt < 10; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < 10) stage_w32(w, WS, w4, lane, kt + 1, buf ^ 1);
        const char* ta = XT + (kt >> 1) * (TM2 * 128);
        const char* tb = WS + buf * WST2 + (w4 * 80) * 64;
        const int fo = (lane & 15) * 128 + ((((kt & 1) * 4 + (lane >> 4)) ^ swz) << 4);
        f16x8 fa[4], fw[5];
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[j] = *reinterpret_cast<const f16x8*>(ta + j * 2048 + fo);
#pragma unroll
        for (int i = 0; i < 5; ++i) fw[i] = *reinterpret_cast<const f16x8*>(tb + i * 1024 + wfo);
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fw[i], fa[j], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <int SPLIT>
__global__ __launch_bounds__(256, 2) void xattn64_kernel(XattnParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const XT = smem;
    char* const VT = smem + XT2_BYTES;              // V half tile [96][352 B] (the weight stages between the GEMMs)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3: LayerNorm rows 16 w.., GEMM columns 80 w.., attention queries 16 w..
    const int g = lane >> 4, i16 = lane & 15;
    const int m_blk = blockIdx.x * TM2;
    const int b = m_blk / p.HW;                      // the tile lies inside one sample (HW % 64 == 0)
    const f16* const vsrc = p.kv + (size_t)b * p.Nk * (2 * C) + C;

    // V half `hh` (heads 4 hh .. 4 hh + 3): 96 rows x 22 chunks, 9 per thread -> registers
    auto load_v = [&](int hh, u32x4 (&vt)[9]) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int id = tid + 256 * k, row = id / VCH2, ch = id - row * VCH2;
            vt[k] = u32x4{0, 0, 0, 0};
            if (id < VROWS * VCH2 && row < p.Nk && ch < 20) vt[k] = *reinterpret_cast<const u32x4*>(vsrc + (size_t)row * (2 * C) + hh * (4 * DH) + ch * 8);
        }
    };
    auto store_v = [&](const u32x4 (&vt)[9]) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int id = tid + 256 * k, row = id / VCH2, ch = id - row * VCH2;
            if (id < VROWS * VCH2) *reinterpret_cast<u32x4*>(VT + row * VS2 + ch * 16) = vt[k];
        }
    };

    // ---------------- phase 0: LayerNorm of the wave's 16 rows -> XT; V half 0 fetched alongside ------------------
    stage_w32(p.wq, VT, w, lane, 0, 0);
    {
        const bool act = lane < 40;
        f16x8 raw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int m = m_blk + 16 * w + u;
            raw[u] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (act && m < p.M) raw[u] = *reinterpret_cast<const f16x8*>(p.h + (size_t)m * C + lane * 8);
        }
        float gam[8], bet[8];
        if (act) {
            const f16x8 gv = *reinterpret_cast<const f16x8*>(p.ln_g + lane * 8), bv = *reinterpret_cast<const f16x8*>(p.ln_b + lane * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { gam[k] = (float)gv[k]; bet[k] = (float)bv[k]; }
        }
        float mean[16], rstd[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (float)raw[u][k];
            mean[u] = s;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 16; ++u) mean[u] += __shfl_xor(mean[u], o, 64);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            mean[u] *= (1.0f / C);
            float q = 0.f;
            if (act) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float d = (float)raw[u][k] - mean[u]; q += d * d; }
            }
            rstd[u] = q;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < 16; ++u) rstd[u] += __shfl_xor(rstd[u], o, 64);
        if (act) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float rs = rsqrtf(rstd[u] * (1.0f / C) + p.ln_eps);
                u32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    o[k] = pk(((float)raw[u][2 * k] - mean[u]) * rs * gam[2 * k] + bet[2 * k],
                              ((float)raw[u][2 * k + 1] - mean[u]) * rs * gam[2 * k + 1] + bet[2 * k + 1]);
                *reinterpret_cast<u32x4*>(XT + xt2_addr(16 * w + u, lane * 8)) = o;
            }
        }
    }
    __syncthreads();

    // ---------------- phase 1: q = LN(h) Wq^T -> XT (fp16, pre-scaled); V half 0 travels under it ------------------------------------------------
    u32x4 vt[9];
    load_v(0, vt);               // (issued here, not next to the LayerNorm's 16 row loads: with both in flight phase 0 spilled 34 registers)
    f32x4 acc[5][4];
    gemm_320_k32<true>(p.wq, smem, w, lane, acc);       // ends with a barrier: XT and the stages are free
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = w * 80 + i * 16 + 4 * g, m = j * 16 + i16;
            float qv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) qv[r] = (float)(f16)acc[i][j][r] * p.c;      // q as the unfused path stores it, then scaled
            *reinterpret_cast<u32x2*>(XT + xt2_addr(m, n)) = u32x2{pk(qv[0], qv[1]), pk(qv[2], qv[3])};
        }

    // ---------------- phases 2 / 3: per head half: V half -> LDS, attention of 16 queries per wave over its four heads -------------
    const f16* kbase = p.kv + (size_t)b * p.Nk * (2 * C);
    const int qrow = 16 * w + i16;
    auto load_k = [&](int hd, u32x4 (&kf)[5][2]) {
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            const int key = kt * 16 + i16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 t = {0, 0, 0, 0};
                if (key < p.Nk && (ks == 0 || g == 0))
                    t = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * (2 * C) + hd * DH + ks * 32 + 8 * g);
                kf[kt][ks] = t;
            }
        }
    };
    u32x4 kf[5][2];
    load_k(0, kf);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        if (hh == 1) __syncthreads();                 // every wave is done with V half 0
        store_v(vt);
        if (hh == 0) load_v(1, vt);                   // the second half's rows travel while the first half's heads compute
        __syncthreads();
#pragma unroll
        for (int hl = 0; hl < 4; ++hl) {
            const int hd = 4 * hh + hl;
            f16x8 qf[2];
            {
                const int c0 = hd * DH + 8 * g;
                qf[0] = *reinterpret_cast<const f16x8*>(XT + xt2_addr(qrow, c0));
                u32x4 z = {0, 0, 0, 0};
                if (g == 0) z = *reinterpret_cast<const u32x4*>(XT + xt2_addr(qrow, hd * DH + 32));
                qf[1] = frag_of(z);
            }
            f32x4 s[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(kf[kt][0]), qf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(kf[kt][1]), qf[1], s[kt], 0, 0, 0);
            }
            if (hd + 1 < NH) load_k(hd + 1, kf);
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (kt * 16 + 4 * g + r >= p.Nk) s[kt][r] = -INFINITY;
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = group_max(mx);
            float l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 5; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx); s[kt][r] = e; l += e; }
            l = group_sum(l);
            f16x8 pf[3];
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2) {
                u32x4 f;
                f[0] = pk(s[2 * t2][0], s[2 * t2][1]); f[1] = pk(s[2 * t2][2], s[2 * t2][3]);
                if (2 * t2 + 1 < 5) { f[2] = pk(s[2 * t2 + 1][0], s[2 * t2 + 1][1]); f[3] = pk(s[2 * t2 + 1][2], s[2 * t2 + 1][3]); }
                else { f[2] = 0; f[3] = 0; }
                pf[t2] = frag_of(f);
            }
            f32x4 o[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) o[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t2 = 0; t2 < 3; ++t2)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const char* addr = VT + (32 * t2 + 4 * g + (i16 >> 2)) * VS2 + (hl * DH + a * 16 + 4 * (i16 & 3)) * 2;
                    const u32x2 lo = tr_read(addr);
                    const u32x2 hi = tr_read(addr + 16 * VS2);
                    o[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag_of(u32x4{lo[0], lo[1], hi[0], hi[1]}), pf[t2], o[a], 0, 0, 0);
                }
            const float inv = 1.0f / l;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int d = a * 16 + 4 * g;
                if (d < DH)
                    *reinterpret_cast<u32x2*>(XT + xt2_addr(qrow, hd * DH + d)) = u32x2{pk(o[a][0] * inv, o[a][1] * inv), pk(o[a][2] * inv, o[a][3] * inv)};
            }
        }
    }
    __syncthreads();

    // ---------------- phase 4: out = O Wo^T + bias + h ------------------------------------------------------------------------------------
    gemm_320_k32<false>(p.wo, smem, w, lane, acc);     // ends with a barrier: XT becomes the epilogue patch
    {
        char* const patch = smem;                    // [64 rows][PROW] = 41,984 B: all of XT and the first KB of weight stage 0
        static_assert(TM2 * PROW <= XT2_BYTES + WST2, "the epilogue patch may reach into weight stage 0 but not beyond it");
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int n = w * 80 + i * 16 + 4 * g;
            const f16x4 bv = *reinterpret_cast<const f16x4*>(p.bo + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = j * 16 + i16;
                *reinterpret_cast<u32x2*>(patch + m * PROW + n * 2) =
                    u32x2{pk(acc[i][j][0] + (float)bv[0], acc[i][j][1] + (float)bv[1]), pk(acc[i][j][2] + (float)bv[2], acc[i][j][3] + (float)bv[3])};
            }
        }
        __syncthreads();
        // 64 rows x 40 chunks = 2560 items, 10 per thread: all residual loads of the thread first, then the patch reads, then the stores
        f16x8 res[10], resl[SPLIT == 1 ? 10 : 1];
        u32x2 resl8[SPLIT == 2 ? 10 : 1];                 // lo8: the lo planes are one e5m2 byte per element (IgemmArgs::lo8)
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int id = tid + 256 * k, row = id / 40, ch = id - row * 40, m = m_blk + row;
            res[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            if (m < p.M) res[k] = *reinterpret_cast<const f16x8*>(p.h + (size_t)m * C + ch * 8);
            if constexpr (SPLIT == 1) {
                resl[k] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
                if (m < p.M) resl[k] = *reinterpret_cast<const f16x8*>(p.h_lo + (size_t)m * C + ch * 8);
            }
            if constexpr (SPLIT == 2) {
                resl8[k] = u32x2{0, 0};
                if (m < p.M) resl8[k] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned char*>(p.h_lo) + (size_t)m * C + ch * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            const int id = tid + 256 * k, row = id / 40, ch = id - row * 40, m = m_blk + row;
            const f16x8 v = *reinterpret_cast<const f16x8*>(patch + row * PROW + ch * 16);
            f16x8 o;
            float s1 = 0.f, s2 = 0.f;
            if constexpr (SPLIT == 1) {
                f16x8 l;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] + (float)res[k][e] + (float)resl[k][e];
                    o[e] = (f16)f; l[e] = (f16)(f - (float)o[e]);
                    s1 += f; s2 = __builtin_fmaf(f, f, s2);
                }
                if (m < p.M) *reinterpret_cast<f16x8*>(p.out_lo + (size_t)m * C + ch * 8) = l;
            } else if constexpr (SPLIT == 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 l01 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][0], false), l23 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][0], true);
                const f2 l45 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][1], false), l67 = __builtin_amdgcn_cvt_pk_f32_bf8((int)resl8[k][1], true);
                const float lv[8] = {l01[0], l01[1], l23[0], l23[1], l45[0], l45[1], l67[0], l67[1]};
                float d[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] + (float)res[k][e] + lv[e];
                    o[e] = (f16)f; d[e] = f - (float)o[e];
                    s1 += f; s2 = __builtin_fmaf(f, f, s2);
                }
                int w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[0], d[1], 0, false); w0 = __builtin_amdgcn_cvt_pk_bf8_f32(d[2], d[3], w0, true);
                int w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[4], d[5], 0, false); w1 = __builtin_amdgcn_cvt_pk_bf8_f32(d[6], d[7], w1, true);
                if (m < p.M) *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(p.out_lo) + (size_t)m * C + ch * 8) = u32x2{(unsigned)w0, (unsigned)w1};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e] + (float)res[k][e]; o[e] = (f16)f; s1 += f; s2 = __builtin_fmaf(f, f, s2); }
            }
            if (m < p.M) *reinterpret_cast<f16x8*>(p.out + (size_t)m * C + ch * 8) = o;
            // row statistics for norm3 folded into the GEGLU GEMM: the item's partial sums go to weight stage 1 ([64 rows][40 chunks] float2 = one k32 stage exactly)
            if (p.row_stats) *reinterpret_cast<float2*>(smem + XT2_BYTES + WST2 + (row * 40 + ch) * 8) = float2{s1, s2};
        }
        if (p.row_stats) {                               // (uniform branch)
            __syncthreads();
            if (tid < TM2 && m_blk + tid < p.M) {
                float a1 = 0.f, a2 = 0.f;
                const char* src = smem + XT2_BYTES + WST2 + tid * 320;
#pragma unroll
                for (int c2 = 0; c2 < 20; ++c2) { const f32x4 t = *reinterpret_cast<const f32x4*>(src + c2 * 16); a1 += t[0] + t[2]; a2 += t[1] + t[3]; }
                *reinterpret_cast<float2*>(p.row_stats + (size_t)(m_blk + tid) * 2) = float2{a1, a2};
            }
        }
    }
}

}  // namespace

int launch_xattn_block(const XattnArgs& a, hipStream_t s) {
    if (!a.h || !a.out || !a.ln_g || !a.ln_b || !a.wq || !a.wo || !a.bo || !a.kv) CS_FAIL(CS_E_ARG, "xattn_block: null pointer");
    if (a.C != 320 || a.heads != 8) CS_FAIL(CS_E_UNSUPPORTED, "xattn_block: built for C = 320, 8 heads (got C = %d, heads = %d)", a.C, a.heads);
    if (a.Nk < 1 || a.Nk > 80) CS_FAIL(CS_E_SHAPE, "xattn_block: 1 <= Nk <= 80 (got %d)", a.Nk);
    if (a.HW % TM && !(tune().xattn_tile == 64 && a.HW % TM2 == 0)) CS_FAIL(CS_E_SHAPE, "xattn_block: rows per sample must be a multiple of %d", TM);
    if (a.M <= 0) return a.M < 0 ? CS_E_SHAPE : CS_OK;
    if (a.M % a.HW) CS_FAIL(CS_E_SHAPE, "xattn_block: M must be a whole number of samples");
    XattnParams p;
    if ((a.h_lo == nullptr) != (a.out_lo == nullptr)) CS_FAIL(CS_E_ARG, "xattn_block: h_lo and out_lo go together");
    p.h = a.h; p.out = a.out; p.h_lo = a.h_lo; p.out_lo = a.out_lo; p.row_stats = a.row_stats; p.ln_g = a.ln_g; p.ln_b = a.ln_b; p.ln_eps = a.ln_eps; p.wq = a.wq; p.wo = a.wo; p.bo = a.bo; p.kv = a.kv;
    p.M = a.M; p.HW = a.HW; p.Nk = a.Nk; p.c = a.scale * 1.4426950408889634f; p.debug = tune().debug;
    if (tune().xattn_tile == 64 && a.HW % TM2 == 0) {     // round 5 default: 64-row tiles, two workgroups per CU
        constexpr size_t lds2 = XT2_BYTES + 2 * WST2;      // 81920
        static bool configured2 = false;
        if (!configured2) {
            CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn64_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
            configured2 = true;
        }
        if (a.h_lo && a.lo8) hipLaunchKernelGGL(xattn64_kernel<2>, dim3(a.M / TM2), dim3(256), lds2, s, p);
        else if (a.h_lo) hipLaunchKernelGGL(xattn64_kernel<1>, dim3(a.M / TM2), dim3(256), lds2, s, p);
        else hipLaunchKernelGGL(xattn64_kernel<0>, dim3(a.M / TM2), dim3(256), lds2, s, p);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    constexpr size_t lds = XT_BYTES + 2 * WST;      // 163840: XT + two weight stages (the V tile reuses the stages, the epilogue patch XT)
    static bool configured = false;
    if (!configured) {
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_block_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_block_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(xattn_block_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    if (a.h_lo && a.lo8) hipLaunchKernelGGL(xattn_block_kernel<2>, dim3(a.M / TM), dim3(512), lds, s, p);
    else if (a.h_lo) hipLaunchKernelGGL(xattn_block_kernel<1>, dim3(a.M / TM), dim3(512), lds, s, p);
    else hipLaunchKernelGGL(xattn_block_kernel<0>, dim3(a.M / TM), dim3(512), lds, s, p);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
