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

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

__device__ __attribute__((aligned(256))) unsigned g_zero_page[64];

struct IgemmParams {
    const f16* a0; const f16* a1; int c0, c1;
    int Hi, Wi, Ho, Wo, HoWo;
    int stride, upsample;
    int M, N, KT, cpt;   // KT = K / 64 ; cpt = chunks (of 64 channels) per tap
    int Ktot;            // row length of w
    const f16* w; const f16* bias; const f16* temb; int temb_stride; const f16* res; f16* out;
    int tiles_n, nblk;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

template <int BN, bool CONV3, bool GEGLU>
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
    int id;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = p.nblk >> 3, r = p.nblk & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = id / p.tiles_n, tn = id - tm * p.tiles_n;
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
                a_y[j] = yo * p.stride - 1;
                a_x[j] = xo * p.stride - 1;
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
        const int tap = kt / p.cpt;
        const int cc = (kt - tap * p.cpt) * BK;
        const f16* src; int cs, coff;
        if (cc < p.c0) { src = p.a0; cs = p.c0; coff = cc; } else { src = p.a1; cs = p.c1; coff = cc - p.c0; }
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
        for (int j = 0; j < NBI; ++j) glds16(b_src[j] + (size_t)kt * BK, lb + j * 1024);
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

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < p.KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < p.KT) stage(kt + 1, buf ^ 1);
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

    // ---- epilogue: lane holds, per (nt, mt) tile, pixel m = ..+(lane&15), channels 4*(lane>>4)..+3 ----
    const int g4 = (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = m_blk + wm * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        const f16* trow = nullptr;
        if (p.temb) trow = p.temb + (size_t)(m / p.HoWo) * p.temb_stride;
        if (GEGLU) {
            const int No = p.N >> 1;
#pragma unroll
            for (int i = 0; i < NT; i += 2) {
                const int nrow = n_blk + wn * (BN / 2) + i * 16 + g4;        // row of the permuted weight (value half)
                const int nout = ((n_blk + wn * (BN / 2) + i * 16) >> 1) + g4;
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[i][j][r], g = acc[i + 1][j][r];
                    if (p.bias) { v += (float)p.bias[nrow + r]; g += (float)p.bias[nrow + 16 + r]; }
                    o[r] = (f16)(v * gelu_erf(g));
                }
                *reinterpret_cast<f16x4*>(p.out + (size_t)m * No + nout) = o;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int n = n_blk + wn * (BN / 2) + i * 16 + g4;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (p.bias) {
                    const f16x4 bv = *reinterpret_cast<const f16x4*>(p.bias + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)bv[r];
                }
                if (trow) {
                    const f16x4 tv = *reinterpret_cast<const f16x4*>(trow + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)tv[r];
                }
                if (p.res) {
                    const f16x4 rv = *reinterpret_cast<const f16x4*>(p.res + (size_t)m * p.N + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
                }
                f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
                *reinterpret_cast<f16x4*>(p.out + (size_t)m * p.N + n) = o;
            }
        }
    }
}

template <int BN, bool CONV3, bool GEGLU>
int launch_variant(const IgemmParams& p, hipStream_t s) {
    constexpr size_t lds = 2 * (BM * BK * 2 + BN * BK * 2);
    static bool configured = false;
    auto kfn = igemm_kernel<BN, CONV3, GEGLU>;
    if (!configured) {
        CS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured = true;
    }
    hipLaunchKernelGGL(kfn, dim3(p.nblk), dim3(256), lds, s, p);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

}  // namespace

double igemm_flops(const IgemmArgs& a) {
    const double M = (double)a.B * a.Ho * a.Wo;
    return 2.0 * M * a.N * (double)a.taps * (a.c0 + a.c1);
}

int launch_igemm(const IgemmArgs& a, hipStream_t s) {
    const int cin = a.c0 + a.c1;
    if (!a.a0 || !a.w || !a.out) CS_FAIL(CS_E_ARG, "igemm: a0, w, out required");
    if (a.taps != 1 && a.taps != 9) CS_FAIL(CS_E_ARG, "igemm: taps must be 1 or 9");
    if (cin % BK || a.c0 % BK || (a.c1 && !a.a1)) CS_FAIL(CS_E_SHAPE, "igemm: channels (%d,%d) must be multiples of %d", a.c0, a.c1, BK);
    if (a.B <= 0 || a.Ho <= 0 || a.Wo <= 0) return (a.B < 0) ? CS_E_SHAPE : CS_OK;
    if (a.taps == 1 && (a.stride != 1 || a.upsample || a.Hi != a.Ho || a.Wi != a.Wo)) CS_FAIL(CS_E_ARG, "igemm: 1x1 needs stride 1, no upsample");
    if (a.geglu && (a.N % 256)) CS_FAIL(CS_E_SHAPE, "igemm: GEGLU needs N %% 256 == 0 (N=%d)", a.N);
    IgemmParams p;
    p.a0 = a.a0; p.a1 = a.a1; p.c0 = a.c0; p.c1 = a.c1;
    p.Hi = a.Hi; p.Wi = a.Wi; p.Ho = a.Ho; p.Wo = a.Wo; p.HoWo = a.Ho * a.Wo;
    p.stride = a.stride; p.upsample = a.upsample;
    p.M = a.B * a.Ho * a.Wo; p.N = a.N; p.cpt = cin / BK; p.KT = a.taps * p.cpt; p.Ktot = a.taps * cin;
    p.w = a.w; p.bias = a.bias; p.temb = a.temb; p.temb_stride = a.temb_stride; p.res = a.res; p.out = a.out;
    const int tiles_m = (p.M + BM - 1) / BM;
    int bn;
    if (a.N % 128 == 0) bn = 128;
    else if (a.N % 160 == 0) bn = 160;
    else CS_FAIL(CS_E_SHAPE, "igemm: N=%d must be a multiple of 128 or 160", a.N);
    p.tiles_n = a.N / bn; p.nblk = tiles_m * p.tiles_n;
    const bool conv3 = a.taps == 9;
    if (a.geglu) {
        if (conv3) CS_FAIL(CS_E_ARG, "igemm: GEGLU epilogue only for linear layers");
        return launch_variant<128, false, true>(p, s);
    }
    if (bn == 128) return conv3 ? launch_variant<128, true, false>(p, s) : launch_variant<128, false, false>(p, s);
    return conv3 ? launch_variant<160, true, false>(p, s) : launch_variant<160, false, false>(p, s);
}
