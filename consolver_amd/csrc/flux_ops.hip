// HBM-bound glue of the FLUX DiT: adaLN-modulated LayerNorm, per-head q/k RMSNorm + rotary embedding
// (in place on the fused qkv buffer), timestep sinusoids and small fp32 helpers.  16 bytes per lane,
// wave-shuffle reductions, f16 or bf16 storage with fp32 math.
#include "ops.h"
#include "el.h"

namespace {

// y = LN_noaffine(x) * (1 + scale[b]) + shift[b]; one wave per token row, row kept in registers
// SPLIT: x is a split residual stream (value = x + x_lo, two planes of T)
// y_lo (SPLIT only, optional): the output as two planes too, y_lo = T(o - float(T(o))) -- the output head's LayerNorm, whose result is the A operand of proj_out
template <typename T, int MAXV, bool SPLIT = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const u16* __restrict__ x, u16* __restrict__ y, int M, int C, int rows_per_sample,
                                                          const float* __restrict__ shift, const float* __restrict__ scale, long mod_stride, float eps,
                                                          const u16* __restrict__ x_lo = nullptr, u16* __restrict__ y_lo = nullptr) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= M) return;
    const int CV = C >> 3;
    float v[MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(x + (size_t)row * C + cv * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[j][2 * k] = El<T>::tof((u16)(t[k] & 0xffff)); v[j][2 * k + 1] = El<T>::tof((u16)(t[k] >> 16)); }
            if constexpr (SPLIT) {
                const u32x4 t2 = *reinterpret_cast<const u32x4*>(x_lo + (size_t)row * C + cv * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[j][2 * k] += El<T>::tof((u16)(t2[k] & 0xffff)); v[j][2 * k + 1] += El<T>::tof((u16)(t2[k] >> 16)); }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += v[j][k];
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j][k] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (lane + 64 * j < CV) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { const float d = v[j][k] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    const float* sh = shift + (size_t)(row / rows_per_sample) * mod_stride;
    const float* sc = scale + (size_t)(row / rows_per_sample) * mod_stride;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + cv * 8), s1 = *reinterpret_cast<const f32x4*>(sc + cv * 8 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + cv * 8), h1 = *reinterpret_cast<const f32x4*>(sh + cv * 8 + 4);
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = (v[j][k] - mean) * rstd * (1.0f + s0[k]) + h0[k];
                o[4 + k] = (v[j][4 + k] - mean) * rstd * (1.0f + s1[k]) + h1[k];
            }
            const u32x4 pk = {pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)row * C + cv * 8) = pk;
            if constexpr (SPLIT) {
                if (y_lo) {
                    float r[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) { r[2 * k] = o[2 * k] - El<T>::tof((u16)(pk[k] & 0xffff)); r[2 * k + 1] = o[2 * k + 1] - El<T>::tof((u16)(pk[k] >> 16)); }
                    *reinterpret_cast<u32x4*>(y_lo + (size_t)row * C + cv * 8) = u32x4{pack2<T>(r[0], r[1]), pack2<T>(r[2], r[3]), pack2<T>(r[4], r[5]), pack2<T>(r[6], r[7])};
                }
            }
        }
    }
}

// out[i] = float(hi[i]) + float(lo[i]): a tensor kept as two planes of the model dtype, handed out in fp32 (the output head's velocity)
template <typename T>
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const u16* __restrict__ hi, const u16* __restrict__ lo, float* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = El<T>::tof(hi[i]) + El<T>::tof(lo[i]);
}

// one wave per (token row, group of 64 / (2 dh/8) heads): every lane holds 8 dims of q or k of one head (dh = 128: lanes 0-15 q(h), 16-31 k(h),
// 32-47 q(h+1), 48-63 k(h+1)), two token rows per wave so that two independent 16-byte loads are in flight per lane; RMS over the head, * weight,
// then rotate the (2i, 2i+1) pairs with cos/sin[pos][i]
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(u16* __restrict__ qkv, long ld, int rows, int seq, int heads, int dh, int q_col, int k_col,
                                                           const u16* __restrict__ wq, const u16* __restrict__ wk, const u16* __restrict__ wq_ctx,
                                                           const u16* __restrict__ wk_ctx, int ctx_rows, const float* __restrict__ cosv,
                                                           const float* __restrict__ sinv, float eps) {
    const int lane = threadIdx.x & 63;
    const int nv = dh >> 3;                      // lanes per tensor per head (dh = 128 -> 16)
    const int hpw = 64 / (2 * nv);               // heads per wave
    const int hgroups = (heads + hpw - 1) / hpw;
    const long item = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int half_rows = (rows + 1) >> 1;
    if (item >= (long)half_rows * hgroups) return;
    const int row0 = (int)(item / hgroups), hg = (int)(item - (long)row0 * hgroups);
    const int sub = lane / (2 * nv), which = (lane / nv) & 1, li = lane % nv;     // head in the group, 0 = q / 1 = k, 8-dim slot
    const int h = hg * hpw + sub;
    const bool head_ok = h < heads;
    u32x4 t[2]; u16* ptr[2]; bool ok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = row0 + u * half_rows;
        ok[u] = head_ok && row < rows;
        ptr[u] = qkv + (size_t)(ok[u] ? row : 0) * ld + (which == 0 ? q_col : k_col) + (head_ok ? h : 0) * dh + li * 8;
        t[u] = *reinterpret_cast<const u32x4*>(ptr[u]);
    }
    const u16* wsel_img = which == 0 ? wq : wk;
    const u16* wsel_ctx = which == 0 ? (wq_ctx ? wq_ctx : wq) : (wk_ctx ? wk_ctx : wk);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = row0 + u * half_rows;
        const int pos = (ok[u] ? row : 0) % seq;
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[2 * k] = El<T>::tof((u16)(t[u][k] & 0xffff)); v[2 * k + 1] = El<T>::tof((u16)(t[u][k] >> 16)); }
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) sq += v[k] * v[k];
        for (int o = nv >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);      // within the nv-lane group (nv is a power of two <= 32)
        const float r = rsqrtf(sq / (float)dh + eps);
        const u32x4 wv = *reinterpret_cast<const u32x4*>((pos < ctx_rows ? wsel_ctx : wsel_img) + li * 8);
        float nrm[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // torch RMSNorm: (x * rsqrt(var + eps)) cast to the weight dtype, then * weight
            nrm[2 * k] = El<T>::tof(El<T>::fromf(v[2 * k] * r)) * El<T>::tof((u16)(wv[k] & 0xffff));
            nrm[2 * k + 1] = El<T>::tof(El<T>::fromf(v[2 * k + 1] * r)) * El<T>::tof((u16)(wv[k] >> 16));
        }
        const f32x4 c4 = *reinterpret_cast<const f32x4*>(cosv + (size_t)pos * (dh >> 1) + li * 4);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinv + (size_t)pos * (dh >> 1) + li * 4);
        float o[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = El<T>::tof(El<T>::fromf(nrm[2 * k])), b = El<T>::tof(El<T>::fromf(nrm[2 * k + 1]));
            o[2 * k] = a * c4[k] - b * s4[k];
            o[2 * k + 1] = b * c4[k] + a * s4[k];
        }
        if (ok[u]) *reinterpret_cast<u32x4*>(ptr[u]) = u32x4{pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7])};
    }
}

__global__ void sinusoid_f32_kernel(const float* __restrict__ t, float mult, int R, int C, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = C / 2;
    if (i >= R * half) return;
    const int r = i / half, k = i - r * half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[r] * mult * f;
    out[(size_t)r * C + k] = cosf(a);
    out[(size_t)r * C + half + k] = sinf(a);
}

__global__ void add3_kernel(const float* a, const float* b, const float* c, float* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + (b ? b[i] : 0.f) + (c ? c[i] : 0.f);
}


// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * w   (no mean subtraction, no bias); one wave per row, row in registers
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void rms_norm_kernel(const u16* __restrict__ x, const u16* __restrict__ wt, u16* __restrict__ y, int M, int C, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + w;
    if (row >= M) return;
    const int CV = C >> 3;
    float v[MAXV][8];
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            const u32x4 t = *reinterpret_cast<const u32x4*>(x + (size_t)row * C + cv * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[j][2 * k] = El<T>::tof((u16)(t[k] & 0xffff)); v[j][2 * k + 1] = El<T>::tof((u16)(t[k] >> 16)); }
#pragma unroll
            for (int k = 0; k < 8; ++k) sq += v[j][k] * v[j][k];
        }
    }
    const float r = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            const u32x4 g = *reinterpret_cast<const u32x4*>(wt + cv * 8);
            float o[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = v[j][2 * k] * r * El<T>::tof((u16)(g[k] & 0xffff));
                o[2 * k + 1] = v[j][2 * k + 1] * r * El<T>::tof((u16)(g[k] >> 16));
            }
            const u32x4 pk = {pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7])};
            *reinterpret_cast<u32x4*>(y + (size_t)row * C + cv * 8) = pk;
        }
    }
}

template <typename T> __global__ void gated_mul_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ out, long nv) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const u32x4 x = *reinterpret_cast<const u32x4*>(a + i * 8), y = *reinterpret_cast<const u32x4*>(b + i * 8);
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = pack2<T>(El<T>::tof((u16)(x[k] & 0xffff)) * El<T>::tof((u16)(y[k] & 0xffff)), El<T>::tof((u16)(x[k] >> 16)) * El<T>::tof((u16)(y[k] >> 16)));
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
}

__global__ void embed_rows_kernel(const int64_t* __restrict__ ids, const u16* __restrict__ table, u16* __restrict__ out, long rows, int C, int vocab) {
    const int CV = C >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * CV) return;
    const long r = i / CV; const int cv = (int)(i - r * CV);
    long id = ids[r]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *reinterpret_cast<u32x4*>(out + r * C + cv * 8) = *reinterpret_cast<const u32x4*>(table + id * C + cv * 8);
}

template <typename T> __global__ void cast_kernel(const float* x, u16* out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = El<T>::fromf(x[i]);
}

}  // namespace

int launch_planes_to_f32(const void* hi, const void* lo, float* out, long n, int dtype, hipStream_t s) {
    if (!hi || !lo || !out) CS_FAIL(CS_E_ARG, "planes_to_f32: null pointer");
    if (n <= 0) return CS_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == CS_BF16) hipLaunchKernelGGL(planes_to_f32_kernel<bf16_el>, grid, block, 0, s, (const u16*)hi, (const u16*)lo, out, n);
    else if (dtype == CS_F16) hipLaunchKernelGGL(planes_to_f32_kernel<f16>, grid, block, 0, s, (const u16*)hi, (const u16*)lo, out, n);
    else CS_FAIL(CS_E_DTYPE, "planes_to_f32: dtype");
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_ln_modulate(const void* x, void* y, int M, int C, int rows_per_sample, const float* shift, const float* scale, long mod_stride,
                       float eps, int dtype, hipStream_t s, const void* x_lo, void* y_lo) {
    if (!x || !y || !shift || !scale) CS_FAIL(CS_E_ARG, "ln_modulate: null pointer");
    if (y_lo && !x_lo) CS_FAIL(CS_E_ARG, "ln_modulate: y_lo is the split form's (x_lo required)");
    if (C % 8 || C > 8 * 64 * 8) CS_FAIL(CS_E_SHAPE, "ln_modulate: C=%d unsupported", C);
    if (M <= 0) return CS_OK;
    const dim3 grid((M + 3) / 4), block(256);
    const int nv = (C / 8 + 63) / 64;
#define LNM(T, V) do { if (x_lo) hipLaunchKernelGGL((ln_modulate_kernel<T, V, true>), grid, block, 0, s, (const u16*)x, (u16*)y, M, C, rows_per_sample, shift, scale, mod_stride, eps, (const u16*)x_lo, (u16*)y_lo); \
                       else hipLaunchKernelGGL((ln_modulate_kernel<T, V, false>), grid, block, 0, s, (const u16*)x, (u16*)y, M, C, rows_per_sample, shift, scale, mod_stride, eps, (const u16*)nullptr); } while (0)
    if (dtype == CS_BF16) { if (nv <= 1) LNM(bf16_el, 1); else if (nv <= 2) LNM(bf16_el, 2); else if (nv <= 4) LNM(bf16_el, 4); else if (nv <= 6) LNM(bf16_el, 6); else LNM(bf16_el, 8); }
    else if (dtype == CS_F16) { if (nv <= 1) LNM(f16, 1); else if (nv <= 2) LNM(f16, 2); else if (nv <= 4) LNM(f16, 4); else if (nv <= 6) LNM(f16, 6); else LNM(f16, 8); }
    else CS_FAIL(CS_E_DTYPE, "ln_modulate: dtype");
#undef LNM
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_qk_norm_rope(void* qkv, long ld, int rows, int seq, int heads, int dh, int q_col, int k_col, const void* wq, const void* wk,
                        const void* wq_ctx, const void* wk_ctx, int ctx_rows, const float* cosv, const float* sinv, float eps, int dtype,
                        hipStream_t s) {
    if (!qkv || !wq || !wk || !cosv || !sinv) CS_FAIL(CS_E_ARG, "qk_norm_rope: null pointer");
    if (dh % 8 || (dh / 8) > 32 || ((dh / 8) & (dh / 8 - 1))) CS_FAIL(CS_E_SHAPE, "qk_norm_rope: head dim %d unsupported", dh);
    if (rows <= 0) return CS_OK;
    const int nvl = dh >> 3, hpw = 64 / (2 * nvl);
    const long items = (long)((rows + 1) / 2) * ((heads + hpw - 1) / hpw);
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    if (dtype == CS_BF16) hipLaunchKernelGGL(qk_norm_rope_kernel<bf16_el>, grid, block, 0, s, (u16*)qkv, ld, rows, seq, heads, dh, q_col, k_col, (const u16*)wq, (const u16*)wk, (const u16*)wq_ctx, (const u16*)wk_ctx, ctx_rows, cosv, sinv, eps);
    else if (dtype == CS_F16) hipLaunchKernelGGL(qk_norm_rope_kernel<f16>, grid, block, 0, s, (u16*)qkv, ld, rows, seq, heads, dh, q_col, k_col, (const u16*)wq, (const u16*)wk, (const u16*)wq_ctx, (const u16*)wk_ctx, ctx_rows, cosv, sinv, eps);
    else CS_FAIL(CS_E_DTYPE, "qk_norm_rope: dtype");
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_sinusoid_f32(const float* t, float mult, int R, int C, float* out, hipStream_t s) {
    const int n = R * (C / 2);
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(sinusoid_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, mult, R, C, out);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_add3_f32(const float* a, const float* b, const float* c, float* out, long n, hipStream_t s) {
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(add3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, c, out, n);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_cast_f32(const float* x, void* out, long n, int dtype, hipStream_t s) {
    if (n <= 0) return CS_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (dtype == CS_BF16) hipLaunchKernelGGL(cast_kernel<bf16_el>, grid, block, 0, s, x, (u16*)out, n);
    else if (dtype == CS_F16) hipLaunchKernelGGL(cast_kernel<f16>, grid, block, 0, s, x, (u16*)out, n);
    else CS_FAIL(CS_E_DTYPE, "cast: dtype");
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_rms_norm(const void* x, const void* w, void* y, int M, int C, float eps, int dtype, hipStream_t s) {
    if (!x || !w || !y) CS_FAIL(CS_E_ARG, "rms_norm: null pointer");
    if (C % 8 || C > 8 * 64 * 8) CS_FAIL(CS_E_SHAPE, "rms_norm: C=%d unsupported", C);
    if (M <= 0) return CS_OK;
    const dim3 grid((M + 3) / 4), block(256);
    const int nv = (C / 8 + 63) / 64;
#define RMS(T, V) hipLaunchKernelGGL((rms_norm_kernel<T, V>), grid, block, 0, s, (const u16*)x, (const u16*)w, (u16*)y, M, C, eps)
    if (dtype == CS_BF16) { if (nv <= 1) RMS(bf16_el, 1); else if (nv <= 2) RMS(bf16_el, 2); else if (nv <= 4) RMS(bf16_el, 4); else RMS(bf16_el, 8); }
    else if (dtype == CS_F16) { if (nv <= 1) RMS(f16, 1); else if (nv <= 2) RMS(f16, 2); else if (nv <= 4) RMS(f16, 4); else RMS(f16, 8); }
    else CS_FAIL(CS_E_DTYPE, "rms_norm: dtype");
#undef RMS
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_gated_mul(const void* a, const void* b, void* out, long n, int dtype, hipStream_t s) {
    if (!a || !b || !out) CS_FAIL(CS_E_ARG, "gated_mul: null pointer");
    if (n % 8) CS_FAIL(CS_E_SHAPE, "gated_mul: element count must be a multiple of 8");
    if (n <= 0) return CS_OK;
    const dim3 grid((unsigned)((n / 8 + 255) / 256)), block(256);
    if (dtype == CS_BF16) hipLaunchKernelGGL(gated_mul_kernel<bf16_el>, grid, block, 0, s, (const u16*)a, (const u16*)b, (u16*)out, n / 8);
    else if (dtype == CS_F16) hipLaunchKernelGGL(gated_mul_kernel<f16>, grid, block, 0, s, (const u16*)a, (const u16*)b, (u16*)out, n / 8);
    else CS_FAIL(CS_E_DTYPE, "gated_mul: dtype");
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_embed_rows(const int64_t* ids, const void* table, void* out, long rows, int C, int vocab, hipStream_t s) {
    if (!ids || !table || !out) CS_FAIL(CS_E_ARG, "embed_rows: null pointer");
    if (C % 8 || vocab <= 0) CS_FAIL(CS_E_SHAPE, "embed_rows: bad dims");
    if (rows <= 0) return CS_OK;
    const long n = rows * (C / 8);
    hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, (const u16*)table, (u16*)out, rows, C, vocab);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
