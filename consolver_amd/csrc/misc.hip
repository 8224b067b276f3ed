// Small ops at the edges of the denoiser: timestep embedding MLP, tiny-M linear (time_emb_proj),
// conv_in (NCHW latents, 4 channels -> NHWC) and conv_out (NHWC -> NCHW, 4 channels).
// None of them is on the FLOP-critical path (<0.1 % of the forward); they are written to be
// coalesced and launch-light, not MFMA-shaped.
#include "ops.h"

namespace {

// emb[b][0:half] = cos(t * f_i), emb[b][half:] = sin(t * f_i), f_i = exp(-ln(10000) * i / half)
// (diffusers Timesteps(320, flip_sin_to_cos=True, downscale_freq_shift=0); cast to fp16 like t_emb.to(dtype))
__global__ void sinusoid_kernel(const float* __restrict__ t, int Bt, int C0, f16* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = C0 / 2;
    if (i >= Bt * half) return;
    const int b = i / half, k = i - b * half;
    const float f = expf(-9.210340371976184f * (float)k / (float)half);
    const float a = t[b] * f;
    out[(size_t)b * C0 + k] = (f16)cosf(a);
    out[(size_t)b * C0 + half + k] = (f16)sinf(a);
}

// out[r][n] = act(sum_k x[r][k] w[n][k] + b[n]) ; one wave per output column n, rows looped (R small)
__global__ __launch_bounds__(256) void rowvec_linear_kernel(const f16* __restrict__ x, int R, int K, const f16* __restrict__ w,
                                                            const f16* __restrict__ bias, int N, f16* __restrict__ out, int act_silu) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int KV = K >> 3;
    for (int r = 0; r < R; ++r) {
        float acc = 0.f;
        for (int kv = lane; kv < KV; kv += 64) {
            const f16x8 wv = *reinterpret_cast<const f16x8*>(w + (size_t)n * K + kv * 8);
            const f16x8 xv = *reinterpret_cast<const f16x8*>(x + (size_t)r * K + kv * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += (float)wv[k] * (float)xv[k];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float v = acc + (bias ? (float)bias[n] : 0.f);
            if (act_silu) v = v / (1.0f + __expf(-v));
            out[(size_t)r * N + n] = (f16)v;
        }
    }
}

// conv_in: thread = output pixel: its 3x3xCIN input patch is loaded ONCE into registers (the (pixel, 8-channel) mapping it replaces
// re-read every input value Cout/8 times and was bound by those scalar loads), then the thread walks the output channels 8 at a time
// against the weights [9*CIN][Cout] staged in LDS (wave-uniform addresses: broadcast reads) and stores 16 bytes per group.
template <int CIN>
__global__ __launch_bounds__(256) void conv_in_kernel(const f16* __restrict__ lat, int n_lat, int B, int H, int W,
                                                      const f16* __restrict__ w, const f16* __restrict__ bias, int Cout,
                                                      f16* __restrict__ out, f16* __restrict__ out_lo) {
    extern __shared__ __attribute__((aligned(16))) f16 ws[];   // [9*CIN][Cout]
    constexpr int KK = 9 * CIN;
    for (int i = threadIdx.x; i < KK * Cout; i += blockDim.x) {
        const int n = i / KK, k = i - n * KK;
        ws[k * Cout + n] = w[i];
    }
    __syncthreads();
    const int NG = Cout >> 3;
    const long total = (long)B * H * W;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < total; m += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(m % W), yo = (int)((m / W) % H), b = (int)(m / ((long)W * H));
        const f16* src = lat + (size_t)(b % n_lat) * CIN * H * W;
        float patch[KK];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yi = yo + tap / 3 - 1, xi = xo + tap % 3 - 1;
            const bool ok = yi >= 0 && yi < H && xi >= 0 && xi < W;
#pragma unroll
            for (int c = 0; c < CIN; ++c) patch[tap * CIN + c] = ok ? (float)src[((size_t)c * H + yi) * W + xi] : 0.f;
        }
        f16* o = out + (size_t)m * Cout;
        for (int ng = 0; ng < NG; ++ng) {
            float acc[8];
            const f16x8 bv = *reinterpret_cast<const f16x8*>(bias + ng * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = (float)bv[k];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const f16x8 wv = *reinterpret_cast<const f16x8*>(ws + kk * Cout + ng * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += patch[kk] * (float)wv[k];
            }
            f16x8 r;
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = (f16)acc[k];
            *reinterpret_cast<f16x8*>(o + ng * 8) = r;
            if (out_lo) {                                      // split-fp16 residual stream: the part the fp16 store dropped
                f16x8 l;
#pragma unroll
                for (int k = 0; k < 8; ++k) l[k] = (f16)(acc[k] - (float)r[k]);
                *reinterpret_cast<f16x8*>(out_lo + (size_t)m * Cout + ng * 8) = l;
            }
        }
    }
}

// conv_out: one wave per output pixel, lanes over 8-channel vectors of the NHWC input
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_kernel(const f16* __restrict__ x, int B, int Cin, int H, int W,
                                                       const f16* __restrict__ w, const f16* __restrict__ bias, f16* __restrict__ out,
                                                       int postprocess, int out_f32) {
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= (long)B * H * W) return;
    const int xo = (int)(m % W), yo = (int)((m / W) % H), b = (int)(m / ((long)W * H));
    const int CV = Cin >> 3;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int yi = yo + tap / 3 - 1, xi = xo + tap % 3 - 1;
        if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;      // wave-uniform
        const f16* px = x + (((size_t)b * H + yi) * W + xi) * Cin;
        for (int cv = lane; cv < CV; cv += 64) {
            const f16x8 v = *reinterpret_cast<const f16x8*>(px + cv * 8);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const f16x8 wv = *reinterpret_cast<const f16x8*>(w + ((size_t)o * 9 + tap) * Cin + cv * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[o] += (float)v[k] * (float)wv[k];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = wave_sum(acc[o]);
    if (lane < COUT) {
        float v = 0.f;
#pragma unroll
        for (int o = 0; o < COUT; ++o) if (lane == o) v = acc[o];
        v += (float)bias[lane];
        if (postprocess) v = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);      // (image / 2 + 0.5).clamp(0, 1), utils.py:29
        const size_t oi = (((size_t)b * COUT + lane) * H + yo) * W + xo;
        if (out_f32) reinterpret_cast<float*>(out)[oi] = v; else out[oi] = (f16)v;
    }
}


// conv_out for big images (the VAE's 128 -> 3 at 512 x 512): one workgroup = one 16 x 16 output patch; per 64-channel
// chunk the 18 x 18 input halo is staged in LDS once (128-byte rows, chunk index XOR (row & 7)) and every thread
// accumulates its pixel's COUT dot products with v_dot2_f32_f16; the (wave-uniform) weights come through the scalar cache.
// The one-wave-per-pixel kernel above re-read the 9-tap neighbourhood from L2 per pixel with a quarter of its lanes.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int COUT>
__global__ __launch_bounds__(256) void conv_out_patch_kernel(const f16* __restrict__ x, int Cin, int H, int W, const f16* __restrict__ w,
                                                             const f16* __restrict__ bias, f16* __restrict__ out, int postprocess, int out_f32) {
    __shared__ __attribute__((aligned(16))) f16 halo[324 * 64];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int PX = W >> 4;
    const int b = blockIdx.y, py = blockIdx.x / PX, px = blockIdx.x - py * PX;
    const int y0 = py * 16, x0 = px * 16;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        __syncthreads();
        for (int i = tid; i < 324 * 8; i += 256) {
            const int row = i >> 3, ch = i & 7;
            const int hy = row / 18, hx = row - hy * 18;
            const int y = y0 + hy - 1, xx = x0 + hx - 1;
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (y >= 0 && y < H && xx >= 0 && xx < W) v = *reinterpret_cast<const f16x8*>(x + (((size_t)b * H + y) * W + xx) * Cin + c0 + ch * 8);
            *reinterpret_cast<f16x8*>(halo + row * 64 + ((ch ^ (row & 7)) << 3)) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - 3 * dy;
            const int hr = (ty + dy) * 18 + tx + dx;
            const f16* hp = halo + hr * 64;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(hp + ((ch ^ (hr & 7)) << 3));
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const f16x8 wv = *reinterpret_cast<const f16x8*>(w + ((size_t)o * 9 + tap) * Cin + c0 + ch * 8);     // uniform -> s_load
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        acc[o] = __builtin_amdgcn_fdot2(f16x2{v[2 * k], v[2 * k + 1]}, f16x2{wv[2 * k], wv[2 * k + 1]}, acc[o], false);
                }
            }
        }
    }
    const int y = y0 + ty, xx = x0 + tx;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
        float v = acc[o] + (float)bias[o];
        if (postprocess) v = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);      // (image / 2 + 0.5).clamp(0, 1), utils.py:29
        const size_t oi = (((size_t)b * COUT + o) * H + y) * W + xx;
        if (out_f32) reinterpret_cast<float*>(out)[oi] = v; else out[oi] = (f16)v;
    }
}


// conv_out on the matrix cores (round 6).  The patch kernel above spends 5760 v_dot2 per thread on a tile whose input it loads with nothing else in flight: the
// UNet's conv_out (320 -> 4 at 64 x 64, batch 32: an 84 MB read) ran 110 us, 6x its memory floor; the VAE's (128 -> 3 at 512 x 512) 553 us per 16 images.  Here the
// same 16 x 16 patch / 18 x 18 x 64-channel swizzled LDS halo is the A operand of v_mfma_f32_16x16x32_f16 -- an M tile is one patch row (16 pixels), the tap shift
// is a row offset into the halo -- against a B fragment holding the COUT (<= 16) filters of one (tap, 32-channel step) in its first COUT columns, zeros elsewhere:
// 360 MFMAs per wave per patch (the padded columns cost matrix-pipe time nobody is waiting for), and the next 64-channel chunk's global loads are issued in front
// of the chunk being multiplied (register prefetch, single LDS buffer: 41.5 KB, three workgroups per CU).  What remains is the streaming read.
template <int COUT>
__global__ __launch_bounds__(256, 3) void conv_out_mfma_kernel(const f16* __restrict__ x, int Cin, int H, int W, const f16* __restrict__ w,
                                                            const f16* __restrict__ bias, f16* __restrict__ out, int postprocess, int out_f32,
                                                            const float* __restrict__ acc_in = nullptr) {
    static_assert(COUT >= 1 && COUT <= 16, "one 16-column MFMA tile");
    __shared__ __attribute__((aligned(16))) f16 halo[324 * 64];
    __shared__ __attribute__((aligned(16))) f16 wts[COUT * 9 * 64];      // this chunk's filters [COUT][tap][64]: staged with the halo, so that the fragment reads wait on lgkmcnt --
                                                                          // a global load issued after the prefetch could only be waited for together with it (vmcnt is in order)
    constexpr int NV = 324 * 8, NIT = (NV + 255) / 256;          // 16-byte vectors of one halo chunk; staging slots per thread
    constexpr int NWV = COUT * 72, NWS = (NWV + 255) / 256;      // ... of one chunk's filters
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int PX = W >> 4;
    const int b = blockIdx.y, py = blockIdx.x / PX, px = blockIdx.x - py * PX;
    const int y0 = py * 16, x0 = px * 16;
    // this thread's staging slots: source offset within the sample (-1 for the zero padding / the unused tail slot); the swizzled LDS position is recomputed at the write
    const f16* xb = x + (size_t)b * H * W * Cin;
    int src[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k;
        const int row = i >> 3, ch = i & 7;
        const int hy = row / 18, hx = row - hy * 18;
        const int y = y0 + hy - 1, xx = x0 + hx - 1;
        const bool ok = i < NV && y >= 0 && y < H && xx >= 0 && xx < W;
        src[k] = ok ? (y * W + xx) * Cin + ch * 8 : -1;
    }
    f16x8 pre[NIT], prew[NWS];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (src[k] >= 0) v = *reinterpret_cast<const f16x8*>(xb + src[k] + c0);
            pre[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NWS; ++k) {
            const int j = tid + 256 * k;                        // vector j = (filter, tap, 8-channel group) in wts order
            f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (j < NWV) v = *reinterpret_cast<const f16x8*>(w + (size_t)(j >> 3) * Cin + c0 + (j & 7) * 8);
            prew[k] = v;
        }
    };
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int col = lane & 15, kq = lane >> 4;                  // A: pixel `col` of the M tile, k slice kq; B: filter `col`, k slice kq
    const f16* wl = wts + (col < COUT ? col : 0) * 9 * 64 + kq * 8;
    prefetch(0);
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        __syncthreads();                                        // the previous chunk's fragment reads are done
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k;
            const int row = i >> 3, ch = i & 7;
            if (i < NV) *reinterpret_cast<f16x8*>(halo + row * 64 + ((ch ^ (row & 7)) << 3)) = pre[k];
        }
#pragma unroll
        for (int k = 0; k < NWS; ++k)
            if (tid + 256 * k < NWV) *reinterpret_cast<f16x8*>(wts + (tid + 256 * k) * 8) = prew[k];
        __syncthreads();
        if (c0 + 64 < Cin) prefetch(c0 + 64);                   // in flight under this chunk's MFMAs
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
            f16x8 bw[3][2];                                     // the (dy, dx, k step) filters of this tap row: zero columns beyond COUT
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (col < COUT) v = *reinterpret_cast<const f16x8*>(wl + (dy * 3 + dx) * 64 + ks * 32);
                    bw[dx][ks] = v;
                }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int kc = ks * 4 + kq;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int hr = (wave * 4 + mt + dy) * 18 + col + dx;
                        const f16x8 a = *reinterpret_cast<const f16x8*>(halo + hr * 64 + ((kc ^ (hr & 7)) << 3));
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bw[dx][ks], acc[mt], 0, 0, 0);
                    }
                }
        }
    }
    // D: column = filter (lane & 15), rows = pixels 4 (lane >> 4) + r of the patch row: one 8-byte NCHW store per lane and patch row
    if (col < COUT) {
        const float bv = bias ? (float)bias[col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int y = y0 + wave * 4 + mt;
            typedef f16 f16x4 __attribute__((ext_vector_type(4)));
            f16x4 o; f32x4 of;
            f32x4 prev = {0.f, 0.f, 0.f, 0.f};
            // acc_in: ADD the fp32 values of a first pass (the second pass of a hi + lo operand: launch_conv_out's x_lo; acc_in may be `out` itself)
            if (acc_in) prev = *reinterpret_cast<const f32x4*>(acc_in + ((((size_t)b * COUT + col) * H + y) * W + x0 + kq * 4));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mt][r] + bv + prev[r];
                if (postprocess) v = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);      // (image / 2 + 0.5).clamp(0, 1), utils.py:29
                o[r] = (f16)v; of[r] = v;
            }
            const size_t oi = (((size_t)b * COUT + col) * H + y) * W + x0 + kq * 4;
            if (out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + oi) = of;      // (the UNet's eps head for the native engine: no fp16 rounding of the denoiser's output)
            else *reinterpret_cast<f16x4*>(out + oi) = o;
        }
    }
}

// 16 x 16-patch conv_out: the MFMA form (default) or the v_dot2 one (cs_set_tuning("conv_out_mfma", 0))
template <int COUT>
static void launch_conv_out_patch(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, f16* out, int postprocess, hipStream_t s, int out_f32 = 0) {
    const dim3 grid((H / 16) * (W / 16), B);
    if (tune().conv_out_mfma) hipLaunchKernelGGL(conv_out_mfma_kernel<COUT>, grid, dim3(256), 0, s, x, Cin, H, W, w, bias, out, postprocess, out_f32);
    else hipLaunchKernelGGL(conv_out_patch_kernel<COUT>, grid, dim3(256), 0, s, x, Cin, H, W, w, bias, out, postprocess, out_f32);
}


__global__ void embed_tokens_kernel(const int64_t* __restrict__ ids, const f16* __restrict__ tok, const f16* __restrict__ pos, f16* __restrict__ out,
                                    long rows, int L, int C, int vocab) {
    const int CV = C >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * CV) return;
    const long r = i / CV; const int cv = (int)(i - r * CV);
    long id = ids[r]; id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f16x8 a = *reinterpret_cast<const f16x8*>(tok + id * C + cv * 8);
    const f16x8 b = *reinterpret_cast<const f16x8*>(pos + (r % L) * C + cv * 8);
    f16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (f16)((float)a[k] + (float)b[k]);
    *reinterpret_cast<f16x8*>(out + r * C + cv * 8) = o;
}

__global__ void quick_gelu_kernel(f16* __restrict__ x, long nv) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    f16x8 v = *reinterpret_cast<f16x8*>(x + i * 8);
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float f = (float)v[k]; v[k] = (f16)(f / (1.0f + __expf(-1.702f * f))); }
    *reinterpret_cast<f16x8*>(x + i * 8) = v;
}

// VAE latents with more than 4 channels (FLUX: 16): z = x * scale + shift [-> post_quant 1x1], written NHWC with the channel
// axis zero-padded to 64 so that conv_in runs through the implicit-GEMM kernel (K = 9 x 64)
__global__ void latent_to_nhwc64_kernel(const f16* __restrict__ x, const f16* __restrict__ w, const f16* __restrict__ b, f16* __restrict__ out,
                                        int B, int C, int HW, float in_scale, float in_shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int bi = (int)(i / HW), px = (int)(i - (long)bi * HW);
    float z[64];
    for (int c = 0; c < C; ++c) z[c] = (float)x[((size_t)bi * C + c) * HW + px] * in_scale + in_shift;
    f16* o = out + (size_t)i * 64;
    for (int oc = 0; oc < 64; oc += 8) {
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < 8 && oc + k < C; ++k) {
            float a;
            if (w) { a = (float)b[oc + k]; for (int c = 0; c < C; ++c) a += (float)w[(oc + k) * C + c] * z[c]; }
            else a = z[oc + k];
            v[k] = (f16)a;
        }
        *reinterpret_cast<f16x8*>(o + oc) = v;
    }
}

// softmax(scale * x) over long rows (8192 < cols <= 65536): one workgroup per row, two passes over the row in L2
__global__ __launch_bounds__(256) void row_softmax_block_kernel(f16* __restrict__ x, int cols, float scale) {
    __shared__ float red[4];
    __shared__ float bc[2];
    f16* p = x + (size_t)blockIdx.x * cols;
    const int CV = cols >> 3;
    float mx = -INFINITY;
    for (int cv = threadIdx.x; cv < CV; cv += 256) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(p + cv * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) mx = fmaxf(mx, (float)t[k]);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) bc[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * scale;
    __syncthreads();
    const float m = bc[0];
    float sum = 0.f;
    for (int cv = threadIdx.x; cv < CV; cv += 256) {
        const f16x8 t = *reinterpret_cast<const f16x8*>(p + cv * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += __expf((float)t[k] * scale - m);
    }
    sum = wave_sum(sum);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) bc[1] = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    const float inv = bc[1];
    for (int cv = threadIdx.x; cv < CV; cv += 256) {
        f16x8 t = *reinterpret_cast<const f16x8*>(p + cv * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = (f16)(__expf((float)t[k] * scale - m) * inv);
        *reinterpret_cast<f16x8*>(p + cv * 8) = t;
    }
}

__global__ void pixel_affine_kernel(const f16* __restrict__ x, int Cin, const f16* __restrict__ w, const f16* __restrict__ b, int Cout, f16* __restrict__ out,
                                    int B, int HW, float out_scale, float out_shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int bi = (int)(i / HW), px = (int)(i - (long)bi * HW);
    for (int o = 0; o < Cout; ++o) {
        float acc;
        if (w) { acc = (float)b[o]; for (int c = 0; c < Cin; ++c) acc += (float)w[o * Cin + c] * (float)x[((size_t)bi * Cin + c) * HW + px]; }
        else acc = (float)x[((size_t)bi * Cin + o) * HW + px];
        out[((size_t)bi * Cout + o) * HW + px] = (f16)((acc - out_shift) * out_scale);
    }
}

__global__ void pixel_linear_kernel(const f16* __restrict__ x, const f16* __restrict__ w, const f16* __restrict__ b, f16* __restrict__ out,
                                    int B, int C, int HW, float in_scale, float in_shift) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW) return;
    const int bi = (int)(i / HW), px = (int)(i - (long)bi * HW);
    for (int o = 0; o < C; ++o) {
        float acc = (float)b[o];
        for (int c = 0; c < C; ++c) acc += (float)w[o * C + c] * ((float)x[((size_t)bi * C + c) * HW + px] * in_scale + in_shift);
        out[((size_t)bi * C + o) * HW + px] = (f16)acc;
    }
}

// one wave per row; the row (cols <= 8192) is held in registers
__global__ __launch_bounds__(256) void row_softmax_kernel(f16* __restrict__ x, long rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    constexpr int MAXV = 16;
    const int CV = cols >> 3;
    float v[MAXV][8];
    float mx = -INFINITY;
    f16* p = x + row * cols;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            const f16x8 t = *reinterpret_cast<const f16x8*>(p + cv * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[j][k] = (float)t[k] * scale; mx = fmaxf(mx, v[j][k]); }
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j)
        if (lane + 64 * j < CV) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[j][k] = __expf(v[j][k] - mx); sum += v[j][k]; }
        }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
        const int cv = lane + 64 * j;
        if (cv < CV) {
            f16x8 o;
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (f16)(v[j][k] * inv);
            *reinterpret_cast<f16x8*>(p + cv * 8) = o;
        }
    }
}

}  // namespace

int launch_pixel_linear_nchw(const f16* x, const f16* w, const f16* b, f16* out, int B, int C, int HW, float in_scale, float in_shift, hipStream_t s) {
    if (!x || !w || !b || !out) CS_FAIL(CS_E_ARG, "pixel_linear: null pointer");
    const long n = (long)B * HW;
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(pixel_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, w, b, out, B, C, HW, in_scale, in_shift);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_row_softmax(f16* x, long rows, int cols, float scale, hipStream_t s) {
    if (!x) CS_FAIL(CS_E_ARG, "row_softmax: null pointer");
    if (cols % 8 || cols > 65536) CS_FAIL(CS_E_SHAPE, "row_softmax: cols=%d unsupported", cols);
    if (rows <= 0) return CS_OK;
    if (cols > 8192) {
        hipLaunchKernelGGL(row_softmax_block_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, cols, scale);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    hipLaunchKernelGGL(row_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, cols, scale);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_conv_out3(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, f16* out, int postprocess, hipStream_t s) {
    if (!x || !w || !bias || !out) CS_FAIL(CS_E_ARG, "conv_out3: null pointer");
    if (Cin % 8) CS_FAIL(CS_E_SHAPE, "conv_out3: Cin must be a multiple of 8");
    if (B <= 0) return CS_OK;
    if (H % 16 == 0 && W % 16 == 0 && Cin % 64 == 0) {
        launch_conv_out_patch<3>(x, B, Cin, H, W, w, bias, out, postprocess, s);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    const long M = (long)B * H * W;
    hipLaunchKernelGGL(conv_out_kernel<3>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, B, Cin, H, W, w, bias, out, postprocess, 0);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_rowvec_linear(const f16* x, int R, int K, const f16* w, const f16* b, int N, f16* out, int act_silu, hipStream_t s) {
    if (!x || !w || !out) CS_FAIL(CS_E_ARG, "rowvec_linear: null pointer");
    if (K % 8) CS_FAIL(CS_E_SHAPE, "rowvec_linear: K=%d must be a multiple of 8", K);
    if (R <= 0 || N <= 0) return CS_OK;
    hipLaunchKernelGGL(rowvec_linear_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, R, K, w, b, N, out, act_silu);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_time_embedding(const float* t, int Bt, int C0, int D, const f16* w1, const f16* b1, const f16* w2, const f16* b2,
                          f16* scratch, f16* out_silu, hipStream_t s) {
    if (!t || !scratch || !out_silu) CS_FAIL(CS_E_ARG, "time_embedding: null pointer");
    f16* emb = scratch;                      // [Bt][C0]
    f16* h1 = scratch + (size_t)Bt * C0;     // [Bt][D]
    const int n = Bt * (C0 / 2);
    hipLaunchKernelGGL(sinusoid_kernel, dim3((n + 255) / 256), dim3(256), 0, s, t, Bt, C0, emb);
    CS_CHECK_LAUNCH();
    int rc = launch_rowvec_linear(emb, Bt, C0, w1, b1, D, h1, 1, s);           // linear_1 + SiLU
    if (rc) return rc;
    return launch_rowvec_linear(h1, Bt, D, w2, b2, D, out_silu, 1, s);         // linear_2, then the resnets' SiLU
}

int launch_conv_in(const f16* lat, int n_lat, int B, int Cin, int H, int W, const f16* w, const f16* bias, int Cout, f16* out, hipStream_t s, f16* out_lo) {
    if (!lat || !w || !bias || !out) CS_FAIL(CS_E_ARG, "conv_in: null pointer");
    if (Cin != 4 || Cout % 8) CS_FAIL(CS_E_UNSUPPORTED, "conv_in: built for 4 input channels (got %d)", Cin);
    if (B <= 0) return CS_OK;
    const long total = (long)B * H * W;
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(conv_in_kernel<4>, dim3(grid), dim3(256), (size_t)36 * Cout * sizeof(f16), s, lat, n_lat, B, H, W, w, bias, Cout, out, out_lo);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_conv_out(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, int Cout, f16* out, hipStream_t s, int out_f32, const f16* x_lo,
                    float* scratch32) {
    if (!x || !w || !bias || !out) CS_FAIL(CS_E_ARG, "conv_out: null pointer");
    if (Cout != 4 || Cin % 8) CS_FAIL(CS_E_UNSUPPORTED, "conv_out: built for 4 output channels (got %d)", Cout);
    if (B <= 0) return CS_OK;
    if (H % 16 == 0 && W % 16 == 0 && Cin % 64 == 0) {
        if (x_lo) {
            // W (x + x_lo) + b: the hi plane's product (+ bias) in fp32 -- into `out` when that is fp32, into scratch32 otherwise -- then the lo plane's product on top of it,
            // written in the output's dtype: ONE rounding of the fp32-class value.  MFMA kernel only.
            if (!tune().conv_out_mfma) CS_FAIL(CS_E_ARG, "conv_out: a lo plane of the operand needs the MFMA kernel");
            if (!out_f32 && !scratch32) CS_FAIL(CS_E_ARG, "conv_out: a lo plane of the operand with a 16-bit output needs scratch32 [B][Cout][H][W]");
            float* first = out_f32 ? reinterpret_cast<float*>(out) : scratch32;
            const dim3 grid((H / 16) * (W / 16), B);
            hipLaunchKernelGGL(conv_out_mfma_kernel<4>, grid, dim3(256), 0, s, x, Cin, H, W, w, bias, reinterpret_cast<f16*>(first), 0, 1, (const float*)nullptr);
            CS_CHECK_LAUNCH();
            hipLaunchKernelGGL(conv_out_mfma_kernel<4>, grid, dim3(256), 0, s, x_lo, Cin, H, W, w, (const f16*)nullptr, out, 0, out_f32 ? 1 : 0, (const float*)first);
            CS_CHECK_LAUNCH();
            return CS_OK;
        }
        launch_conv_out_patch<4>(x, B, Cin, H, W, w, bias, out, 0, s, out_f32);
        CS_CHECK_LAUNCH();
        return CS_OK;
    }
    if (x_lo) CS_FAIL(CS_E_ARG, "conv_out: a lo plane of the operand is built for the 16 x 16-patch shapes");
    const long M = (long)B * H * W;
    hipLaunchKernelGGL(conv_out_kernel<4>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, B, Cin, H, W, w, bias, out, 0, out_f32);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_embed_tokens(const int64_t* ids, const f16* tok, const f16* pos, f16* out, long rows, int L, int C, int vocab, hipStream_t s) {
    if (!ids || !tok || !pos || !out) CS_FAIL(CS_E_ARG, "embed_tokens: null pointer");
    if (C % 8 || L <= 0 || vocab <= 0) CS_FAIL(CS_E_SHAPE, "embed_tokens: bad dims");
    if (rows <= 0) return CS_OK;
    const long n = rows * (C / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ids, tok, pos, out, rows, L, C, vocab);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_quick_gelu(f16* x, long n, hipStream_t s) {
    if (!x) CS_FAIL(CS_E_ARG, "quick_gelu: null pointer");
    if (n % 8) CS_FAIL(CS_E_SHAPE, "quick_gelu: element count must be a multiple of 8");
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, s, x, n / 8);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_latent_to_nhwc64(const f16* x, const f16* w, const f16* b, f16* out, int B, int C, int HW, float in_scale, float in_shift, hipStream_t s) {
    if (!x || !out || (w && !b)) CS_FAIL(CS_E_ARG, "latent_to_nhwc64: null pointer");
    if (C < 1 || C > 64) CS_FAIL(CS_E_SHAPE, "latent_to_nhwc64: %d channels unsupported", C);
    const long n = (long)B * HW;
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(latent_to_nhwc64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, w, b, out, B, C, HW, in_scale, in_shift);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_conv_out_small(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, int Cout, f16* out, hipStream_t s) {
    if (!x || !w || !bias || !out) CS_FAIL(CS_E_ARG, "conv_out_small: null pointer");
    if (H % 16 || W % 16 || Cin % 64) CS_FAIL(CS_E_SHAPE, "conv_out_small: H, W must be multiples of 16 and Cin of 64");
    if (B <= 0) return CS_OK;
    if (Cout == 4) launch_conv_out_patch<4>(x, B, Cin, H, W, w, bias, out, 0, s);
    else if (Cout == 8) launch_conv_out_patch<8>(x, B, Cin, H, W, w, bias, out, 0, s);
    else if (Cout == 16) launch_conv_out_patch<16>(x, B, Cin, H, W, w, bias, out, 0, s);
    else CS_FAIL(CS_E_UNSUPPORTED, "conv_out_small: %d output channels not built", Cout);
    CS_CHECK_LAUNCH();
    return CS_OK;
}

int launch_pixel_affine_nchw(const f16* x, int Cin, const f16* w, const f16* bias, int Cout, f16* out, int B, int HW, float out_scale, float out_shift, hipStream_t s) {
    if (!x || !out || (w && !bias)) CS_FAIL(CS_E_ARG, "pixel_affine: null pointer");
    if (!w && Cout > Cin) CS_FAIL(CS_E_SHAPE, "pixel_affine: identity form needs Cout <= Cin");
    const long n = (long)B * HW;
    if (n <= 0) return CS_OK;
    hipLaunchKernelGGL(pixel_affine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, Cin, w, bias, Cout, out, B, HW, out_scale, out_shift);
    CS_CHECK_LAUNCH();
    return CS_OK;
}
