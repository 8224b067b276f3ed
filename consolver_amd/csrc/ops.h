// Internal op-level launchers shared by the UNet executor (unet.cpp) and the
// exported test entry points (include/consolver_hip_ops.h).
#pragma once
#include "common.h"

struct IgemmArgs {
    // activations, NHWC fp16; up to two sources concatenated along channels (skip-concat fusion)
    const f16* a0; const f16* a1; int c0, c1;
    int B, Hi, Wi;          // input spatial size (before the optional nearest x2 upsample)
    int Ho, Wo;             // output spatial size
    int taps;               // 1 = 1x1 conv / linear, 9 = 3x3 conv (pad 1)
    int stride;             // 1 or 2 (3x3 only)
    int upsample;           // 1: nearest x2 upsample fused into the gather
    int N;                  // output channels (rows of w)
    const f16* w;           // [N][taps*(c0+c1)] K contiguous, tap-major / channel-minor
    const f16* bias;        // [N] or null
    const f16* temb; int temb_stride;   // per-sample [B][>=N] add (time embedding) or null; stride 0 broadcasts
    const f16* res;         // [M][N] residual add or null (may alias out)
    f16* out;               // [M][N] (GEGLU: [M][N/2])
    int geglu;              // rows of w pre-permuted in (value16 | gate16) blocks; out = v * gelu(g)
    float* splitk_ws; size_t splitk_ws_bytes;   // optional fp32 scratch for split-K on small images (may be null)
};
int launch_igemm(const IgemmArgs& a, hipStream_t s);
double igemm_flops(const IgemmArgs& a);

struct AttnArgs {
    const f16* q; int q_stride;   // [B, Nq, H*dh] rows of q_stride halfs
    const f16* k; int k_stride;   // [B, Nk, ...]
    const f16* v; int v_stride;
    f16* out; int out_stride;     // [B, Nq, H*dh]
    int B, H, Nq, Nk, dh;
    float scale;
};
int launch_attention(const AttnArgs& a, hipStream_t s);

// GroupNorm over NHWC [B][HW][C0+C1] (two-source concat), 32 groups.
// stats -> partial[B][S][C] (sum, sumsq); apply normalises (+SiLU) into out [B][HW][C].
struct GroupNormArgs {
    const f16* x0; const f16* x1; int c0, c1;
    int B, HW, groups; float eps; int silu;
    const f16* gamma; const f16* beta;     // [C]
    float* partial;                         // workspace >= B * GN_SPLITS * C * 2 floats
    f16* out;
};
#define GN_SPLITS 16
int launch_group_norm(const GroupNormArgs& a, hipStream_t s);

int launch_layer_norm(const f16* x, const f16* gamma, const f16* beta, f16* out, int M, int C, float eps, hipStream_t s);

// timestep sinusoid (flip_sin_to_cos, shift 0) -> Linear -> SiLU -> Linear -> SiLU (the SiLU that
// every resnet applies before time_emb_proj) ; out_silu [Bt][D] fp16
int launch_time_embedding(const float* t, int Bt, int C0, int D, const f16* w1, const f16* b1, const f16* w2, const f16* b2,
                          f16* scratch, f16* out_silu, hipStream_t s);
// out[r][n] = sum_k x[r][k] w[n][k] + b[n]  (tiny M; one wave per n)
int launch_rowvec_linear(const f16* x, int R, int K, const f16* w, const f16* b, int N, f16* out, int act_silu, hipStream_t s);

// conv_in: NCHW fp16 latents (n_lat samples, sample b reads b % n_lat) -> NHWC [B][H][W][Cout], 3x3 pad 1
int launch_conv_in(const f16* lat, int n_lat, int B, int Cin, int H, int W, const f16* w /*[Cout][9][Cin]*/, const f16* bias,
                   int Cout, f16* out, hipStream_t s);
// conv_out: NHWC [B][H][W][Cin] -> NCHW [B][Cout][H][W], 3x3 pad 1 (Cout small)
int launch_conv_out(const f16* x, int B, int Cin, int H, int W, const f16* w /*[Cout][9][Cin]*/, const f16* bias, int Cout,
                    f16* out, hipStream_t s);
