/*
 * consolver_hip_ops.h -- op-level entry points of libconsolver_hip.so.
 *
 * These are the building blocks the UNet executor (cs_unet_forward) is made of, exported so
 * that each HIP kernel can be parity-tested on its own against a plain fp32 reference
 * (tests/test_ops_gpu.py).  They replace, one for one, the third-party torch/diffusers ops the
 * reference's denoiser call (denoise_ppo.py:89-94) runs: F.conv2d / nn.Linear (+bias, +time
 * embedding, +residual, GEGLU), F.scaled_dot_product_attention (xformers in the reference,
 * gen_ppo.py:197-198), nn.GroupNorm(+SiLU), nn.LayerNorm.
 *
 * Layout: activations are NHWC fp16 ([B, H, W, C] == [B, HW, C]); weights are fp16,
 * [Cout][kh*kw][Cin] for convolutions and [N][K] for linear layers.  All pointers are borrowed
 * device pointers; calls are asynchronous on `stream`; return codes as in consolver_hip.h.
 */
#ifndef CONSOLVER_HIP_OPS_H
#define CONSOLVER_HIP_OPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* conv (taps = 9: 3x3 pad 1, stride 1|2, optional fused nearest-x2 upsample of the input; taps = 1: 1x1)
 * over the channel concatenation of x0 [B,Hi,Wi,c0] and x1 [B,Hi,Wi,c1] (x1 may be NULL, c1 = 0).
 * out[B,Ho,Wo,N] = conv + bias[N] + temb[b*temb_stride + n] + res[B,Ho,Wo,N]  (each optional).
 * splitk_ws: optional fp32 scratch (device) that lets small-image 3x3 convs split their channel chunks over
 * more workgroups (needs splits * M * N * 4 bytes; NULL disables). */
int cs_op_conv2d(const void* x0, int c0, const void* x1, int c1, int B, int Hi, int Wi, int taps, int stride, int upsample,
                 const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, void* out,
                 void* splitk_ws, size_t splitk_ws_bytes, void* stream);

/* cs_op_conv2d that also leaves the GroupNorm statistics of its OUTPUT in gn_stats (device, fp32):
 *   gn_stats[B][Ho*Wo/64][N/2][2] = (sum, sum of squares) of the channel pair (2q, 2q+1) over each 64-pixel block of a sample
 * (the block order inside a sample is the kernel's own; only sums over all blocks are meaningful).  Written by the conv epilogue from the
 * final fp16 values (after bias / temb / residual), or by a statistics pass when the chosen kernel has none (split-K forms).
 * Needs Ho*Wo % 64 == 0.  Consumer: cs_op_group_norm_pre.  (diffusers ResnetBlock2D: norm2 follows conv1, the next block's norm1 follows conv2.) */
int cs_op_conv2d_gn(const void* x0, int c0, const void* x1, int c1, int B, int Hi, int Wi, int taps, int stride, int upsample,
                    const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, void* out,
                    void* splitk_ws, size_t splitk_ws_bytes, float* gn_stats, void* stream);
/* nearest-x2 upsample + 3x3 conv in its SUB-PIXEL form (round 6; replaces the upsample = 1 call of cs_op_conv2d for the UNet's three upsamplers,
 * reference: diffusers Upsample2D = F.interpolate(scale 2, nearest) + Conv2d(3, pad 1), called from gen_pretrain/pipeline.py:1058-1066's UNet).  Output pixel
 * (2 y + py, 2 x + px) reads the 2 x 2 input neighbourhood rows {y - 1 + py, y + py} x columns {x - 1 + px, x + px}; the filter taps that land on one neighbour
 * are summed ONCE on the host: cs_op_conv_up_fold_pack: w [N][9 Cin] (tap-major, host) -> out [4 phases = 2 py + px][N][4 Cin] (host), sums in fp32, one rounding
 * to fp16.  16 multiplies per input pixel and channel pair instead of 36; the results differ from the plain form by that one rounding of the weights (~2^-12).
 * cs_op_conv_up_sub: x NHWC [B, Hi, Wi, Cin] -> out [B, 2 Hi, 2 Wi, N]; input a multiple of 16 x 16 (N % 160 == 0 or N % 128 == 0: the UNet's and the VAE decoder's
 * widths) or 8 x 8 (N % 160 == 0), Cin % 64 == 0; w = the plain packed
 * filter (unused by the kernel, kept for the shapes the sub-pixel kernel does not take), gn_stats as cs_op_conv2d_gn or null. */
int cs_op_conv_up_fold_pack(const void* w, int N, int Cin, void* out);
int cs_op_conv_up_sub(const void* x, int Cin, int B, int Hi, int Wi, const void* w, const void* w_sub, const void* bias, int N, void* out, float* gn_stats,
                      void* stream);

/* out[M,N] = x[M,K] w[N,K]^T + bias + res ; geglu != 0: w rows pre-permuted in (16 value | 16 gate)
 * blocks (see cs_op_geglu_pack) and out[M,N/2] = value * gelu(gate). */
int cs_op_linear(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, void* out, int geglu, void* stream);

/* host helper: permute a [2*Hd, K] GEGLU projection (rows [0,Hd) value, [Hd,2Hd) gate; fp16) and its
 * bias into the blocked order cs_op_linear(geglu=1) expects.  Pure host memory. */
int cs_op_geglu_pack(const void* w_host, const void* b_host, int Hd, int K, void* w_out_host, void* b_out_host);

/* softmax(scale * q k^T) v per (batch, head); rows are token-major with the heads interleaved,
 * q/k/v/out row strides in halfs (so fused qkv buffers work).  dh in {40, 80, 160}. */
int cs_op_attention(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                    int B, int H, int Nq, int Nk, int dh, float scale, void* stream);

/* GroupNorm(groups) [+ SiLU] over the channel concatenation of x0/x1 ([B,HW,c0], [B,HW,c1]) -> out [B,HW,c0+c1].
 * workspace: cs_op_group_norm_workspace(B, c0 + c1) bytes.  c0, c1 multiples of 8; (c0 + c1) / groups EVEN (the partial sums are kept per
 * channel pair; every SD1.5 / AutoencoderKL width / 32 is): CS_E_SHAPE otherwise. */
size_t cs_op_group_norm_workspace(int B, int C);
int cs_op_group_norm(const void* x0, int c0, const void* x1, int c1, int B, int HW, int groups, float eps, int silu,
                     const void* gamma, const void* beta, void* workspace, void* out, void* stream);

/* cs_op_group_norm with the statistics pass skipped for the sources whose producer already wrote them (cs_op_conv2d_gn layout, HW/64
 * blocks per sample); stats0 / stats1 may be NULL (that source is then reduced here).  HW % 64 == 0. */
int cs_op_group_norm_pre(const void* x0, int c0, const float* stats0, const void* x1, int c1, const float* stats1, int B, int HW, int groups,
                         float eps, int silu, const void* gamma, const void* beta, void* workspace, void* out, void* stream);

/* conv_out: 3x3 conv (pad 1) from NHWC x[B][H][W][Cin] to NCHW out[B][Cout][H][W], Cout small -- the UNet's 320 -> 4 eps head (diffusers
 * UNet2DConditionModel.conv_out behind denoise_ppo.py:89-94) and the AutoencoderKL decoder's 128 -> 3 (utils.py:6-34).  w [Cout][9][Cin] (tap-major,
 * channel-minor), bias [Cout].  Cout in {3, 4} (any H, W; Cin % 8 == 0) -- on 16 x 16 patches with Cin % 64 == 0 the matrix-core kernel -- or
 * {8, 16} (H, W % 16 == 0, Cin % 64 == 0: the encoder's moments head).  postprocess (Cout 3 only): out = (v / 2 + 0.5).clamp(0, 1). */
int cs_op_conv_out(const void* x, int B, int Cin, int H, int W, const void* w, const void* bias, int Cout, void* out, int postprocess, void* stream);

/* Fused cross-attention sub-block of the SD1.5 transformer block (diffusers BasicTransformerBlock: norm2 -> attn2 -> residual; reference
 * call site denoise_ppo.py:89-94) at C = 320, 8 heads, <= 80 context keys:
 *   out[M,C] = h + (softmax(scale * (LayerNorm(h) wq^T) K^T) V) wo^T + bo,   kv[M/HW][Nk][2C] = (K | V) projections of the context.
 * One kernel instead of LayerNorm + to_q GEMM + attention + to_out GEMM: the three [M,C] intermediates stay in LDS.  out may alias h. */
int cs_op_xattn_block(const void* h, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv, int Nk,
                      const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* stream);

/* LayerNorm over the last dim of x[M,C] */
int cs_op_layer_norm(const void* x, const void* gamma, const void* beta, void* out, int M, int C, float eps, void* stream);

/* ---- split-fp16 residual-stream forms (CS_RESIDUAL_F16X2, include/consolver_hip.h) ----------------------------------------------
 * A stream tensor is two fp16 planes, value = hi + lo.  `res_lo` (may be NULL) is the lo plane of `res`; `out_lo` (may be NULL) receives
 * f16(v - float(f16(v))) for the fp32 value v whose fp16 rounding goes to `out`.  The adds run in fp32 on hi + lo.  Same kernels as the plain
 * forms (the lo planes ride in the fp32 epilogue).  x0_lo / x1_lo / x_lo (NULL = plain fp16 operand; taps = 1 only) are the lo planes of a split-fp16 A
 * operand: the k loop runs over the hi planes and then over the lo planes against the same weights, i.e. w (hi + lo) exactly at twice the MFMA work
 * (the resnet shortcut 1x1 over [x | skip], which consumes the residual stream directly). */
int cs_op_conv2d_x2(const void* x0, const void* x0_lo, int c0, const void* x1, const void* x1_lo, int c1, int B, int Hi, int Wi, int taps, int stride,
                    int upsample, const void* w, const void* bias, int N, const void* temb, int temb_stride, const void* res, const void* res_lo,
                    void* out, void* out_lo, float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cs_op_linear_x2(const void* x, const void* x_lo, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo,
                    void* out, void* out_lo, float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* row_stats (may be NULL; then row_groups too): the layer also leaves, per OUTPUT row, (sum, sum of squares) over G column groups, row_stats[M][G][2] fp32
 * (device; room for N / 64 groups), G written to *row_groups (host) -- from the epilogue's fp32 values, or by a pass over the output where the chosen kernel has no
 * such epilogue (split-K forms; G = 1).  Consumer: cs_op_linear_ln. */

/* LayerNorm folded into the linear layer that consumes it (diffusers BasicTransformerBlock: norm1 -> to_q | to_k | to_v, norm2 -> attn2.to_q, norm3 -> GEGLU proj):
 *   LN(h) W^T + b = rstd (h W'^T - mean s) + b',  W' = fp16(W diag(gamma)),  s[n] = sum_k W'[n][k],  b' = W beta + b.
 * cs_op_ln_fold_pack (host memory only) builds W' (fp16 [N][K]), s and b' (fp32 [N]) from W, the optional bias, gamma and beta (fp16).
 * cs_op_linear_ln multiplies the RAW hidden state x[M,K] (fp16; the hi plane of a split-fp16 stream) with W' and applies (mean, rstd) of each row in the epilogue,
 * taken from the row statistics its producer left (cs_op_linear_x2 / cs_op_conv2d_x2 / cs_op_xattn_block_x2 row_stats, or cs_op_row_stats).  geglu != 0: W' rows
 * pre-permuted with cs_op_geglu_pack BEFORE folding; out[M, N/2].  No LayerNorm kernel and no normalised copy of x exist on this path. */
int cs_op_ln_fold_pack(const void* w_host, const void* bias_host, const void* gamma_host, const void* beta_host, int N, int K, void* w_out_host,
                       float* s_out_host, float* b_out_host);
int cs_op_linear_ln(const void* x, int M, int K, const void* w_folded, const float* ln_s, const float* ln_b, int N, const float* row_stats, int groups,
                    float eps, void* out, int geglu, void* splitk_ws, size_t splitk_ws_bytes, void* stream);
/* stats[M][1][2] = (sum, sum of squares) of every row of x[M,C] (+ x_lo[M,C] when not NULL) */
int cs_op_row_stats(const void* x, const void* x_lo, int M, int C, float* stats, void* stream);
/* GroupNorm / LayerNorm of split-fp16 sources (x*_lo may be NULL = plain fp16 source) */
int cs_op_group_norm_x2(const void* x0, const void* x0_lo, int c0, const void* x1, const void* x1_lo, int c1, int B, int HW, int groups,
                        float eps, int silu, const void* gamma, const void* beta, void* workspace, void* out, void* stream);
int cs_op_layer_norm_x2(const void* x, const void* x_lo, const void* gamma, const void* beta, void* out, int M, int C, float eps, void* stream);
/* cs_op_xattn_block on a split-fp16 residual stream: the residual add takes h + h_lo and writes out + out_lo (h_lo = out_lo = NULL: plain fp16 stream);
 * row_stats (may be NULL): [M][1][2] row statistics of the output for a folded norm3 */
int cs_op_xattn_block_x2(const void* h, const void* h_lo, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv,
                         int Nk, const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* out_lo, float* row_stats,
                         void* stream);

/* 8-bit lo planes (round 6).  Inside a transformer block the hidden state's lo plane is only ever ADDED (to_out / cross-attention / feed-forward epilogues), never
 * multiplied, so it is kept as ONE BYTE per element: e5m2 (sign, the fp16 exponent, two mantissa bits = the fp16 lo value rounded to its top byte; OCP e5m2, what
 * torch.float8_e5m2 holds), converted two elements per instruction (v_cvt_pk_bf8_f32 / v_cvt_pk_f32_bf8).  hi + lo8 carries 14 significant bits against 22 with an fp16
 * lo plane; on the SD1.5 UNet that moves the per-forward eps error from 8.8224e-4 to 8.8209e-4 (tools/sim_precision_r06.py) and halves the lo plane's bytes.
 * The forms below are cs_op_linear_x2 / cs_op_xattn_block_x2 / cs_op_row_stats with res_lo8 / out_lo8 / h_lo8 / x_lo8 as [M][N] BYTE planes (same element layout);
 * the GEMM's A operand is a plain fp16 tensor (an 8-bit plane is not an MFMA operand here). */
int cs_op_linear_lo8(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo8, void* out, void* out_lo8,
                     float* row_stats, int* row_groups, void* splitk_ws, size_t splitk_ws_bytes, void* stream);
int cs_op_xattn_block_lo8(const void* h, const void* h_lo8, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* wq, const void* kv,
                          int Nk, const void* wo, const void* bo, int M, int HW, int C, int heads, float scale, void* out, void* out_lo8, float* row_stats,
                          void* stream);
int cs_op_row_stats_lo8(const void* x, const void* x_lo8, int M, int C, float* stats, void* stream);
/* the detector behind cs_unet_calibrate_ln_fold (round 6): *sum += sum over the M rows of mean^2 / (var + eps), the moments taken from row statistics
 * [M][groups][2] as a producer epilogue or cs_op_row_stats leaves them.  sqrt(*sum / M) is the RMS of |mean| / sigma over the rows: how many sigma the
 * rows of a hidden state sit away from zero, i.e. how many times 2^-11 a LayerNorm folded into the next GEMM loses on them (it multiplies the hi plane).
 * The executor unfolds a block whose figure exceeds the caller's bound (default 4).  *sum must be zeroed by the caller; float atomics, order-free. */
int cs_op_ln_dc_ratio(const float* row_stats, int M, int groups, int C, float eps, float* sum, void* stream);

/* transformer GEMM (f16 / bf16, dtype = CS_F16 1 | CS_BF16 2): out[m][n] = act(x[m,:] . w[n,:] + bias[n]) (+ res, * gate);
 * w must have ceil(N/256)*256 rows (zero padded).  act: 0 none, 1 GELU(tanh).  gate: fp32 [M / rows_per_sample][gate_stride]. */
int cs_op_gemm2(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const float* gate, long gate_stride,
                int rows_per_sample, int act, void* out, long ldc, int col_off, int dtype, void* stream);
/* cs_op_gemm2's gated-residual form on a SPLIT residual stream (round 5; FLUX hidden states as hi + lo planes of the model dtype, value = hi + lo):
 * out + out_lo = (res + res_lo) + gate * T(x . w + bias), the sum taken in fp32; out_lo = T(value - float(out)).  res / res_lo / out / out_lo are [M][N]; out may
 * alias res and out_lo res_lo.  tail_ws: optional split-K tail scratch (cs_op_gemm2_workspace).  Replaces `hidden_states = hidden_states + gate * attn_output` and
 * its three siblings in diffusers' FluxTransformerBlock / FluxSingleTransformerBlock (call site edit_ppo/pipeline.py:1087-1097). */
int cs_op_gemm2_x2(const void* x, int M, int K, const void* w, const void* bias, int N, const void* res, const void* res_lo, const float* gate, long gate_stride,
                   int rows_per_sample, void* out, void* out_lo, int dtype, void* tail_ws, size_t tail_ws_bytes, void* stream);
/* y = LayerNorm_noaffine(x + x_lo) * (1 + scale[b]) + shift[b]: the adaLN LayerNorm reading a split stream (x_lo = NULL: one plane) */
int cs_op_ln_modulate_x2(const void* x, const void* x_lo, void* y, int M, int C, int rows_per_sample, const float* shift, const float* scale, long mod_stride, float eps,
                         int dtype, void* stream);
/* Grouped form: two independent cs_op_gemm2 problems of the same dtype in ONE launch (problem b's tiles are appended to problem a's tile
 * list).  FLUX's double-stream blocks (reference: diffusers FluxTransformerBlock, called from FLUX/train_ppo_flux.py:150-163 through
 * pipe.transformer) run the image-token and the text-token linear of each stage this way. */
typedef struct CsGemm2Problem {
    const void* x; int M, K; const void* w; const void* bias; int N;
    const void* res; const float* gate; long gate_stride; int rows_per_sample; int act;
    void* out; long ldc; int col_off;
} CsGemm2Problem;
int cs_op_gemm2_pair(const CsGemm2Problem* a, const CsGemm2Problem* b /* may be NULL */, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Scratch for the split-K tail of a launch of `tiles` 256 x 256 output tiles (both problems together) with inner dimension K: when the last
 * round of tiles (one per CU) is partly empty and K >= 6144, that round's tiles are computed as 2-6 k ranges side by side into fp32 partial
 * tiles and summed (same rounding points as the unsplit epilogue; the fp32 summation order differs).  0: the launch never splits;
 * workspace may be NULL (no split). */
size_t cs_op_gemm2_workspace(int tiles, int K);
/* cs_op_attention with an explicit dtype (bf16: head dim 128) */
int cs_op_attention_ex(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                       int B, int H, int Nq, int Nk, int dh, float scale, int dtype, void* stream);

/* cs_op_attention_ex with scratch memory for the split-KV tail: when the workgroup count leaves a last round that fills under half of the
 * chip (head dim 128, e.g. FLUX-Kontext's 8704-token joint sequence), that round's query blocks are computed once per key range and merged.
 * cs_op_attention_workspace returns the bytes to provide (0: the shape never splits); results equal cs_op_attention_ex up to fp32 rounding. */
size_t cs_op_attention_workspace(int B, int H, int Nq, int Nk, int dh);
int cs_op_attention_ws(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                       int B, int H, int Nq, int Nk, int dh, float scale, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* causal form (key j visible to query i iff j <= i): f16, head dim 64, Nq == Nk = N (CLIP text encoder) */
int cs_op_attention_causal(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                           int B, int H, int N, int dh, float scale, void* stream);

/* T5 encoder pieces (dtype = CS_F16 1 | CS_BF16 2) */
int cs_op_rms_norm(const void* x, const void* weight, void* out, int M, int C, float eps, int dtype, void* stream);
int cs_op_gated_mul(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
int cs_op_embed_rows(const int64_t* ids, const void* table, void* out, int64_t rows, int C, int vocab, void* stream);
/* softmax(scale q k^T + bias) v with bias_log2e = bias * log2(e) as fp32 [H][N][N]; head dim 64, no mask */
int cs_op_attention_bias(const void* q, int q_stride, const void* k, int k_stride, const void* v, int v_stride, void* out, int out_stride,
                         int B, int H, int N, int dh, float scale, const float* bias_log2e, int dtype, void* stream);

/* kernel-selection knobs (tests / tuning):
 *   "conv_halo": 0 never, 1 auto (default), 2 whenever the shape allows, 3 also force the 256x320 / 256x256 k32 tiles, 4 never use those;
 *   "gemm_big":  0 off, 1 auto (default), 2 force the 256x320 GEMM, 3 force the 256x160 GEMM;
 *   "attn_qt40": query tiles per wave at head dim 40 (2 | 4, default 4);
 *   "xattn_fused": 1 (default): the cross-attention sub-block at C = 320 runs as one kernel (cs_op_xattn_block) inside cs_unet_forward;
 *   "cfg_share": 1 (default) evaluate the CFG halves' common prefix once (cs_unet_forward with dup = 2), 0 full dual batch;
 *   "x2_split_a": CS_RESIDUAL_F16X2 only, bit mask of the GEMMs that read hi + lo of the residual stream (cs_op_conv2d_x2 x0_lo) inside cs_unet_forward:
 *                1 (default) the resnet shortcut 1x1, 2 proj_out, 3 both, 0 none;
 *   "xcd_grid":  1 (default) weight-heavy conv / linear launches map the 8 XCDs as a 2-D grid over (row tiles, column tiles) so that each L2 streams a part
 *                of the weights instead of all of them, 0 contiguous tile runs per XCD always;
 *   "epi_fast":  bit 0: conv / linear epilogues that add a residual (+ its lo plane) or a time embedding issue those loads ahead of their use, branch-free
 *                (igemm_epilogue_impl FAST), bit 1 (round 6): also the form "residual + its lo plane, no lo plane out" (the feed-forward's second linear in the
 *                split mode); 3 (default), 0 the generic load-where-added code; bit-identical, see profiles/r04_ab_epi_fast_*.txt;
 *   "x2_sc_skip": bit i: the i-th resnet shortcut 1x1 (creation order) reads the hi plane only although x2_split_a bit 0 is set (default 0; 0x78 = the four 2560 -> 1280
 *                shortcuts at the 8 x 8 / 16 x 16 levels whose hi + lo operand buys the least: -0.12 ms per forward, +3.5 % on the tightest gated number);
 *   "lo8":       1 (default) inside cs_unet_forward (CS_RESIDUAL_F16X2) the transformer blocks' hidden state carries an 8-bit e5m2 lo plane (cs_op_linear_lo8),
 *                0 an fp16 one;
 *   "up_fold": the UNet's three upsamplers in the sub-pixel form (cs_op_conv_up_sub: 16 instead of 36 multiplies per input pixel, taps summed on the host and
 *                rounded to fp16 once more): 0 never, 1 (default) in forwards whose residual stream is one fp16 plane, 2 also on the split stream;
 *   "head_x2": 1 (default) with the split stream the UNet's output head keeps its GroupNorm + SiLU output as hi + lo planes and conv_out multiplies both (a second
 *                pass over the lo plane on top of the first pass's fp32 result; the output, fp32 or the model dtype, is rounded from that once), 0 one fp16 plane as in round 5;
 *   "conv_out_mfma": 1 (default) the 16 x 16-patch conv_out kernels (cs_op_conv_out) on the matrix cores, 0 the v_dot2 patch kernel;
 *   "conv_in_mfma": 1 (default) the UNet's conv_in runs on the MFMA conv kernel over latents zero-padded to 64 channels, 0 the scalar conv_in kernel;
 *   "ln_fold":   1 (default) the transformer blocks' LayerNorms are folded into the linear layers that consume them inside cs_unet_forward (cs_op_linear_ln),
 *                0 LayerNorm kernel + plain GEMM;
 *   "gn_fuse":   1 (default) GroupNorm statistics of conv / 1x1 outputs come from the producer's epilogue inside cs_unet_forward and
 *                cs_op_conv2d_gn, 0 always a separate statistics pass;
 *   "conv_lw":   1 (default) stride-1 3x3 convolutions with N % 160 == 0 (BN 160: the UNet) or N % 128 == 0 (BN 128: the VAE) run on the loader-wave kernel
 *                (conv3_lw_kernel: waves 0-3 multiply, waves 4-7 stage), 2 the same without its immediate-offset (FAST) addressing, 3 BN 160 only, 0 the 8-wave
 *                halo kernels;
 *   "gemm_w8":   1 (default) the 256x320 GEMM runs its hand-scheduled k loop (gemm_w8_kernel; needs 32-bit operand offsets), 0 the
 *                compiler-scheduled gemm_big_kernel (bit-identical results);
 *   "gemm_lw":   1 (default) layers served by the 256x160 GEMM tile run the loader-wave kernel (gemm_lw_kernel), 0 gemm_big_kernel<.,160>;
 *   "gemm2_w8":  1 (default) cs_op_gemm2 / cs_op_gemm2_pair run the hand-scheduled k loop of gemm2_kernel where 32-bit operand offsets suffice (bit-identical),
 *                0 the compiler-scheduled loop;
 *   "attn_lw":   1 (default) head dim 40 self-attention with Nq % 256 == 0 and Nk % 64 == 0 runs attn40_lw_kernel, 2 the same with 16x16x32 MFMAs for both
 *                score k steps (bit-identical to attn_kernel), 0 attn_kernel;
 *   "gemm_gm":   tile order of the 256-row GEMM kernel: -1 (default) bands of 4 tile rows when there are >= 12 tile columns, 0 / 1 row-major, n bands of n;
 *   "attn_prio" / "gemm2_prio": static wave priority experiments (attn_prio -1 = auto: head dim 128 only);
 *   "debug":     timing experiments only (results are wrong or the run is slowed): 1 skip the GEMM epilogue, 2 skip its k loop,
 *                4096 static priority in the GEMM kernel, 8192 + (n << 16) late start of every other CU by n x s_sleep(127),
 *                16384 per-workgroup stamps (cs_debug_trace_read; gemm_big_kernel and the TRACE instantiation of
 *                conv3_lw_kernel, tools/conv_lw_trace.py), 32768 no staging inside the k loop, 65536 activations from the
 *                zero page, 131072 the same k step staged every time; the halo conv kernel has its own bits (csrc/igemm.hip) */
int cs_set_tuning(const char* key, int value);
/* values outside a knob's documented range are rejected (CS_E_ARG).  cs_get_tuning reads a knob; cs_reset_tuning restores every default
 * (the test suite calls it after each test, so a test that fails inside a try / finally cannot leak a knob into the next one). */
int cs_get_tuning(const char* key, int* value);
int cs_reset_tuning(void);
/* Timing experiments only: with cs_set_tuning("debug", 16384) every workgroup of the 256-row GEMM kernel records wall-clock stamps
 * (100 MHz) -- [slot][6] uint64: entry, first stage landed, k loop done, epilogue issued, stores drained, (XCC id << 32 | HW_ID) --
 * which this call copies to host memory (at most 8192 slots; tools/gemm_timeline.py). */
int cs_debug_trace_read(void* dst_host, size_t bytes);
/* the same for attn40_lw_kernel (head dim 40 self-attention): [slot][6] uint64: shader cycles of compute wave 0 over the steady loop, 100 MHz ticks
 * over the same, tiles in it, loader wave 4's cycles in its DMA waits, at the tile barriers, unused (at most 4096 slots; tools/attn_lw_trace.py). */
int cs_debug_attn_trace_read(void* dst_host, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
