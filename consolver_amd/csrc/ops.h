// Internal op-level launchers shared by the UNet executor (unet.cpp) and the
// exported test entry points (include/consolver_hip_ops.h).
#pragma once
#include "common.h"

// ---- kernel-selection knobs (cs_set_tuning / cs_unet_set_tuning) -------------------------------------------------------------------------------------
// One process-wide set (cs_set_tuning) and, for the duration of a host call that runs with per-handle overrides (cs_unet_forward on a handle with
// cs_unet_set_tuning entries), a per-THREAD set: the call builds its own TuneSet (process-wide values + the handle's overrides) on its stack and installs a
// pointer to it in `t_tune`; every launcher reads the knobs through tune().  Nothing writes the process-wide set on behalf of a handle, so another thread's
// forward / op call never sees a handle's overrides, and forwards of different handles do not serialise on a lock.
struct TuneSet {
    int halo = 1;           // conv3_halo_kernel use: 0 never, 1 when it pays, 2 whenever the shape allows (tests), 3 force the 320 / 256-wide form, 4 never the wide form
    int conv_lw = 1;        // 1: stride-1 3x3 convs with N % 160 == 0 or N % 128 == 0 through conv3_lw_kernel (loader waves), 2: the same without its immediate-offset (FAST) path, 3: N % 160 == 0 only, 0: the 8-wave halo kernels
    int gemm_w8 = 1;        // 1: the 256 x 320 linear / 1x1 layers through gemm_w8_kernel (hand-scheduled k loop), 0: gemm_big_kernel
    int gemm_lw = 1;        // 1: the 256 x 160 linear / 1x1 layers (too few 256 x 320 tiles) through gemm_lw_kernel (loader waves), 0: gemm_big_kernel<false, 160>
    int gemm2_w8 = 1;       // 1: gemm2 main launches run the hand-scheduled k loop (W8 instantiation of gemm2_kernel; bit-identical), 0: the compiler-scheduled one
    int attn_lw = 1;        // 1: head dim 40 self-attention (Nq % 256 == 0, Nk % 64 == 0) runs attn40_lw_kernel, 2: the same with 16x16x32 MFMAs for both k steps (bit-identical to attn_kernel), 0: attn_kernel
    int debug = 0;          // experiment / trace bits
    int gemm_gm = -1;       // gemm_big_kernel tile order: -1 auto (bands of 4 tile rows when there are >= 12 tile columns), 0 / 1 row-major, n bands of n
    int gn_fuse = 1;        // 1: GroupNorm statistics of a conv / 1x1 output come from its epilogue (IgemmArgs::gn_stats), 0: always a statistics pass
    int cfg_share = 1;      // 0 runs the CFG dual batch without the shared prefix (A/B, tests)
    int xattn_fused = 1;    // 1: the cross-attention sub-block at C = 320 runs as ONE kernel (xattn.hip); 0: four kernels
    int attn_prio = -1;     // -1 auto (head dim 128 only: -3.4 % on the FLUX shape, +1.5 % at head dim 40), 0 off, 1 on
    int gemm2_prio = 0;
    int biggemm = 1;        // 256-row GEMM tiles: 0 never, 1 when the tile count fills the chip, 2 / 3 force the 320 / 160-wide form
    int attn_qt40 = 4;      // query tiles per wave at head dim 40 (2 | 4)
    // CS_RESIDUAL_F16X2 only: which GEMMs that consume the residual stream DIRECTLY read hi + lo (two passes of the k loop, IgemmArgs::a0_lo) instead of the hi
    // plane: bit 0 the resnet shortcut 1x1 (default: its operand rounding is the largest single stream-level error left), bit 1 proj_out
    int x2_split_a = 1;
    // bit i: the i-th resnet shortcut 1x1 (creation order: down blocks, then up blocks) reads the hi plane ONLY even when x2_split_a bit 0 is set.  Default 0: every
    // shortcut reads hi + lo.  Measured (round 6, same box): 0x78 on the SD1.5 topology = up_blocks.0.resnets.1 / .2 and up_blocks.1.resnets.0 / .1 (the 2560 -> 1280
    // shortcuts at the 8 x 8 / 16 x 16 levels, the cheapest per unit of error by tools/sim_precision_r06.py) takes 0.12 ms off the forward and adds 3.5 % to the
    // tightest gated number (step 0 of the 4-step trajectory: 9.24e-4 -> 9.57e-4, tools/parity_knobs_n4.py) -- margin the gate does not have to spare.
    int x2_sc_skip = 0;
    // 1: the transformer blocks' LayerNorms are folded into the linear layers that consume them (gamma in the packed weights, (mean, rstd) applied in the
    // GEMM epilogue from row statistics the producing layer's epilogue left): no LayerNorm kernel, no normalised copy of the hidden state.  0: ln_kernel + plain GEMMs.
    int ln_fold = 1;
    int xcd_grid = 1;       // 1: weight-heavy layers map the 8 XCDs as a 2-D grid over (row tiles, column tiles) (tile_of, IgemmParams::pn), 0: contiguous runs always
    int epi_fast = 3;       // bit 0: the FAST forms of the fp32-patch epilogue (all loads of a pass in front of its phase 1) in conv3_lw / gemm_w8 / gemm_lw, 0: the generic code;
                            // bit 1 (round 6): also the form "residual + its lo plane, no lo plane out" (the feed-forward's second linear in the split mode)
    int lo8 = 1;            // 1: the transformer blocks' hidden state carries an 8-bit (e5m2) lo plane in the split mode (IgemmArgs::lo8), 0: an fp16 one
    // 1: conv_in runs on the MFMA conv kernel (latents -> NHWC with the 4 channels zero-padded to 64, weights padded alike): coalesced stores, the lo plane and the
    // GroupNorm statistics of its output from the conv epilogue.  0: conv_in_kernel (one thread per pixel, 640-byte strided stores: 111 us at batch 32 = 0.75 TB/s).
    int conv_in_mfma = 1;
    int up_fold = 1;        // the UNet's upsamplers (nearest x2 + 3x3 conv) in the sub-pixel form on pre-summed taps (IgemmArgs::w_up_sub): 0 never, 1 in forwards on one fp16 plane, 2 always
    int head_x2 = 1;        // split stream: the UNet's output head keeps its GroupNorm + SiLU output as hi + lo planes and conv_out multiplies both (0: one fp16 plane)
    int conv_out_mfma = 1;  // 1: the 16 x 16-patch conv_out kernels (UNet 320 -> 4, VAE 128 -> 3) on v_mfma_f32_16x16x32_f16 (conv_out_mfma_kernel), 0: the v_dot2 patch kernel
    int xattn_tile = 64;    // 64: xattn64_kernel, 64-row tiles at two workgroups per CU; 128: xattn_block_kernel (one 160 KB workgroup per CU)
};
extern TuneSet g_tune;                              // process-wide (ops_api.cpp)
extern thread_local const TuneSet* t_tune;          // this thread's override for the duration of a host call, or null
inline const TuneSet& tune() { return t_tune ? *t_tune : g_tune; }
struct TuneScope {                                  // RAII: install / remove a per-thread set
    const TuneSet* prev;
    explicit TuneScope(const TuneSet* t) : prev(t_tune) { t_tune = t; }
    ~TuneScope() { t_tune = prev; }
    TuneScope(const TuneScope&) = delete; TuneScope& operator=(const TuneScope&) = delete;
};
// key -> field of `set` (validated against the knob table: known key, value in range); CS_OK or CS_E_ARG with the error text set.  Writes only `set`.
int tune_apply(TuneSet& set, const char* key, int value);

struct IgemmArgs {
    // activations, NHWC fp16; up to two sources concatenated along channels (skip-concat fusion)
    const f16* a0; const f16* a1; int c0, c1;
    int B, Hi, Wi;          // input spatial size (before the optional nearest x2 upsample)
    int Ho, Wo;             // output spatial size
    int taps;               // 1 = 1x1 conv / linear, 9 = 3x3 conv (pad 1)
    int stride;             // 1 or 2 (3x3 only)
    int upsample;           // 1: nearest x2 upsample fused into the gather
    int pad_after_only;     // stride 2 only: pad (0,1,0,1) instead of 1 on every side (AutoencoderKL encoder downsample)
    int N;                  // output channels (rows of w)
    const f16* w;           // [N][taps*(c0+c1)] K contiguous, tap-major / channel-minor
    const f16* bias;        // [N] or null
    const f16* temb; int temb_stride;   // per-sample [B][>=N] add (time embedding) or null; stride 0 broadcasts
    const f16* res;         // [M][N] residual add or null (may alias out)
    f16* out;               // [M][N] (GEGLU: [M][N/2])
    int geglu;              // rows of w pre-permuted in (value16 | gate16) blocks; out = v * gelu(g)
    float* splitk_ws; size_t splitk_ws_bytes;   // optional fp32 scratch for split-K on small images (may be null)
    // optional: GroupNorm statistics of the OUTPUT, written by the epilogue (or by a statistics pass when the chosen kernel cannot):
    // gn_stats[B][Ho*Wo/64][N/2][2] = (sum, sum of squares) of channel pairs over each 64-row block of a sample; needs Ho*Wo % 64 == 0, no GEGLU
    float* gn_stats;
    // optional: split-fp16 residual stream (CS_RESIDUAL_F16X2).  A stream tensor is two fp16 planes whose sum is the value: `res_lo` is the lo plane of
    // `res` (added in fp32 with it), `out_lo` receives what the fp16 store of `out` dropped: out_lo = f16(v - float(f16(v))).  hi + lo carries 22
    // significant bits, so the adds along the residual stream are fp32-class while every GEMM operand stays a plain fp16 tensor (the hi plane).
    const f16* res_lo; f16* out_lo;
    // lo8 != 0: res_lo / out_lo are 8-BIT planes, one e5m2 byte per element (the fp16 lo value rounded to its top byte: hi + lo8 = 14 significant bits), same [M][N]
    // element layout.  For stream tensors whose lo plane is only ever ADDED (the transformer blocks' hidden state: the to_out / feed-forward epilogues and the fused
    // cross-attention block); a lo plane that is a GEMM operand (a0_lo) stays fp16.
    int lo8;
    // optional (1x1 / linear only): the A operand itself is a split-fp16 stream tensor -- a0_lo / a1_lo are the lo planes of a0 / a1 (same shapes).  The kernel
    // runs the k loop twice over the same weights, hi planes then lo planes, into one accumulator: W (hi + lo) exactly, 2x the layer's MFMA work.  Used where
    // a GEMM consumes the residual stream directly and its operand rounding is a stream-level error (the resnet shortcut 1x1 over [x | skip]).
    const f16* a0_lo; const f16* a1_lo;
    // ---- LayerNorm folded into the linear layer that consumes it (diffusers BasicTransformerBlock: norm1 -> to_q/k/v, norm2 -> attn2.to_q, norm3 -> GEGLU proj).
    // LN(h) W^T + b = rstd (h W'^T - mean s) + b'  with  W' = W diag(gamma) (packed once, fp16),  s[n] = sum_k W'[n][k],  b' = W beta + b  (fp32).  The GEMM reads the
    // RAW hidden state; the per-row (mean, rstd) come from partial sums its PRODUCER left:
    //   producer: row_stats != null -> row_stats[M][G][2] = (sum, sum of squares) of every output row over G column groups, G returned in *row_stats_groups
    //             (written by the epilogue from the fp32 values, or by a pass over the output where the chosen kernel cannot: split-K forms; then G = 1);
    //   consumer: ln_stats (that buffer), ln_groups = G, ln_eps, ln_s, ln_b (fp32 [N]); w = W'; bias is ignored (b' holds it).  The row length is c0.
    float* row_stats; int* row_stats_groups;
    const float* ln_stats; int ln_groups; float ln_eps; const float* ln_s; const float* ln_b;
    // optional (upsample != 0, 3x3): the filter in its sub-pixel form, [4][N][4 c0] from conv_up_fold_pack_host.  When given (and the shape fits: input 8 x 8 or a
    // multiple of 16 x 16, N % 160 == 0, no residual / temb) the layer runs 16 instead of 36 multiplies per input pixel on pre-summed taps -- weights rounded to fp16
    // once more, so the CALLER decides where that is acceptable (the UNet executor: one-plane forwards only, tune().up_fold).
    const f16* w_up_sub;
};
int launch_igemm(const IgemmArgs& a, hipStream_t s);
double igemm_flops(const IgemmArgs& a);
void conv_up_fold_pack_host(const f16* w, int N, int Cin, f16* out);     // w [N][9 Cin] -> out [4][N][4 Cin]: the sub-pixel filters of IgemmArgs::w_up_sub
void ln_fold_pack_host(const f16* w, const f16* bias, const f16* gamma, const f16* beta, int N, int K, f16* w_out, float* s_out, float* b_out);
// (sum, sum of squares) per row of x [M][C] (+ x_lo): stats[M][1][2]; the statistics pass behind IgemmArgs::row_stats and after the fused cross-attention block
int launch_row_stats(const f16* x, const f16* x_lo, int M, int C, float* stats, hipStream_t s, int lo8 = 0);
// *dst += sum over the M rows of mean^2 / (var + eps), rows described by their statistics rs[M][G][2] over C channels (cs_unet_calibrate_ln_fold)
int launch_ln_dc_ratio(const float* rs, int M, int G, int C, float eps, float* dst, hipStream_t s);

struct AttnArgs {
    const f16* q; int q_stride;   // [B, Nq, H*dh] rows of q_stride halfs
    const f16* k; int k_stride;   // [B, Nk, ...]
    const f16* v; int v_stride;
    f16* out; int out_stride;     // [B, Nq, H*dh]
    int B, H, Nq, Nk, dh;
    float scale;
    int dtype;               // CS_F16 (default 0 is treated as f16) or CS_BF16
    int causal;              // 1: key j visible to query i iff j <= i (CLIP text encoder; f16, dh 64, Nq == Nk)
    const float* bias;       // additive score bias [H][Nq][Nk] fp32, already multiplied by log2(e) (T5 relative positions; bf16, dh 64), or null
    void* split_ws; size_t split_ws_bytes;   // optional scratch for the split-KV tail (head dim 128): attention_split_workspace_bytes()
};
int launch_attention(const AttnArgs& a, hipStream_t s);
size_t attention_split_workspace_bytes(int B, int H, int Nq, int Nk, int dh);

// GroupNorm over NHWC [B][HW][C0+C1] (two-source concat), 32 groups.
// stats -> partial[B][S][C] (sum, sumsq); apply normalises (+SiLU) into out [B][HW][C].
struct GroupNormArgs {
    const f16* x0; const f16* x1; int c0, c1;
    int B, HW, groups; float eps; int silu;
    const f16* gamma; const f16* beta;     // [C]
    int splits;                             // 0 -> GN_SPLITS; larger for big images (VAE decoder), power of two
    float* partial;                         // workspace >= B * (splits + 1) * C * 2 floats
    f16* out;
    // optional: partial sums of a source already written by its producer (IgemmArgs::gn_stats layout [B][S][C/2][2]); null -> computed here
    const float* stats0; int S0; const float* stats1; int S1;
    // optional: lo planes of split-fp16 sources (value = x + x_lo; IgemmArgs::out_lo), null = plain fp16 source
    const f16* x0_lo; const f16* x1_lo;
    // optional (split sources): the OUTPUT as two planes too, out_lo = f16(y - float(out)) -- the UNet's output head, whose normalised tensor is conv_out's operand
    f16* out_lo;
};
// statistics of a [B][HW][C] tensor in the producer layout: partial[B][HW/64][C/2][2] (sum, sum of squares of channel pairs per 64-row block)
int launch_gn_stats64(const f16* x, int B, int HW, int C, float* partial, hipStream_t s);
#define GN_SPLITS 16
int launch_group_norm(const GroupNormArgs& a, hipStream_t s);

// x_lo: lo plane of a split-fp16 input (value = x + x_lo) or null
int launch_layer_norm(const f16* x, const f16* gamma, const f16* beta, f16* out, int M, int C, float eps, hipStream_t s, const f16* x_lo = nullptr);

// fused cross-attention sub-block (xattn.hip): out = h + to_out(softmax(scale * to_q(LayerNorm(h)) K^T) V) + bo; C = 320, 8 heads, Nk <= 80
struct XattnArgs {
    const f16* h; f16* out;                 // [M][C]; out may alias h
    const f16* h_lo; f16* out_lo;           // split-fp16 residual stream: lo planes of h / out (both or neither); out_lo may alias h_lo
    int lo8;                                // the lo planes are one e5m2 byte per element (IgemmArgs::lo8)
    float* row_stats;                       // optional: (sum, sum of squares) of every OUTPUT row, [M][1][2] (IgemmArgs::row_stats layout with one group): norm3 folded into the GEGLU GEMM
    const f16* ln_g; const f16* ln_b; float ln_eps;
    const f16* wq; const f16* wo; const f16* bo;
    const f16* kv;                          // [M / HW samples][Nk][2 C]: K | V projections of the text context
    int M, HW, Nk, C, heads; float scale;
};
int launch_xattn_block(const XattnArgs& a, hipStream_t s);

// timestep sinusoid (flip_sin_to_cos, shift 0) -> Linear -> SiLU -> Linear -> SiLU (the SiLU that
// every resnet applies before time_emb_proj) ; out_silu [Bt][D] fp16
int launch_time_embedding(const float* t, int Bt, int C0, int D, const f16* w1, const f16* b1, const f16* w2, const f16* b2,
                          f16* scratch, f16* out_silu, hipStream_t s);
// out[r][n] = sum_k x[r][k] w[n][k] + b[n]  (tiny M; one wave per n)
int launch_rowvec_linear(const f16* x, int R, int K, const f16* w, const f16* b, int N, f16* out, int act_silu, hipStream_t s);

// conv_in: NCHW fp16 latents (n_lat samples, sample b reads b % n_lat) -> NHWC [B][H][W][Cout], 3x3 pad 1
// out_lo: optional lo plane of a split-fp16 output (IgemmArgs::out_lo)
int launch_conv_in(const f16* lat, int n_lat, int B, int Cin, int H, int W, const f16* w /*[Cout][9][Cin]*/, const f16* bias,
                   int Cout, f16* out, hipStream_t s, f16* out_lo = nullptr);
// conv_out: NHWC [B][H][W][Cin] -> NCHW [B][Cout][H][W], 3x3 pad 1 (Cout small)
// out_f32 != 0: `out` is an fp32 tensor (the same NCHW layout)
// x_lo (16 x 16-patch shapes on the MFMA kernel): a lo plane of the operand (value = x + x_lo); the hi plane's product is kept in fp32 (in `out` when out_f32, else in
// scratch32 [B][Cout][H][W]) and a second pass adds the lo plane's product and writes the output's dtype
int launch_conv_out(const f16* x, int B, int Cin, int H, int W, const f16* w /*[Cout][9][Cin]*/, const f16* bias, int Cout,
                    f16* out, hipStream_t s, int out_f32 = 0, const f16* x_lo = nullptr, float* scratch32 = nullptr);

// ---- transformer (FLUX DiT) ops, f16 or bf16 (dtype = CS_F16 / CS_BF16) -------------------------------------
struct Gemm2Args {
    const void* a; long lda; int a_seg_rows, a_seg_stride; long a_row_off;   // A row r -> (r/seg)*stride + r%seg + off  (seg = 0: identity)
    const void* w; const void* bias;      // w: [ceil(N/256)*256][K] (zero padded rows), bias [N]
    int M, N, K;
    void* out; const void* res; long ldc; int c_col_off; int c_seg_rows, c_seg_stride; long c_row_off;
    const float* gate; long gate_stride; int rows_per_sample;   // out = res + gate[m / rows_per_sample][n] * v
    int act;                               // 0 none, 1 GELU(tanh)
    int dtype;
    void* tail_ws; size_t tail_ws_bytes;   // optional scratch for the split-K tail (gemm2_tail_workspace_bytes)
    // optional split residual stream (round 5; FLUX hidden states as hi + lo planes of the model dtype, value = hi + lo): res_lo is added with res in fp32,
    // out_lo receives what the 16-bit store of out dropped.  Same addressing as res / out (ldc, column offset, segment map).  Both or neither; needs res.
    const void* res_lo; void* out_lo;
};
int launch_gemm2(const Gemm2Args& a, hipStream_t s);
size_t gemm2_tail_workspace_bytes(int tiles, int K);   // tiles = 256 x 256 output tiles of the launch (both problems of a pair); 0: never splits
// two independent problems (same dtype) in one launch: b's tiles are appended to a's tile list
int launch_gemm2_pair(const Gemm2Args& a, const Gemm2Args& b, hipStream_t s);
int launch_small_linear(const float* x, int R, int K, const void* w, const void* bias, int N, float* out, int silu_in, int silu_out,
                        int dtype, hipStream_t s);
// y = LayerNorm_noaffine(x) * (1 + scale[b]) + shift[b] ; x,y [M][C] (dtype), scale/shift fp32 rows of stride mod_stride
// x_lo: optional lo plane of a split stream (value = x + x_lo)
// y_lo (with x_lo): the result as two planes too, y_lo = T(o - float(y)) (the output head, whose LayerNorm output is proj_out's A operand)
int launch_ln_modulate(const void* x, void* y, int M, int C, int rows_per_sample, const float* shift, const float* scale, long mod_stride,
                       float eps, int dtype, hipStream_t s, const void* x_lo = nullptr, void* y_lo = nullptr);
// out[i] = float(hi[i]) + float(lo[i]) (two planes of `dtype` -> fp32)
int launch_planes_to_f32(const void* hi, const void* lo, float* out, long n, int dtype, hipStream_t s);
// in place on a fused qkv buffer [S_total rows][ld]: per head RMSNorm(q) * wq, RMSNorm(k) * wk, then RoPE (pairs) with cos/sin [S][dh/2]
int launch_qk_norm_rope(void* qkv, long ld, int rows, int seq, int heads, int dh, int q_col, int k_col, const void* wq, const void* wk,
                        const void* wq_ctx, const void* wk_ctx, int ctx_rows, const float* cosv, const float* sinv, float eps, int dtype,
                        hipStream_t s);
// sinusoidal embedding (flip_sin_to_cos, shift 0) of scalars: out[r][C] fp32
int launch_sinusoid_f32(const float* t, float mult, int R, int C, float* out, hipStream_t s);
int launch_add3_f32(const float* a, const float* b, const float* c, float* out, long n, hipStream_t s);
int launch_cast_f32(const float* x, void* out, long n, int dtype, hipStream_t s);

// ---- VAE decoder helpers -------------------------------------------------------------------------------------
// per-pixel CxC linear on NCHW fp16 (post_quant_conv), with an input scale (1 / scaling_factor)
int launch_pixel_linear_nchw(const f16* x, const f16* w, const f16* b, f16* out, int B, int C, int HW, float in_scale, float in_shift, hipStream_t s);
// row softmax of scores [rows][cols] fp16 in place: softmax(scale * x)
int launch_row_softmax(f16* x, long rows, int cols, float scale, hipStream_t s);
// conv 3x3 to 3 output channels, NHWC in -> NCHW out, optional (y/2+0.5).clamp(0,1)
int launch_conv_out3(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, f16* out, int postprocess, hipStream_t s);

// ---- CLIP text encoder helpers -------------------------------------------------------------------------------
// out[r][:] = tok[ids[r]][:] + pos[r % L][:]   (ids int64, tables fp16)
int launch_embed_tokens(const int64_t* ids, const f16* tok, const f16* pos, f16* out, long rows, int L, int C, int vocab, hipStream_t s);
// x <- x * sigmoid(1.702 x) in place (quick_gelu)
int launch_quick_gelu(f16* x, long n, hipStream_t s);
// latents with up to 64 channels: z = x * scale + shift [-> 1x1 post_quant when w != null], NHWC with channels zero-padded to 64
int launch_latent_to_nhwc64(const f16* x, const f16* w, const f16* b, f16* out, int B, int C, int HW, float in_scale, float in_shift, hipStream_t s);
// conv 3x3 (pad 1) to a few output channels (Cout in {4, 8, 16}), NHWC in -> NCHW out (VAE encoder moments)
int launch_conv_out_small(const f16* x, int B, int Cin, int H, int W, const f16* w, const f16* bias, int Cout, f16* out, hipStream_t s);
// out[b][o][px] = (sum_c w[o][c] x[b][c][px] + bias[o] - out_shift) * out_scale ; w == null: identity on the first Cout channels
int launch_pixel_affine_nchw(const f16* x, int Cin, const f16* w, const f16* bias, int Cout, f16* out, int B, int HW, float out_scale, float out_shift, hipStream_t s);

// ---- T5 encoder ops (f16 / bf16) -------------------------------------------------------------------------------
int launch_rms_norm(const void* x, const void* w, void* y, int M, int C, float eps, int dtype, hipStream_t s);
int launch_gated_mul(const void* a, const void* b, void* out, long n, int dtype, hipStream_t s);
int launch_embed_rows(const int64_t* ids, const void* table, void* out, long rows, int C, int vocab, hipStream_t s);
