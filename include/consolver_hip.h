/*
 * consolver_hip.h -- C ABI of libconsolver_hip.so (MI355X / gfx950 only).
 *
 * Drop-in boundary for the ConsistencySolver sampling hot path of
 * G-U-N/consolver.  The reference is pure Python on top of torch/diffusers and
 * has no FFI of its own; the entry points below are what a Python (ctypes),
 * C++ or cgo/JNI host would bind to replace, one for one, the arithmetic inside
 *
 *   scheduler_ppo.py:178-299      PPOScheduler.step            -> cs_lms_ddim_step
 *   scheduler_ppo.py:306-332      PPOScheduler._get_prev_sample   (fused into the above)
 *   scheduler_ppo.py:165-175      set_default_coefficients        (fused into the above)
 *   denoise_ppo.py:96-100         CFG combine u + g (c - u)       (fused into the above)
 *   edit_ppo/scheduler_fmppo.py:306-455  FMPPOScheduler.step   -> cs_lms_euler_step
 *   factor_net_ppo.py:137-157     FactorNetPPO.forward_        -> cs_factor_probs
 *   factor_net_ppo.py:108-130     compute_cosine_similarity    -> cs_cosine_features
 *   factor_net_ppo.py:159-168     sample_action (gather part)  -> cs_sample_actions / cs_gather_actions
 *   factor_net_ppo.py:170-184     get_action_probs             -> cs_action_probs
 *   scheduler_ppo.py:222-232      zero-padded history stack    -> cs_stack_history
 *   denoise_ppo.py:89-94          unet(latents, t, ctx)[0]     -> cs_unet_forward
 *                                 (third-party diffusers UNet2DConditionModel, SD1.5 config)
 *
 * Conventions
 *   - every pointer is a BORROWED DEVICE pointer unless its name ends in _host;
 *   - every call is asynchronous on the hipStream_t passed as `stream`
 *     (void* so that the header needs no HIP include);
 *   - no allocation, no host synchronisation, no global state in the compute
 *     entry points (they are graph-capturable); cs_unet_create/destroy own
 *     device memory for packed weights;
 *   - return value: 0 = CS_OK, negative = error (see below);
 *     cs_last_error() returns a thread-local human readable message.
 */
#ifndef CONSOLVER_HIP_H
#define CONSOLVER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CS_ABI_VERSION 2
#define CS_MAX_ORDER 8        /* order_dim <= 8 */
#define CS_MAX_ACTION_DIMS 16 /* order + scaler + mu - 1 */

enum { CS_OK = 0, CS_E_ARG = -1, CS_E_SHAPE = -2, CS_E_DTYPE = -3, CS_E_HIP = -4, CS_E_STATE = -5, CS_E_UNSUPPORTED = -6 };
enum { CS_F32 = 0, CS_F16 = 1, CS_BF16 = 2 };

int cs_abi_version(void);
const char* cs_error_string(int code);
const char* cs_last_error(void);
/* name of the device the library was built for ("gfx950") */
const char* cs_target_arch(void);

/* ------------------------------------------------------------------------
 * Policy network (FactorNetPPO) : Linear-ReLU-Linear-ReLU-Linear + softmax
 * ---------------------------------------------------------------------- */
typedef struct CsFactorNet {
    const float* w0; const float* b0; /* [hidden, in_dim], [hidden]             */
    const float* w1; const float* b1; /* [hidden, hidden], [hidden]             */
    const float* w2; const float* b2; /* [action_dims*num_actions, hidden], [.] */
    int in_dim;        /* 2 (+ order_dim-1 when use_conv)                        */
    int hidden;        /* <= 1024                                                */
    int action_dims;   /* A <= CS_MAX_ACTION_DIMS                                */
    int num_actions;   /* K <= 1024                                              */
    float input_scale;     /* 1/999 for SD (factor_net_ppo.py:106), 1 for FLUX  */
    float inv_temperature; /* 1 for SD, 100 for FLUX (softmax(logits/0.01))     */
} CsFactorNet;

/* probs[B, A, K] = softmax(MLP(cat(x * input_scale, cos_feat)) * inv_temperature).
 * x: [B, 2] fp32.  cos_feat: [B, in_dim-2] fp32 or NULL when in_dim == 2.
 * x_row_stride = 0 broadcasts one conditioning row to all B samples. */
int cs_factor_probs(const CsFactorNet* net, const float* x, int x_row_stride, const float* cos_feat,
                    int B, float* probs, void* stream);

/* cos(eps[k], eps[0]) for k = 1..order-1 over the flattened sample; slots k >= m
 * (zero padded in the reference) give 0.  hist[k]: [B, elems] newest first.
 * out: [B, order-1] fp32. */
int cs_cosine_features(const void* const* hist_host, int m, int order, int B, int64_t elems,
                       int dtype, float* out, void* stream);

/* The same under classifier-free guidance with the combine fused (denoise_ppo.py:96-100 in
 * front of scheduler_ppo.py:207-240): hist[0] is the TEXT branch, eps_uncond the other one,
 * and the newest history entry is e = eps_uncond + guidance * (hist[0] - eps_uncond) rounded
 * to `dtype` exactly as cs_lms_ddim_step rounds it.  e is formed on the fly for the features
 * and written to eps_out [B, elems] (required), which the caller then passes to the update
 * kernel as an already combined eps.  eps_uncond == NULL: identical to cs_cosine_features. */
int cs_cosine_features_cfg(const void* const* hist_host, int m, int order, int B, int64_t elems,
                           int dtype, const void* eps_uncond, float guidance, void* eps_out,
                           float* out, void* stream);

/* inverse-CDF categorical sampling from probs[B,A,K] with caller supplied
 * uniforms[B,A] in [0,1).  Writes idx[B,A] (int64), actions[B,A] = action_values[a, idx],
 * action_probs[B,A] = probs[b,a,idx].  Any output pointer may be NULL. */
int cs_sample_actions(const float* probs, const float* uniforms, const float* action_values,
                      int B, int A, int K, int64_t* idx, float* actions, float* action_probs, void* stream);

/* same gather with caller supplied indices (replay / torch.multinomial output). */
int cs_gather_actions(const float* probs, const int64_t* idx, const float* action_values,
                      int B, int A, int K, float* actions, float* action_probs, void* stream);

/* PPO re-evaluation (factor_net_ppo.py:170-184): nearest-bin index of each action,
 * selected probability and normalised entropy H/ln K.  All [B, A] fp32. */
int cs_action_probs(const float* probs, const float* actions, const float* action_values,
                    int B, int A, int K, float* selected, float* entropy, void* stream);

/* masks[B, A] = 1 except columns [m-1, order-1) = 0 (scheduler_ppo.py:248-249). */
int cs_step_masks(int B, int A, int m, int order, float* masks, void* stream);

/* out[B, order, elems] = stack(hist newest first) zero padded (scheduler_ppo.py:222-232). */
int cs_stack_history(const void* const* hist_host, int m, int order, int B, int64_t elems,
                     int dtype, void* out, void* stream);

/* ------------------------------------------------------------------------
 * Fused solver update
 * ---------------------------------------------------------------------- */
typedef struct CsStepArgs {
    const void* x;            /* [B, elems] current sample                          */
    const void* eps_text;     /* [B, elems] model output (conditional branch)       */
    const void* eps_uncond;   /* [B, elems] unconditional branch, or NULL (no CFG)  */
    float guidance;           /* eps = u + guidance * (c - u)                       */
    const void* hist[CS_MAX_ORDER]; /* hist[k] = eps_{t-1-k}, k < m-1 (newest first) */
    int m;                    /* history length AFTER pushing the new eps, 1..order */
    int order_dim, scaler_dim;
    const float* actions;     /* [B, actions_stride] sampled grid values, device    */
    int actions_stride;
    int B;
    int64_t elems;            /* elements per sample                                */
    int io_dtype;             /* dtype of x, eps_*, hist, eps_out                   */
    int out_dtype;            /* dtype of x_out                                     */
    void* x_out;              /* [B, elems]                                         */
    void* eps_out;            /* [B, elems] CFG-combined eps (next steps' history)
                                 required when eps_uncond != NULL, else optional     */
    /* DDIM scalars (host fp32, scheduler_ppo.py:309-312): sqrt(a_t), sqrt(1-a_t),
       sqrt(a_prev), sqrt(1-a_prev) */
    float sqrt_at, sqrt_1mat, sqrt_ap, sqrt_1map;
    int v_prediction;
    /* Euler (flow matching): dt = sigma_next - sigma (scheduler_fmppo.py:376) */
    float dt;
    /* nonzero: x is fp32 while eps_* / hist stay io_dtype (scheduler_fmppo.py:354 upcasts the sample
       before the update and rounds only the result); ABI version 2 */
    int x_is_f32;
    /* optional (may be NULL): a second, 16-bit copy of an fp32 result (out_dtype = CS_F32), [B, elems] in lp_dtype (CS_F16 | CS_BF16) -- the view of an fp32 solver
       state that the denoiser reads at the next step, written in the same pass instead of by a cast kernel per step; ABI version 3 */
    void* x_out_lp;
    int lp_dtype;
} CsStepArgs;

int cs_lms_ddim_step(const CsStepArgs* args, void* stream);
int cs_lms_euler_step(const CsStepArgs* args, void* stream);

/* ------------------------------------------------------------------------
 * SD1.5 UNet2DConditionModel forward (denoiser).  See DESIGN.md for the
 * activation layout (NHWC fp16) and the kernels behind it.
 * ---------------------------------------------------------------------- */
typedef struct CsUNetConfig {
    int in_channels, out_channels;     /* 4, 4                        */
    int block_out_channels[4];         /* 320, 640, 1280, 1280        */
    int layers_per_block;              /* 2                           */
    int num_heads;                     /* 8 (attention_head_dim)      */
    int cross_attention_dim;           /* 768                         */
    int norm_num_groups;               /* 32                          */
    int sample_size;                   /* 64 (latent H = W)           */
    int ctx_len;                       /* 77                          */
    int down_has_attn[4];              /* 1,1,1,0                     */
    int up_has_attn[4];                /* 0,1,1,1                     */
} CsUNetConfig;

typedef struct CsUNet CsUNet;

int cs_unet_create(const CsUNetConfig* cfg, CsUNet** out);
void cs_unet_destroy(CsUNet* u);
/* Upload one tensor by its diffusers state-dict name (e.g.
 * "down_blocks.0.resnets.0.conv1.weight").  data_host: fp32 host memory in the
 * PyTorch layout.  The library repacks into its own fp16 device layout. */
int cs_unet_set_weight(CsUNet* u, const char* name, const float* data_host, const int64_t* shape, int ndim);
/* number of tensors the config expects / names, for manifest checks */
int cs_unet_num_weights(const CsUNet* u);
const char* cs_unet_weight_name(const CsUNet* u, int i, int64_t* shape4, int* ndim);
/* all weights present? pack; must be called once before forward */
int cs_unet_finalize(CsUNet* u);
size_t cs_unet_workspace_bytes(const CsUNet* u, int batch);
/* FLOPs of one forward at `batch` samples: the REFERENCE GRAPH's algorithmic count (2*MAC; SURVEY 8(d): batch x 803.27 GFLOP for SD1.5), independent of
 * every execution knob (conv_in is counted at its 4 input channels even when it runs zero-padded on the MFMA conv; the time MLP per sample) */
double cs_unet_flops(const CsUNet* u, int batch);
/* FLOPs actually executed by cs_unet_forward(n_lat, dup): with dup = 2 and one timestep the layers in front of the first
 * cross attention are evaluated once for both CFG halves (same latents, same timestep; bit-identical results), so this is
 * slightly below cs_unet_flops(n_lat * dup), which stays the algorithmic count of the reference graph. */
double cs_unet_flops_executed(const CsUNet* u, int n_lat, int dup);

/* latents: [n_lat, C, H, W] (NCHW, fp16).  The effective batch is
 * n_lat * dup (dup = 2 for the CFG dual batch, sample b reads latent b % n_lat;
 * gen_pretrain/pipeline.py:1054).  timesteps: device fp32 [1] or [batch].
 * ctx: [batch, ctx_len, cross_attention_dim] fp16.  out: [batch, C, H, W] fp16.
 * kv_cache_valid != 0 reuses the cross-attention K/V computed by a previous
 * call with the same ctx (they do not depend on latents or t). */
int cs_unet_forward(CsUNet* u, const void* latents, int n_lat, int dup, const float* timesteps,
                    int n_timesteps, const void* ctx, void* out, void* workspace, size_t workspace_bytes,
                    int kv_cache_valid, void* stream);

/* Storage precision of the UNet's RESIDUAL STREAM (resnet outputs, transformer hidden states, proj_out / down / upsample / conv_in
 * outputs) between kernels.  Every GEMM operand is fp16 in both modes (the reference pipeline's own dtype, gen_ppo.py:193-195).
 *   CS_RESIDUAL_F16X2 (default) two fp16 planes per stream tensor, value = hi + lo (22 significant bits): adds onto the stream are carried in fp32
 *                     (hi + lo in, hi + lo out), norms read hi + lo, the resnet shortcut 1x1 multiplies hi + lo (two passes of its k loop), the other GEMMs
 *                     that consume the stream directly read the hi plane.  This is the mode that meets north_star's 1e-3 latent gate against the fp32
 *                     oracle (8-step latents 0.90e-3, tests/test_parity_e2e_gpu.py); it costs the lo planes' bytes (~9 % of the forward).
 *   CS_RESIDUAL_F16   one fp16 tensor per stream tensor: the arithmetic class of the reference's own fp16 pipeline; the 8-step latents land 1.4e-3
 *                     (relative L2) from an fp32 evaluation of the same graph, because every add onto the stream rounds it.
 * Switching changes cs_unet_workspace_bytes(): query it again. */
#define CS_RESIDUAL_F16   0
#define CS_RESIDUAL_F16X2 1
int cs_unet_set_residual_precision(CsUNet* u, int mode);
/* dtype of cs_unet_forward's `out` tensor: CS_F16 (default: the model dtype, what the reference's `unet(...)[0]` returns, denoise_ppo.py:89-94) or CS_F32: conv_out
 * stores its fp32 accumulator unrounded.  The native engine asks for CS_F32: an fp16 eps tensor carries 2.8e-4 of relative rounding, and the CFG combine
 * u + g (c - u) of two such tensors (rounded again as the history entry) more -- both gone from the 1e-3 latent budget for 128 KiB per image and step. */
int cs_unet_set_output_dtype(CsUNet* u, int dtype);
int cs_unet_get_output_dtype(const CsUNet* u);
/* The folded LayerNorm (knob ln_fold, default on) evaluates LN(h) W^T as rstd (h_fp16 W'^T - mean s) + b': exact algebra, but on hidden states whose rows sit many
 * sigma away from zero the two terms cancel in fp16-rounded operands (error ~ 1.6e-4 x |mean| / sigma on that layer's output; synthetic N(0, 1/fan_in) weights
 * never get there, a trained checkpoint's outlier channels can).  cs_unet_calibrate_ln_fold runs ONE forward on representative inputs (same arguments as
 * cs_unet_forward, which it also performs: `out` is valid afterwards), measures per transformer block the RMS over rows of |mean| / sigma of the hidden state in front
 * of each of its LayerNorms -- from the row statistics the producers leave anyway -- and marks the blocks above `bound` (4.0 is the measured default: there the
 * folded form's error is about twice an fp16 LayerNorm output's rounding) to run UNFOLDED from then on (LayerNorm kernel on hi + lo, plain GEMMs) while every other
 * block keeps the fold.  One device -> host read, at load time; the mask is part of the handle (get / set for checkpoints whose mask is known) and changes
 * cs_unet_workspace_bytes(): query it again. */
int cs_unet_calibrate_ln_fold(CsUNet* u, const void* latents, int n_lat, int dup, const float* timesteps, int n_timesteps, const void* ctx, void* out,
                              void* workspace, size_t workspace_bytes, float bound, void* stream, unsigned* mask_out, float* worst_ratio_out);
int cs_unet_set_ln_unfold_mask(CsUNet* u, unsigned mask);
unsigned cs_unet_get_ln_unfold_mask(const CsUNet* u);
/* Kernel-selection knobs for THIS handle (keys and ranges: cs_set_tuning in consolver_hip_ops.h): cs_unet_forward runs with the process-wide values overridden by the
 * handle's entries for the duration of its host call -- a per-thread knob set, the process-wide one is not written -- so two handles in one process can run
 * different knob sets concurrently.  The workspace size does not depend on them (cs_unet_workspace_bytes covers every variant).  Unknown keys and
 * out-of-range values are rejected.  cs_unet_clear_tuning drops the handle's overrides. */
int cs_unet_set_tuning(CsUNet* u, const char* key, int value);
int cs_unet_clear_tuning(CsUNet* u);
int cs_unet_get_residual_precision(const CsUNet* u);

/* per-kernel-class profile of the last forward recorded with events
 * (enable with cs_unet_set_profiling(u, 1); adds synchronisation -- never in a timed run) */
int cs_unet_set_profiling(CsUNet* u, int on);
int cs_unet_profile_entries(const CsUNet* u);
const char* cs_unet_profile_entry(const CsUNet* u, int i, double* ms, double* flops, double* bytes, int* launches);

/* ------------------------------------------------------------------------
 * FLUX.1-Kontext DiT forward (FluxTransformer2DModel): replaces the third-party call
 *   transformer(hidden_states, timestep / 1000, guidance, pooled_projections, encoder_hidden_states,
 *               txt_ids, img_ids)[0]        (edit_ppo/pipeline.py:1087-1097, edit_ppo/denoise_diffusion.py:135-144)
 * ---------------------------------------------------------------------- */
typedef struct CsFluxConfig {
    int in_channels;           /* 64  (packed 2x2 latents)   */
    int num_layers;            /* 19  double-stream blocks   */
    int num_single_layers;     /* 38  single-stream blocks   */
    int num_heads;             /* 24                         */
    int head_dim;              /* 128                        */
    int joint_attention_dim;   /* 4096 (T5)                  */
    int pooled_projection_dim; /* 768  (CLIP pooled)         */
    int guidance_embeds;       /* 1                          */
    int axes_dims_rope[3];     /* 16, 56, 56                 */
    int dtype;                 /* CS_BF16 (reference) or CS_F16 */
} CsFluxConfig;

typedef struct CsFlux CsFlux;

int cs_flux_create(const CsFluxConfig* cfg, CsFlux** out);
void cs_flux_destroy(CsFlux* f);
int cs_flux_num_weights(const CsFlux* f);
const char* cs_flux_weight_name(const CsFlux* f, int i, int64_t* shape2, int* ndim);
/* data: 16-bit elements in the model dtype, diffusers layout; on_device != 0: device pointer (copied). */
int cs_flux_set_weight(CsFlux* f, const char* name, const void* data, int on_device, const int64_t* shape, int ndim);
int cs_flux_finalize(CsFlux* f);
size_t cs_flux_workspace_bytes(const CsFlux* f, int batch, int txt_len, int img_len);
double cs_flux_flops(const CsFlux* f, int batch, int txt_len, int img_len);
/* hidden_states [B, img_len, in_channels], encoder_hidden_states [B, txt_len, joint_attention_dim] (model dtype);
 * pooled_f32 [B, pooled_projection_dim], timestep [B] (= sigma, i.e. t/1000), guidance [B]: device fp32;
 * rope_cos / rope_sin [txt_len + img_len, head_dim/2] device fp32 (host-computed from txt_ids/img_ids,
 * edit_ppo/pipeline.py:574-585); out [B, img_len, in_channels]. */
int cs_flux_forward(CsFlux* f, const void* hidden_states, int batch, int img_len, const void* encoder_hidden_states, int txt_len,
                    const float* pooled_f32, const float* timestep, const float* guidance, const float* rope_cos, const float* rope_sin,
                    void* out, void* workspace, size_t workspace_bytes, void* stream);
/* The Kontext edit step (edit_ppo/pipeline.py:1080-1098, edit_ppo/denoise_diffusion.py:102,135-145) without the per-step
 * `torch.cat([latents, image_latents], dim=1)` and without the `[:, :latents.size(1)]` slice copy: the image sequence is
 * [latents [B, lat_len, C] | image_latents [B, image_len, C]] read from the two buffers in place, rope tables cover
 * txt_len + lat_len + image_len tokens, workspace = cs_flux_workspace_bytes(f, batch, txt_len, lat_len + image_len),
 * out [B, lat_len, C] = the prediction for the latent rows only.  image_len = 0: same as cs_flux_forward. */
int cs_flux_forward_joint(CsFlux* f, const void* latents, int lat_len, const void* image_latents, int image_len, int batch,
                          const void* encoder_hidden_states, int txt_len, const float* pooled_f32, const float* timestep,
                          const float* guidance, const float* rope_cos, const float* rope_sin, void* out, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Storage of the DiT's hidden-state (residual) stream between kernels, as for the UNet (CS_RESIDUAL_* above): CS_RESIDUAL_F16X2 (default) keeps the image / text /
 * joint hidden states as two planes of the model dtype (value = hi + lo): the 57 + 38 gated-residual epilogues add in fp32 and write both planes, the adaLN
 * LayerNorms read both; every GEMM operand stays a plain 16-bit tensor.  CS_RESIDUAL_F16: one plane -- the reference bf16 pipeline's own arithmetic class.  At
 * full depth (19 + 38 blocks, bf16) the one-plane stream is 1.16e-2 of the forward's 1.2e-2 relative error against an fp32 evaluation (tools/sim_precision_flux.py).
 * Changes the workspace size: query cs_flux_workspace_bytes after switching. */
int cs_flux_set_residual_precision(CsFlux* f, int mode);
int cs_flux_get_residual_precision(const CsFlux* f);
/* dtype of cs_flux_forward(_joint)'s `out`: the model dtype (default: what the reference's `transformer(...)[0]` returns, edit_ppo/pipeline.py:1087-1097) or CS_F32.
 * With the split stream the output head keeps its two tensors -- the modulated LayerNorm output and proj_out's result -- as hi + lo planes (round 6: they were the
 * last one-plane stations of the stream's value, 3.42e-3 -> 2.5e-3 per forward at full depth): the model-dtype `out` is the hi plane, the fp32 `out` the sum of
 * the two.  CS_F32 with CS_RESIDUAL_F16 (one plane) is refused at the forward. */
int cs_flux_set_output_dtype(CsFlux* f, int dtype);
int cs_flux_get_output_dtype(const CsFlux* f);

/* ------------------------------------------------------------------------
 * AutoencoderKL decoder (SD1.5 VAE): latents -> images.  Replaces
 * `vae.decode(latents / vae.config.scaling_factor).sample` and the
 * `(image / 2 + 0.5).clamp(0, 1)` that follows it in decode_latents
 * (utils.py:6-34; gen_pretrain/pipeline.py:589-593).
 * ---------------------------------------------------------------------- */
typedef struct CsVaeConfig {
    int latent_channels, out_channels; /* 4, 3                          */
    int block_out_channels[4];         /* 128, 256, 512, 512            */
    int layers_per_block;              /* 2 (decoder blocks hold 3)     */
    int norm_num_groups;               /* 32                            */
    int sample_size;                   /* 64 (latent H = W); FLUX 128   */
    int use_post_quant_conv;           /* 1 (SD1.5); 0 for the FLUX VAE (16 latent channels) */
    int with_encoder;                  /* 1: also load "encoder.*" [+ "quant_conv.*"] and enable cs_vae_encode */
    int use_quant_conv;                /* 1 (SD1.5); 0 for the FLUX VAE */
} CsVaeConfig;

typedef struct CsVae CsVae;

int cs_vae_create(const CsVaeConfig* cfg, CsVae** out);
void cs_vae_destroy(CsVae* v);
/* tensors by their diffusers AutoencoderKL state-dict names ("post_quant_conv.weight",
 * "decoder.up_blocks.0.resnets.0.conv1.weight", ...); fp32 host memory, PyTorch layout */
int cs_vae_set_weight(CsVae* v, const char* name, const float* data_host, const int64_t* shape, int ndim);
int cs_vae_num_weights(const CsVae* v);
const char* cs_vae_weight_name(const CsVae* v, int i, int64_t* shape4, int* ndim);
int cs_vae_finalize(CsVae* v);
size_t cs_vae_workspace_bytes(const CsVae* v, int batch);
double cs_vae_flops(const CsVae* v, int batch);
/* latents: [batch, 4, h, w] NCHW fp16 (device).  The decoder input is
 * latents * in_scale + in_shift (in_scale = 1 / scaling_factor, utils.py:21).
 * images: [batch, 3, 8h, 8w] NCHW fp16; postprocess != 0 applies
 * (x / 2 + 0.5).clamp(0, 1) (utils.py:29) in the last kernel. */
int cs_vae_decode(CsVae* v, const void* latents, int batch, float in_scale, float in_shift, void* images,
                  int postprocess, void* workspace, size_t workspace_bytes, void* stream);
/* Encoder (edit_ppo/pipeline.py:613-623, `retrieve_latents(vae.encode(image), sample_mode="argmax")`):
 * images [batch, 3, 8h, 8w] NCHW fp16 in [-1, 1] -> latents [batch, L, h, w] fp16 =
 * (mode of the posterior - out_shift) * out_scale   (FLUX: shift_factor, scaling_factor). */
size_t cs_vae_encode_workspace_bytes(const CsVae* v, int batch);
int cs_vae_encode(CsVae* v, const void* images, int batch, float out_scale, float out_shift, void* latents,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * PPO rollout consumer arithmetic (train_ppo.py:352-427)
 * ---------------------------------------------------------------------- */
/* Per-image PSNR of two image batches [B, elems] (fp16 or fp32, values in [0,1]):
 * clamp(10 log10(1 / (mean((pred - target)^2) + 1e-8)), 0, clamp_hi) -> out[B] fp32.
 * clamp_hi = 100 is calculate_image_psnr_reward (edit_ppo/reward_model.py:484-509);
 * clamp_hi <= 0 means "clamp(min=0) only", the PSNR tail of the depth reward (:404-422). */
size_t cs_psnr_workspace_bytes(int batch);
int cs_image_psnr(const void* pred, const void* target, int B, int64_t elems, int dtype, float clamp_hi,
                  float* out, void* workspace, size_t workspace_bytes, void* stream);
/* advantages[(b, j), a] = (r[b] - mean(r)) / (std_unbiased(r) + 1e-8) * 10 * masks[(b, j), a],
 * j over the recorded steps (n - 1) (train_ppo.py:376-390).  All fp32. */
int cs_ppo_advantages(const float* rewards, int B, int recorded_steps, int A, const float* masks,
                      float* out, void* stream);
/* value of the clipped surrogate + entropy bonus (train_ppo.py:408-421); inputs [R, A] fp32,
 * loss: one fp32 on the device. */
int cs_ppo_loss(const float* curr_probs, const float* old_probs, const float* entropy,
                const float* advantages, int R, int A, float clip_range, float entropy_coef,
                float* loss, void* stream);

/* ------------------------------------------------------------------------
 * PPO policy update (train_ppo.py:404-437): gradients of the loss with respect to the
 * factor net, clip_grad_norm_, AdamW.  All fp32 device memory.
 * ---------------------------------------------------------------------- */
/* number of trainable parameters = floats in the packed gradient vector, laid out in state-dict
 * order: mlp.0.weight [H, in], mlp.0.bias, mlp.2.weight [H, H], mlp.2.bias, mlp.4.weight [A K, H], mlp.4.bias */
size_t cs_policy_param_count(const CsFactorNet* net);
size_t cs_policy_workspace_bytes(const CsFactorNet* net, int R);
/* x [R, 2] (timestep pairs, un-normalised), cos_feat [R, in_dim - 2] or NULL, actions / old_probs / advantages [R, A],
 * action_values [A, K] -> grads (packed, see above) = d loss / d params, loss (one float, may be NULL) with
 * loss = -mean min(adv rho, adv clip(rho, 1 +- clip_range)) - entropy_coef mean(H / ln K) (train_ppo.py:408-427). */
int cs_ppo_policy_grads(const CsFactorNet* net, const float* x, const float* cos_feat, const float* actions,
                        const float* action_values, const float* old_probs, const float* advantages, int R,
                        float clip_range, float entropy_coef, float* grads, float* loss,
                        void* workspace, size_t workspace_bytes, void* stream);
/* torch.nn.utils.clip_grad_norm_ over one packed vector (train_ppo.py:432-435); total_norm may be NULL */
int cs_clip_grad_norm(float* grads, int64_t n, float max_norm, float* total_norm, void* stream);
/* one torch.optim.AdamW step on one tensor (train_ppo.py:223-229,436); step counts from 1 */
int cs_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------------
 * CLIP text encoder (prompt front-end): replaces `text_encoder(input_ids)[0]`
 * (denoise_ppo.py:25-50; third-party transformers CLIPTextModel).
 * ---------------------------------------------------------------------- */
typedef struct CsClipConfig {
    int vocab_size;                /* 49408 */
    int hidden_size;               /* 768   */
    int intermediate_size;         /* 3072  */
    int num_hidden_layers;         /* 12    */
    int num_attention_heads;       /* 12 (head dim 64) */
    int max_position_embeddings;   /* 77    */
    float layer_norm_eps;          /* 1e-5  */
} CsClipConfig;

typedef struct CsClip CsClip;

int cs_clip_create(const CsClipConfig* cfg, CsClip** out);
void cs_clip_destroy(CsClip* c);
/* tensors by their transformers CLIPTextModel names without the "text_model." prefix
 * ("embeddings.token_embedding.weight", "encoder.layers.0.self_attn.q_proj.weight", ...); fp32 host memory */
int cs_clip_set_weight(CsClip* c, const char* name, const float* data_host, const int64_t* shape, int ndim);
int cs_clip_num_weights(const CsClip* c);
const char* cs_clip_weight_name(const CsClip* c, int i, int64_t* shape4, int* ndim);
int cs_clip_finalize(CsClip* c);
size_t cs_clip_workspace_bytes(const CsClip* c, int batch, int seq_len);
double cs_clip_flops(const CsClip* c, int batch, int seq_len);
/* input_ids: [batch, seq_len] int64 (device).  out: last_hidden_state [batch, seq_len, hidden] fp16
 * (after final_layer_norm), i.e. element [0] of the reference's text_encoder(...) output. */
int cs_clip_encode(CsClip* c, const int64_t* input_ids, int batch, int seq_len, void* out,
                   void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONSOLVER_HIP_H */
