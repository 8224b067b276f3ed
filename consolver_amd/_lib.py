"""ctypes binding of libconsolver_hip.so (the C ABI in include/consolver_hip.h).

There is NO fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised.  The product never computes on the CPU.
"""
import ctypes as C
import os

import torch

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CONSOLVER_HIP_LIB") or os.path.join(PKG, "libconsolver_hip.so")   # override: A/B builds in tools/

CS_F32, CS_F16, CS_BF16 = 0, 1, 2
CS_MAX_ORDER = 8
_DT = {torch.float32: CS_F32, torch.float16: CS_F16, torch.bfloat16: CS_BF16}


class CsFactorNet(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("in_dim", C.c_int), ("hidden", C.c_int), ("action_dims", C.c_int), ("num_actions", C.c_int),
                ("input_scale", C.c_float), ("inv_temperature", C.c_float)]


class CsStepArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("eps_text", C.c_void_p), ("eps_uncond", C.c_void_p), ("guidance", C.c_float),
                ("hist", C.c_void_p * CS_MAX_ORDER), ("m", C.c_int), ("order_dim", C.c_int), ("scaler_dim", C.c_int),
                ("actions", C.c_void_p), ("actions_stride", C.c_int), ("B", C.c_int), ("elems", C.c_int64),
                ("io_dtype", C.c_int), ("out_dtype", C.c_int), ("x_out", C.c_void_p), ("eps_out", C.c_void_p),
                ("sqrt_at", C.c_float), ("sqrt_1mat", C.c_float), ("sqrt_ap", C.c_float), ("sqrt_1map", C.c_float),
                ("v_prediction", C.c_int), ("dt", C.c_float), ("x_is_f32", C.c_int), ("x_out_lp", C.c_void_p), ("lp_dtype", C.c_int)]


class CsFluxConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("num_layers", C.c_int), ("num_single_layers", C.c_int), ("num_heads", C.c_int),
                ("head_dim", C.c_int), ("joint_attention_dim", C.c_int), ("pooled_projection_dim", C.c_int),
                ("guidance_embeds", C.c_int), ("axes_dims_rope", C.c_int * 3), ("dtype", C.c_int)]


class CsUNetConfig(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("num_heads", C.c_int), ("cross_attention_dim", C.c_int),
                ("norm_num_groups", C.c_int), ("sample_size", C.c_int), ("ctx_len", C.c_int),
                ("down_has_attn", C.c_int * 4), ("up_has_attn", C.c_int * 4)]


class CsVaeConfig(C.Structure):
    _fields_ = [("latent_channels", C.c_int), ("out_channels", C.c_int), ("block_out_channels", C.c_int * 4),
                ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int), ("sample_size", C.c_int), ("use_post_quant_conv", C.c_int),
                ("with_encoder", C.c_int), ("use_quant_conv", C.c_int)]


class CsClipConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("intermediate_size", C.c_int), ("num_hidden_layers", C.c_int),
                ("num_attention_heads", C.c_int), ("max_position_embeddings", C.c_int), ("layer_norm_eps", C.c_float)]


class CsGemm2Problem(C.Structure):
    _fields_ = [("x", C.c_void_p), ("M", C.c_int), ("K", C.c_int), ("w", C.c_void_p), ("bias", C.c_void_p), ("N", C.c_int),
                ("res", C.c_void_p), ("gate", C.c_void_p), ("gate_stride", C.c_long), ("rows_per_sample", C.c_int), ("act", C.c_int),
                ("out", C.c_void_p), ("ldc", C.c_long), ("col_off", C.c_int)]


# every symbol declared in include/consolver_hip.h: name -> (restype, argtypes)
SYMBOLS = {
    "cs_abi_version": (C.c_int, []),
    "cs_error_string": (C.c_char_p, [C.c_int]),
    "cs_last_error": (C.c_char_p, []),
    "cs_target_arch": (C.c_char_p, []),
    "cs_factor_probs": (C.c_int, [C.POINTER(CsFactorNet), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_cosine_features": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_cosine_features_cfg": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_float, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "cs_sample_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_gather_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_action_probs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_step_masks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_stack_history": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_lms_ddim_step": (C.c_int, [C.POINTER(CsStepArgs), C.c_void_p]),
    "cs_lms_euler_step": (C.c_int, [C.POINTER(CsStepArgs), C.c_void_p]),
    "cs_psnr_workspace_bytes": (C.c_size_t, [C.c_int]),
    "cs_image_psnr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t,
                                C.c_void_p]),
    "cs_ppo_advantages": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_ppo_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p,
                              C.c_void_p]),
    "cs_policy_param_count": (C.c_size_t, [C.POINTER(CsFactorNet)]),
    "cs_policy_workspace_bytes": (C.c_size_t, [C.POINTER(CsFactorNet), C.c_int]),
    "cs_ppo_policy_grads": (C.c_int, [C.POINTER(CsFactorNet), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_clip_grad_norm": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "cs_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_void_p]),
    "cs_clip_create": (C.c_int, [C.POINTER(CsClipConfig), C.POINTER(C.c_void_p)]),
    "cs_clip_destroy": (None, [C.c_void_p]),
    "cs_clip_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "cs_clip_num_weights": (C.c_int, [C.c_void_p]),
    "cs_clip_weight_name": (C.c_char_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "cs_clip_finalize": (C.c_int, [C.c_void_p]),
    "cs_clip_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int]),
    "cs_clip_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "cs_clip_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_vae_create": (C.c_int, [C.POINTER(CsVaeConfig), C.POINTER(C.c_void_p)]),
    "cs_vae_destroy": (None, [C.c_void_p]),
    "cs_vae_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "cs_vae_num_weights": (C.c_int, [C.c_void_p]),
    "cs_vae_weight_name": (C.c_char_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "cs_vae_finalize": (C.c_int, [C.c_void_p]),
    "cs_vae_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "cs_vae_flops": (C.c_double, [C.c_void_p, C.c_int]),
    "cs_vae_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                C.c_void_p]),
    "cs_vae_encode_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "cs_vae_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_unet_create": (C.c_int, [C.POINTER(CsUNetConfig), C.POINTER(C.c_void_p)]),
    "cs_unet_destroy": (None, [C.c_void_p]),
    "cs_unet_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "cs_unet_num_weights": (C.c_int, [C.c_void_p]),
    "cs_unet_weight_name": (C.c_char_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "cs_unet_finalize": (C.c_int, [C.c_void_p]),
    "cs_unet_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int]),
    "cs_unet_flops": (C.c_double, [C.c_void_p, C.c_int]),
    "cs_unet_flops_executed": (C.c_double, [C.c_void_p, C.c_int, C.c_int]),
    "cs_unet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "cs_unet_set_residual_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "cs_unet_get_residual_precision": (C.c_int, [C.c_void_p]),
    "cs_unet_calibrate_ln_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_float, C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_float)]),
    "cs_unet_set_ln_unfold_mask": (C.c_int, [C.c_void_p, C.c_uint]),
    "cs_unet_get_ln_unfold_mask": (C.c_uint, [C.c_void_p]),
    "cs_unet_set_output_dtype": (C.c_int, [C.c_void_p, C.c_int]),
    "cs_unet_get_output_dtype": (C.c_int, [C.c_void_p]),
    "cs_unet_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "cs_unet_profile_entries": (C.c_int, [C.c_void_p]),
    "cs_unet_profile_entry": (C.c_char_p, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "cs_flux_create": (C.c_int, [C.POINTER(CsFluxConfig), C.POINTER(C.c_void_p)]),
    "cs_flux_destroy": (None, [C.c_void_p]),
    "cs_flux_num_weights": (C.c_int, [C.c_void_p]),
    "cs_flux_weight_name": (C.c_char_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "cs_flux_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "cs_flux_finalize": (C.c_int, [C.c_void_p]),
    "cs_flux_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cs_flux_flops": (C.c_double, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cs_flux_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_flux_set_residual_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "cs_flux_get_residual_precision": (C.c_int, [C.c_void_p]),
    "cs_flux_set_output_dtype": (C.c_int, [C.c_void_p, C.c_int]),
    "cs_flux_get_output_dtype": (C.c_int, [C.c_void_p]),
    "cs_flux_forward_joint": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    # include/consolver_hip_ops.h
    "cs_op_gemm2_x2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_ln_modulate_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_int, C.c_void_p]),
    "cs_op_gemm2": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_long,
                              C.c_int, C.c_int, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "cs_op_attention_workspace": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "cs_op_attention_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_gemm2_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_gemm2_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "cs_op_attention_causal": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "cs_op_rms_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "cs_op_gated_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "cs_op_embed_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "cs_op_attention_bias": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "cs_op_attention_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "cs_op_conv2d": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_linear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cs_op_geglu_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_op_attention": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "cs_op_conv_out": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "cs_op_group_norm_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "cs_op_group_norm": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_op_xattn_block": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "cs_op_conv2d_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_linear_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_ln_fold_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_op_linear_ln": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_row_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_op_row_stats_lo8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cs_op_conv_up_fold_pack": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cs_op_conv_up_sub": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_op_ln_dc_ratio": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "cs_op_linear_lo8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cs_op_xattn_block_lo8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_op_group_norm_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_op_layer_norm_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "cs_op_xattn_block_x2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_unet_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "cs_unet_clear_tuning": (C.c_int, [C.c_void_p]),
    "cs_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "cs_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "cs_reset_tuning": (C.c_int, []),
    "cs_op_conv2d_gn": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "cs_op_group_norm_pre": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cs_debug_trace_read": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cs_debug_attn_trace_read": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cs_op_layer_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the bound library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run `python -m consolver_amd.build`). There is no CPU fallback.")
    l = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            f = getattr(l, name)
        except AttributeError as e:
            raise RuntimeError(f"libconsolver_hip.so does not export {name}") from e
        f.restype = res
        f.argtypes = args
    if l.cs_abi_version() != 2:
        raise RuntimeError("libconsolver_hip.so ABI version mismatch")
    _lib = l
    try:
        apply_env_tuning()
    except Exception:
        _lib = None          # a malformed CS_TUNE raises on EVERY call, not only the first (the library must not come up with the remaining knobs silently unapplied)
        raise
    return l


def apply_env_tuning():
    """CS_TUNE="key=value,..." applies kernel-selection knobs (cs_set_tuning): A/B runs of tests and tools without code changes.  Called at load
    time and by reset_tuning()."""
    for kv in os.environ.get("CS_TUNE", "").split(","):
        if "=" in kv:
            k, v = kv.split("=", 1)
            if _lib.cs_set_tuning(k.strip().encode(), int(v)) != 0:
                raise RuntimeError(f"CS_TUNE: {_lib.cs_last_error().decode()}")


def reset_tuning():
    """every knob back to its default (then CS_TUNE re-applied): what tests/conftest.py runs after each test."""
    lib().cs_reset_tuning()
    apply_env_tuning()


def check(code):
    if code != 0:
        l = lib()
        raise RuntimeError(f"libconsolver_hip: {l.cs_error_string(code).decode()}: {l.cs_last_error().decode()}")


def dtype_code(dt):
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}; expected float32/float16/bfloat16")


def require_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{name} must be a CUDA (ROCm) tensor: consolver_amd runs only on the MI355X HIP path; "
            "the CPU restatement lives in oracle/ and is test infrastructure, not a fallback.")
    return t


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
