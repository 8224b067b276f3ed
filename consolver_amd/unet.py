"""HIP-backed SD1.5 ``UNet2DConditionModel`` stand-in.

Call-compatible with the denoiser the reference drives
(``unet(latent_in, t, encoder_hidden_states=ctx, return_dict=False)[0]``, denoise_ppo.py:89-94;
gen_pretrain/pipeline.py:1058-1066): NCHW fp16 latents in, NCHW fp16 eps out.  Weights are loaded
by their diffusers state-dict names (``load_state_dict``) and packed once by the library.

Extension used by the native sampling engine: ``dup=2`` runs the CFG dual batch without
``torch.cat([latents] * 2)`` (sample b reads latent ``b % B``).

``residual``: storage of the residual stream between kernels (include/consolver_hip.h, CS_RESIDUAL_*):
``"f16x2"`` (default: hi + lo fp16 planes, 22 significant bits, the adds along the residual stream are
fp32-class; the mode that meets the 1e-3 latent gate against an fp32 evaluation of the graph) or ``"f16"``
(one fp16 plane: the reference pipeline's own arithmetic class, 1.4e-3, ~9 % faster).  ``"residual_fp32"``
is accepted as an alias of ``"f16x2"``.  Every GEMM operand is fp16 in both modes.
"""
import ctypes as C

import torch

from . import _lib as L

SD15_CONFIG = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                   num_heads=8, cross_attention_dim=768, norm_num_groups=32, sample_size=64, ctx_len=77,
                   down_has_attn=(1, 1, 1, 0), up_has_attn=(0, 1, 1, 1))


class HipUNet2DConditionModel:
    is_consolver_hip = True
    dtype = torch.float16

    RESIDUAL_MODES = {"f16": 0, "f16x2": 1, "residual_fp32": 1}

    def __init__(self, config=None, device="cuda:0", residual="f16x2"):
        cfg = dict(SD15_CONFIG)
        cfg.update(config or {})
        self.config = cfg
        self.device = torch.device(device)
        c = L.CsUNetConfig()
        c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
        for i in range(4):
            c.block_out_channels[i] = cfg["block_out_channels"][i]
            c.down_has_attn[i] = cfg["down_has_attn"][i]
            c.up_has_attn[i] = cfg["up_has_attn"][i]
        c.layers_per_block, c.num_heads = cfg["layers_per_block"], cfg["num_heads"]
        c.cross_attention_dim, c.norm_num_groups = cfg["cross_attention_dim"], cfg["norm_num_groups"]
        c.sample_size, c.ctx_len = cfg["sample_size"], cfg["ctx_len"]
        h = C.c_void_p()
        L.check(L.lib().cs_unet_create(C.byref(c), C.byref(h)))
        self._h = h
        self._ws = None
        self._ws_batch = 0
        self._kv_ctx_ref = None      # strong reference to the conditioning tensor the K/V cache was built from
        self._kv_ctx_version = -1
        self._kv_batch = -1
        self._finalized = False
        self._t_buf = None
        self._out_code = L.dtype_code(torch.float16)      # the handle's output dtype (cs_unet_set_output_dtype), switched on demand by __call__
        self.residual = "f16x2"                       # the library's default (CS_RESIDUAL_F16X2)
        if residual != "f16x2":
            self.set_residual_precision(residual)

    def set_residual_precision(self, mode):
        """"f16" | "f16x2" (alias "residual_fp32"); drops the workspace (its size depends on the mode) and the cached cross-attention K/V."""
        if mode not in self.RESIDUAL_MODES:
            raise ValueError(f"residual must be one of {sorted(self.RESIDUAL_MODES)}, got {mode!r}")
        L.check(L.lib().cs_unet_set_residual_precision(self._h, self.RESIDUAL_MODES[mode]))
        self.residual = "f16x2" if self.RESIDUAL_MODES[mode] else "f16"
        self._ws, self._ws_batch = None, 0
        self.invalidate_kv()
        return self

    def set_residual_precision_keep(self, mode):
        """switch the mode WITHOUT dropping the workspace / K/V cache (both are sized for and shared by the two modes, `_workspace`)"""
        if mode not in self.RESIDUAL_MODES:
            raise ValueError(f"residual must be one of {sorted(self.RESIDUAL_MODES)}, got {mode!r}")
        L.check(L.lib().cs_unet_set_residual_precision(self._h, self.RESIDUAL_MODES[mode]))
        self.residual = "f16x2" if self.RESIDUAL_MODES[mode] else "f16"
        return self

    def set_tuning(self, key, value):
        """kernel-selection knob for THIS model only (cs_unet_set_tuning): applied around each of its forwards, the process-wide ``ops.set_tuning`` state is untouched"""
        L.check(L.lib().cs_unet_set_tuning(self._h, key.encode(), int(value)))
        return self

    def clear_tuning(self):
        L.check(L.lib().cs_unet_clear_tuning(self._h))
        return self

    def calibrate_ln_fold(self, sample, timestep, encoder_hidden_states, dup=1, bound=4.0):
        """cs_unet_calibrate_ln_fold: one forward on representative inputs; transformer blocks whose hidden states sit more than ``bound`` sigma (RMS over rows) away
        from zero in front of a LayerNorm run their LayerNorms UNFOLDED from then on (the folded form cancels in fp16-rounded operands there).  Returns
        ``dict(mask=..., worst_ratio=...)``; call once after ``load_state_dict`` of a real checkpoint (synthetic weights: mask 0).  The workspace is re-sized."""
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        L.require_cuda(sample, "sample")
        ctx = L.require_cuda(encoder_hidden_states, "encoder_hidden_states").to(torch.float16).contiguous()
        sample = sample.to(torch.float16).contiguous()
        n_lat = sample.shape[0]
        B = n_lat * dup
        if ctx.shape[0] != B:
            raise ValueError(f"encoder_hidden_states batch {ctx.shape[0]} != {B}")
        t = (timestep.to(torch.float32).reshape(-1) if isinstance(timestep, torch.Tensor) else torch.tensor([float(timestep)], dtype=torch.float32)).to(sample.device)
        # the workspace covers every mask only after the mask is known: size it for the all-unfolded graph as well (the larger of the two), then drop it
        lib = L.lib()
        L.check(lib.cs_unet_set_ln_unfold_mask(self._h, 0))
        ws = torch.empty(int(lib.cs_unet_workspace_bytes(self._h, B)), dtype=torch.uint8, device=self.device)
        out = torch.empty(B, self.config["out_channels"], sample.shape[2], sample.shape[3], dtype=torch.float16, device=sample.device)
        if self._out_code != L.dtype_code(torch.float16):
            L.check(lib.cs_unet_set_output_dtype(self._h, L.dtype_code(torch.float16)))
            self._out_code = L.dtype_code(torch.float16)
        mask, worst = C.c_uint(0), C.c_float(0.0)
        L.check(lib.cs_unet_calibrate_ln_fold(self._h, L.ptr(sample), n_lat, dup, L.ptr(t), t.numel(), L.ptr(ctx), L.ptr(out), L.ptr(ws), ws.numel(), float(bound),
                                              L.stream_ptr(sample.device), C.byref(mask), C.byref(worst)))
        self._ws, self._ws_batch = None, 0
        self.invalidate_kv()
        return dict(mask=int(mask.value), worst_ratio=float(worst.value))

    @property
    def ln_unfold_mask(self):
        return int(L.lib().cs_unet_get_ln_unfold_mask(self._h))

    @ln_unfold_mask.setter
    def ln_unfold_mask(self, mask):
        L.check(L.lib().cs_unet_set_ln_unfold_mask(self._h, int(mask)))
        self._ws, self._ws_batch = None, 0
        self.invalidate_kv()

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().cs_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def manifest(self):
        """[(name, shape)] in diffusers state-dict naming."""
        lib = L.lib()
        out = []
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(lib.cs_unet_num_weights(self._h)):
            name = lib.cs_unet_weight_name(self._h, i, shape, C.byref(nd)).decode()
            out.append((name, tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, sd, strict=True):
        lib = L.lib()
        want = dict(self.manifest())
        missing = [k for k in want if k not in sd]
        if missing and strict:
            raise KeyError(f"missing {len(missing)} tensors, e.g. {missing[:3]}")
        for name, shape in want.items():
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            sh = (C.c_int64 * len(shape))(*shape)
            L.check(lib.cs_unet_set_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), sh, len(shape)))
        torch.cuda.set_device(self.device)
        L.check(lib.cs_unet_finalize(self._h))
        self._finalized = True
        return self

    # ------------------------------------------------------------------ run
    def flops(self, batch):
        return float(L.lib().cs_unet_flops(self._h, batch))

    def flops_executed(self, n_lat, dup=1):
        """FLOPs the library actually executes for ``dup`` copies of ``n_lat`` latents (CFG shared prefix, see consolver_hip.h)."""
        return float(L.lib().cs_unet_flops_executed(self._h, n_lat, dup))

    def _workspace(self, batch):
        if self._ws is None or batch > self._ws_batch:
            # sized for BOTH residual-stream modes (the larger arena; the K/V cache, GroupNorm and split-K regions in front of it do not depend on the mode), so that a
            # forward can pick its mode per call (`residual=`: the sampling engine's precision schedule) on one workspace and one K/V cache
            lib = L.lib()
            cur = self.RESIDUAL_MODES[self.residual]
            n = 0
            for mode in (0, 1):
                L.check(lib.cs_unet_set_residual_precision(self._h, mode))
                n = max(n, int(lib.cs_unet_workspace_bytes(self._h, batch)))
            L.check(lib.cs_unet_set_residual_precision(self._h, cur))
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
            self.invalidate_kv()
        return self._ws

    def invalidate_kv(self):
        """drop the cached cross-attention K/V (the next forward recomputes them from its conditioning)."""
        self._kv_ctx_ref = None
        self._kv_ctx_version = -1
        self._kv_batch = -1

    def set_profiling(self, on):
        L.check(L.lib().cs_unet_set_profiling(self._h, int(on)))

    def profile(self):
        lib = L.lib()
        out = {}
        ms, fl, by, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        for i in range(lib.cs_unet_profile_entries(self._h)):
            name = lib.cs_unet_profile_entry(self._h, i, C.byref(ms), C.byref(fl), C.byref(by), C.byref(n)).decode()
            out[name] = dict(ms=ms.value, flops=fl.value, bytes=by.value, launches=n.value)
        return out

    def __call__(self, sample, timestep, encoder_hidden_states=None, return_dict=False, dup=1, reuse_kv=None, out=None, out_dtype=None,
                 residual=None, **_ignored):
        """``out_dtype`` (or the dtype of ``out``): torch.float16 (default: the model dtype, what the reference's UNet returns) or torch.float32 -- conv_out stores its
        fp32 accumulator unrounded (cs_unet_set_output_dtype): the native engine's choice.  ``residual`` ("f16x2" | "f16"): the residual-stream mode of THIS forward (the
        handle's mode afterwards); workspace and cached cross-attention K/V are shared by the two modes."""
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        L.require_cuda(sample, "sample")
        ctx = L.require_cuda(encoder_hidden_states, "encoder_hidden_states")
        sample = sample.to(torch.float16).contiguous()
        ctx = ctx.to(torch.float16).contiguous()
        n_lat = sample.shape[0]
        B = n_lat * dup
        if ctx.shape[0] != B:
            raise ValueError(f"encoder_hidden_states batch {ctx.shape[0]} != {B}")
        if isinstance(timestep, torch.Tensor) and timestep.is_cuda:
            t = timestep.to(torch.float32).reshape(-1)
        else:
            # host scalar -> tiny pinned-free H2D through a cached device buffer (no sync)
            t = torch.tensor([float(timestep)], dtype=torch.float32).to(sample.device, non_blocking=True)
        if t.numel() not in (1, B):
            raise ValueError("timestep must be a scalar or one value per sample")
        ws = self._workspace(B)
        if residual is not None and residual != self.residual:
            if residual not in self.RESIDUAL_MODES:
                raise ValueError(f"residual must be one of {sorted(self.RESIDUAL_MODES)}, got {residual!r}")
            L.check(L.lib().cs_unet_set_residual_precision(self._h, self.RESIDUAL_MODES[residual]))
            self.residual = "f16x2" if self.RESIDUAL_MODES[residual] else "f16"
        # Cross-attention K/V of the prompt are cached across the steps of one generation.  reuse_kv=True/False
        # is the caller's explicit statement (the sampling engine and the rollout pass ``i > 0``).  With
        # reuse_kv=None the cache is reused only when the SAME tensor object, unmodified, is passed again: the
        # cache holds a strong reference to it, so its storage cannot be recycled for another prompt batch
        # (a (data_ptr, _version) key alone would match a fresh ``torch.cat`` that the caching allocator placed
        # at the freed address of the previous one).
        if reuse_kv is None:
            kv_valid = (encoder_hidden_states is self._kv_ctx_ref and ctx is encoder_hidden_states
                        and ctx._version == self._kv_ctx_version and self._kv_batch == B)
        else:
            kv_valid = bool(reuse_kv) and self._kv_batch == B
            if reuse_kv and not kv_valid:
                raise RuntimeError("reuse_kv=True but no K/V cache exists for this batch size")
        if out is None:
            out = torch.empty(B, self.config["out_channels"], sample.shape[2], sample.shape[3], dtype=out_dtype or torch.float16,
                              device=sample.device)
        if out.dtype not in (torch.float16, torch.float32) or not out.is_contiguous() or (out_dtype is not None and out.dtype != out_dtype):
            raise ValueError("out must be a contiguous fp16 or fp32 tensor (of out_dtype when both are given)")
        want = L.dtype_code(out.dtype)
        if want != self._out_code:
            L.check(L.lib().cs_unet_set_output_dtype(self._h, want))
            self._out_code = want
        L.check(L.lib().cs_unet_forward(self._h, L.ptr(sample), n_lat, dup, L.ptr(t), t.numel(), L.ptr(ctx), L.ptr(out),
                                        L.ptr(ws), ws.numel(), int(kv_valid), L.stream_ptr(sample.device)))
        self._kv_ctx_ref, self._kv_ctx_version, self._kv_batch = encoder_hidden_states, ctx._version, B
        if return_dict:
            return {"sample": out}
        return (out,)
