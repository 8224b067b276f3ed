"""HIP-backed SD1.5 ``AutoencoderKL`` decoder stand-in and the reference's ``decode_latents``.

``decode_latents(vae, latents, batch_size)`` mirrors utils.py:6-34 (same name, arguments and chunking):
``1 / vae.config.scaling_factor * latents`` -> ``vae.decode(chunk, return_dict=False)[0]`` ->
``(image / 2 + 0.5).clamp(0, 1)`` -> ``torch.cat``.  With the HIP model the scale and the [0, 1] map
are folded into the first / last kernel of each chunk's decode and every chunk is written straight
into its slice of the output (no ``torch.cat`` copy); the values are the same.
"""
import ctypes as C
import types

import torch

from . import _lib as L

SD15_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, sample_size=64, scaling_factor=0.18215, shift_factor=0.0, use_post_quant_conv=True,
                       use_quant_conv=True, with_encoder=False)
# FLUX.1 VAE: 16 latent channels, no quant convs, scaling / shift applied at edit_ppo/pipeline.py:623,1148
FLUX_VAE_CONFIG = dict(latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, sample_size=128, scaling_factor=0.3611, shift_factor=0.1159, use_post_quant_conv=False,
                       use_quant_conv=False, with_encoder=False)


class HipAutoencoderKL:
    is_consolver_hip = True
    dtype = torch.float16

    def __init__(self, config=None, device="cuda:0"):
        cfg = dict(SD15_VAE_CONFIG)
        cfg.update(config or {})
        self.config = types.SimpleNamespace(**cfg)
        self.device = torch.device(device)
        c = L.CsVaeConfig()
        c.latent_channels, c.out_channels = cfg["latent_channels"], cfg["out_channels"]
        for i in range(4):
            c.block_out_channels[i] = cfg["block_out_channels"][i]
        c.layers_per_block, c.norm_num_groups, c.sample_size = cfg["layers_per_block"], cfg["norm_num_groups"], cfg["sample_size"]
        c.use_post_quant_conv = int(bool(cfg.get("use_post_quant_conv", True)))
        c.with_encoder = int(bool(cfg.get("with_encoder", False)))
        c.use_quant_conv = int(bool(cfg.get("use_quant_conv", True)))
        self._ews = None
        h = C.c_void_p()
        L.check(L.lib().cs_vae_create(C.byref(c), C.byref(h)))
        self._h = h
        self._ws = None
        self._ws_batch = 0
        self._finalized = False

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().cs_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def manifest(self):
        lib = L.lib()
        out = []
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(lib.cs_vae_num_weights(self._h)):
            name = lib.cs_vae_weight_name(self._h, i, shape, C.byref(nd)).decode()
            out.append((name, tuple(shape[k] for k in range(nd.value))))
        return out

    def load_state_dict(self, sd, strict=True):
        """Decoder-side tensors of a diffusers AutoencoderKL state dict (encoder keys are ignored)."""
        lib = L.lib()
        want = dict(self.manifest())
        missing = [k for k in want if k not in sd]
        if missing:
            raise KeyError(f"missing {len(missing)} tensors, e.g. {missing[:3]}")
        for name, shape in want.items():
            t = sd[name].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{name}: shape {tuple(t.shape)} != {shape}")
            sh = (C.c_int64 * len(shape))(*shape)
            L.check(lib.cs_vae_set_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), sh, len(shape)))
        torch.cuda.set_device(self.device)
        L.check(lib.cs_vae_finalize(self._h))
        self._finalized = True
        return self

    def flops(self, batch):
        return float(L.lib().cs_vae_flops(self._h, batch))

    def _workspace(self, batch):
        if self._ws is None or batch > self._ws_batch:
            n = int(L.lib().cs_vae_workspace_bytes(self._h, batch))
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_batch = batch
        return self._ws

    def decode_into(self, z, out, in_scale=1.0, in_shift=0.0, postprocess=False):
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        L.require_cuda(z, "z")
        z = z.to(torch.float16).contiguous()
        B, Lc, h, w = z.shape
        if Lc != self.config.latent_channels or h != self.config.sample_size or w != self.config.sample_size:
            raise ValueError(f"latents {tuple(z.shape)} do not match the configured [*, {self.config.latent_channels}, "
                             f"{self.config.sample_size}, {self.config.sample_size}]")
        if out.shape != (B, self.config.out_channels, 8 * h, 8 * w) or out.dtype != torch.float16 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp16 [B, 3, 8h, 8w] tensor")
        if B == 0:
            return out
        ws = self._workspace(B)
        L.check(L.lib().cs_vae_decode(self._h, C.c_void_p(z.data_ptr()), B, float(in_scale), float(in_shift), C.c_void_p(out.data_ptr()),
                                      int(postprocess), C.c_void_p(ws.data_ptr()), ws.numel(),
                                      C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        return out

    def encode_into(self, images, out, out_scale=1.0, out_shift=0.0):
        """images [B,3,8h,8w] in [-1,1] -> out [B,L,h,w] = (posterior mode - out_shift) * out_scale"""
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        if not self.config.with_encoder:
            raise RuntimeError("this HipAutoencoderKL was created without the encoder (config with_encoder=True)")
        L.require_cuda(images, "images")
        x = images.to(torch.float16).contiguous()
        B, C3, H, W = x.shape
        S = self.config.sample_size
        if C3 != self.config.out_channels or H != 8 * S or W != 8 * S:
            raise ValueError(f"images {tuple(x.shape)} do not match the configured [*, 3, {8 * S}, {8 * S}]")
        if out.shape != (B, self.config.latent_channels, S, S) or out.dtype != torch.float16 or not out.is_contiguous():
            raise ValueError("out must be a contiguous fp16 [B, L, h, w] tensor")
        if B == 0:
            return out
        lib = L.lib()
        need = int(lib.cs_vae_encode_workspace_bytes(self._h, B))
        if self._ews is None or self._ews.numel() < need:
            self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)
        L.check(lib.cs_vae_encode(self._h, L.ptr(x), B, float(out_scale), float(out_shift), L.ptr(out), L.ptr(self._ews), self._ews.numel(),
                                  L.stream_ptr(self.device)))
        return out

    def encode(self, images, return_dict=True, **_ignored):
        """-> object with ``.latent_dist.mode()`` (= mean); ``sample()`` is not available: the path takes the argmax
        (edit_ppo/pipeline.py:616,621 ``sample_mode="argmax"``) and the log-variance half is never computed."""
        B = images.shape[0]
        S = self.config.sample_size
        mode = torch.empty(B, self.config.latent_channels, S, S, dtype=torch.float16, device=images.device)
        self.encode_into(images, mode)

        class _Dist:
            def mode(self_inner):
                return mode

            def sample(self_inner, generator=None):
                raise NotImplementedError("only the mode of the posterior is computed (sample_mode='argmax')")
        return types.SimpleNamespace(latent_dist=_Dist())

    def decode(self, z, return_dict=False, **_ignored):
        B, _, h, w = z.shape
        out = torch.empty(B, self.config.out_channels, 8 * h, 8 * w, dtype=torch.float16, device=z.device)
        self.decode_into(z, out)
        if return_dict:
            return types.SimpleNamespace(sample=out)
        return (out,)


def flux_decode_latents(vae, packed_latents, height=1024, width=1024, vae_scale_factor=8, batch_size=8):
    """edit_ppo/utils.py:11-28 (and edit_ppo/pipeline.py:1147-1150): unpack the [B, (h/2)(w/2), 64] token latents to
    [B, 16, h, w], ``latents / scaling_factor + shift_factor``, decode, ``(x / 2 + 0.5).clamp(0, 1)``."""
    from .flux import unpack_latents
    if not getattr(vae, "is_consolver_hip", False):
        raise RuntimeError("flux_decode_latents needs the HIP AutoencoderKL (no CPU fallback in the product path)")
    lat = unpack_latents(packed_latents, height, width, vae_scale_factor)
    N, _, h, w = lat.shape
    out = torch.empty(N, vae.config.out_channels, 8 * h, 8 * w, dtype=torch.float16, device=lat.device)
    for s in range(0, N, batch_size):
        e = min(s + batch_size, N)
        vae.decode_into(lat[s:e], out[s:e], in_scale=1.0 / vae.config.scaling_factor, in_shift=vae.config.shift_factor, postprocess=True)
    return out


def encode_image_latents(vae, image):
    """edit_ppo/pipeline.py:613-623 ``_encode_vae_image``: ``(argmax of vae.encode(image) - shift_factor) * scaling_factor``
    (shift and scale folded into the encoder's last kernel)."""
    if not getattr(vae, "is_consolver_hip", False):
        raise RuntimeError("encode_image_latents needs the HIP AutoencoderKL (no CPU fallback in the product path)")
    B, S = image.shape[0], vae.config.sample_size
    out = torch.empty(B, vae.config.latent_channels, S, S, dtype=torch.float16, device=image.device)
    return vae.encode_into(image, out, out_scale=vae.config.scaling_factor, out_shift=vae.config.shift_factor)


def decode_latents(vae, latents, batch_size=1):
    """utils.py:6-34.  Returns [N, 3, H, W] in [0, 1]."""
    if not getattr(vae, "is_consolver_hip", False):
        raise RuntimeError("decode_latents needs the HIP AutoencoderKL (no CPU fallback in the product path)")
    if batch_size < 1:
        raise ValueError("batch_size must be >= 1")
    N, _, h, w = latents.shape
    out = torch.empty(N, vae.config.out_channels, 8 * h, 8 * w, dtype=torch.float16, device=latents.device)
    for s in range(0, N, batch_size):
        e = min(s + batch_size, N)
        vae.decode_into(latents[s:e], out[s:e], in_scale=1.0 / vae.config.scaling_factor, postprocess=True)
    return out
