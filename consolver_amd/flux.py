"""HIP-backed FLUX.1-Kontext DiT (``FluxTransformer2DModel`` stand-in) + the latent layout helpers and
the edit sampling loop of the FLUX side of the hot path.

Mirrors what the reference drives (edit_ppo/pipeline.py:1074-1140, edit_ppo/denoise_diffusion.py:96-160):

    x_in = cat([latents, image_latents], dim=1)
    v    = transformer(hidden_states=x_in, timestep=t/1000, guidance=g, pooled_projections=pooled,
                       encoder_hidden_states=t5, txt_ids=txt_ids, img_ids=ids, return_dict=False)[0][:, :L]
    latents = scheduler.step(v, t, latents, return_dict=False)[0]

Weights load by their diffusers state-dict names.  There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .tables import calculate_shift

FLUX_KONTEXT_CONFIG = dict(in_channels=64, num_layers=19, num_single_layers=38, num_heads=24, head_dim=128,
                           joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                           axes_dims_rope=(16, 56, 56), dtype=torch.bfloat16)


class _Config(dict):
    """dict with attribute access (``pipe.transformer.config.in_channels``, edit_ppo/denoise_diffusion.py:50,69)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


# ---- layout helpers (pure index shuffles, edit_ppo/pipeline.py:574-611) ------------------------------------------
def prepare_latent_image_ids(height, width, first=0.0):
    ids = np.zeros((height, width, 3), np.float32)
    ids[..., 0] = first
    ids[..., 1] += np.arange(height, dtype=np.float32)[:, None]
    ids[..., 2] += np.arange(width, dtype=np.float32)[None, :]
    return ids.reshape(height * width, 3)


def pack_latents(latents):
    """[B, C, H, W] -> [B, (H/2)(W/2), 4C] (2x2 patches)."""
    B, Cc, H, W = latents.shape
    x = latents.view(B, Cc, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (H // 2) * (W // 2), Cc * 4)


def unpack_latents(latents, height, width, vae_scale_factor=8):
    B, _, ch = latents.shape
    h = 2 * (int(height) // (vae_scale_factor * 2))
    w = 2 * (int(width) // (vae_scale_factor * 2))
    x = latents.view(B, h // 2, w // 2, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, h, w)


def rope_tables(ids, axes_dims, theta=10000.0):
    """cos/sin [S, head_dim/2] fp32 of the 3-axis rotary embedding (float64 angles like diffusers)."""
    cos, sin = [], []
    ids = np.asarray(ids, np.float64)
    for i, d in enumerate(axes_dims):
        freqs = 1.0 / (theta ** (np.arange(0, d, 2, dtype=np.float64)[: d // 2] / d))
        ang = np.outer(ids[:, i], freqs)
        cos.append(np.cos(ang)); sin.append(np.sin(ang))
    return np.concatenate(cos, 1).astype(np.float32), np.concatenate(sin, 1).astype(np.float32)


class HipFluxTransformer2DModel:
    is_consolver_hip = True

    RESIDUAL_MODES = {"plain": 0, "bf16": 0, "f16": 0, "split": 1, "bf16x2": 1, "f16x2": 1}

    def __init__(self, config=None, device="cuda:0", residual="split"):
        """``residual``: storage of the hidden-state stream between kernels (include/consolver_hip.h, cs_flux_set_residual_precision): ``"split"`` (default: hi + lo
        planes of the model dtype, fp32-class adds along the 57-block stream) or ``"plain"`` (one plane: the reference bf16 pipeline's own arithmetic class)."""
        cfg = _Config(FLUX_KONTEXT_CONFIG)
        cfg.update(config or {})
        self.config = cfg
        self.dtype = cfg["dtype"]
        self.device = torch.device(device)
        c = L.CsFluxConfig()
        c.in_channels, c.num_layers, c.num_single_layers = cfg["in_channels"], cfg["num_layers"], cfg["num_single_layers"]
        c.num_heads, c.head_dim = cfg["num_heads"], cfg["head_dim"]
        c.joint_attention_dim, c.pooled_projection_dim = cfg["joint_attention_dim"], cfg["pooled_projection_dim"]
        c.guidance_embeds = int(cfg["guidance_embeds"])
        for i in range(3):
            c.axes_dims_rope[i] = cfg["axes_dims_rope"][i]
        c.dtype = L.dtype_code(self.dtype)
        h = C.c_void_p()
        L.check(L.lib().cs_flux_create(C.byref(c), C.byref(h)))
        self._h = h
        self._finalized = False
        self._ws = None
        self._ws_key = None
        self._rope = {}
        self.residual = "split"
        self._out_f32 = False
        if residual != "split":
            self.set_residual_precision(residual)

    def set_residual_precision(self, mode):
        if mode not in self.RESIDUAL_MODES:
            raise ValueError(f"residual must be one of {sorted(self.RESIDUAL_MODES)}, got {mode!r}")
        L.check(L.lib().cs_flux_set_residual_precision(self._h, self.RESIDUAL_MODES[mode]))
        self.residual = "split" if self.RESIDUAL_MODES[mode] else "plain"
        self._ws, self._ws_key = None, None          # the workspace size depends on the mode
        return self

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().cs_flux_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def manifest(self):
        lib = L.lib()
        out, shape, nd = [], (C.c_int64 * 2)(), C.c_int()
        for i in range(lib.cs_flux_num_weights(self._h)):
            name = lib.cs_flux_weight_name(self._h, i, shape, C.byref(nd)).decode()
            out.append((name, tuple(shape[k] for k in range(nd.value))))
        return out

    def set_weight(self, name, tensor):
        """tensor: torch tensor (cpu or cuda) of the manifest shape; converted to the model dtype."""
        t = tensor.detach().to(self.dtype).contiguous()
        sh = (C.c_int64 * t.dim())(*t.shape)
        L.check(L.lib().cs_flux_set_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), int(t.is_cuda), sh, t.dim()))

    def load_state_dict(self, sd, strict=True):
        want = dict(self.manifest())
        missing = [k for k in want if k not in sd]
        if missing and strict:
            raise KeyError(f"missing {len(missing)} tensors, e.g. {missing[:3]}")
        torch.cuda.set_device(self.device)
        for name, shape in want.items():
            if tuple(sd[name].shape) != shape:
                raise ValueError(f"{name}: shape {tuple(sd[name].shape)} != {shape}")
            self.set_weight(name, sd[name])
        return self.finalize()

    def finalize(self):
        torch.cuda.set_device(self.device)
        L.check(L.lib().cs_flux_finalize(self._h))
        self._finalized = True
        return self

    def flops(self, batch, txt_len, img_len):
        return float(L.lib().cs_flux_flops(self._h, batch, txt_len, img_len))

    def _rope_dev(self, txt_ids, img_ids):
        # same id objects as the previous call (the edit loop passes them unchanged every step): no D2H copy, no sync
        last = getattr(self, "_rope_last", None)
        ver = lambda t: t._version if torch.is_tensor(t) else 0                     # (in-place edits of either id tensor invalidate the cache)
        if last is not None and last[0] is txt_ids and last[1] is img_ids and last[2] == (ver(txt_ids), ver(img_ids)):
            return last[3]
        out = self._rope_from_values(txt_ids, img_ids)
        self._rope_last = (txt_ids, img_ids, (ver(txt_ids), ver(img_ids)), out)
        return out

    def _rope_from_values(self, txt_ids, img_ids):
        ids = np.concatenate([np.asarray(txt_ids.detach().float().cpu() if torch.is_tensor(txt_ids) else txt_ids, np.float32),
                              np.asarray(img_ids.detach().float().cpu() if torch.is_tensor(img_ids) else img_ids, np.float32)], 0)
        key = ids.tobytes()
        if key not in self._rope:
            cos, sin = rope_tables(ids, self.config["axes_dims_rope"])
            self._rope = {key: (torch.from_numpy(cos).to(self.device), torch.from_numpy(sin).to(self.device))}
        return self._rope[key]

    def __call__(self, hidden_states, timestep, guidance=None, pooled_projections=None, encoder_hidden_states=None,
                 txt_ids=None, img_ids=None, joint_attention_kwargs=None, return_dict=False, out=None, image_latents=None,
                 out_dtype=None, **_ignored):
        """the reference's call (edit_ppo/pipeline.py:1087-1097): ``hidden_states`` [B, I, 64] (the caller's
        ``cat([latents, image_latents], 1)``) -> [B, I, 64].

        Extension used by the native edit loop: ``image_latents=`` [B, I2, 64] passes the Kontext reference-image
        tokens as a second buffer -- the joint sequence [hidden_states | image_latents] is read in place (no per-step
        ``torch.cat``) and the result holds the rows of ``hidden_states`` only ([B, I, 64], i.e. the reference's
        ``noise_pred[:, :latents.size(1)]`` without the slice copy); ``img_ids`` cover I + I2 tokens.
        ``out_dtype=torch.float32`` (split stream only; cs_flux_set_output_dtype): the prediction as the unrounded sum of the output head's two planes instead of
        their hi plane in the model dtype."""
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        L.require_cuda(hidden_states, "hidden_states")
        hs = hidden_states.to(self.dtype).contiguous()
        enc = L.require_cuda(encoder_hidden_states, "encoder_hidden_states").to(self.dtype).contiguous()
        B, I, _ = hs.shape
        T = enc.shape[1]
        il, I2 = None, 0
        if image_latents is not None:
            il = L.require_cuda(image_latents, "image_latents").to(self.dtype).contiguous()
            if il.shape[0] != B or il.shape[2] != hs.shape[2]:
                raise ValueError("image_latents must be [B, I2, in_channels]")
            I2 = il.shape[1]
        pooled = L.require_cuda(pooled_projections, "pooled_projections").to(torch.float32).contiguous()
        t = timestep.to(device=hs.device, dtype=torch.float32).reshape(-1) if torch.is_tensor(timestep) else \
            torch.full((B,), float(timestep), dtype=torch.float32, device=hs.device)
        if t.numel() == 1:
            t = t.expand(B).contiguous()
        g = None
        if self.config["guidance_embeds"]:
            if guidance is None:
                raise ValueError("guidance is required (guidance_embeds=True)")
            g = guidance.to(device=hs.device, dtype=torch.float32).reshape(-1)
            if g.numel() == 1:
                g = g.expand(B).contiguous()
        cos, sin = self._rope_dev(txt_ids, img_ids)
        if cos.shape[0] != T + I + I2:
            raise ValueError(f"ids cover {cos.shape[0]} tokens, expected {T + I + I2}")
        want_f32 = out_dtype == torch.float32 or (out_dtype is None and out is not None and out.dtype == torch.float32)
        if out_dtype not in (None, torch.float32, self.dtype):
            raise ValueError(f"out_dtype must be the model dtype {self.dtype} or torch.float32, got {out_dtype}")
        if want_f32 != self._out_f32:
            L.check(L.lib().cs_flux_set_output_dtype(self._h, L.CS_F32 if want_f32 else L.dtype_code(self.dtype)))
            self._out_f32 = want_f32
        key = (B, T, I + I2, want_f32)
        if self._ws_key != key:
            n = int(L.lib().cs_flux_workspace_bytes(self._h, B, T, I + I2))
            self._ws = torch.empty(n, dtype=torch.uint8, device=hs.device)
            self._ws_key = key
        odt = torch.float32 if want_f32 else self.dtype
        if out is None:
            out = torch.empty(B, I, self.config["in_channels"], dtype=odt, device=hs.device)
        elif out.dtype != odt or tuple(out.shape) != (B, I, self.config["in_channels"]) or not out.is_contiguous():
            raise ValueError(f"out must be a contiguous {odt} tensor of shape {(B, I, self.config['in_channels'])}")
        if il is None:
            L.check(L.lib().cs_flux_forward(self._h, L.ptr(hs), B, I, L.ptr(enc), T, L.ptr(pooled), L.ptr(t), L.ptr(g), L.ptr(cos),
                                            L.ptr(sin), L.ptr(out), L.ptr(self._ws), self._ws.numel(), L.stream_ptr(hs.device)))
        else:
            L.check(L.lib().cs_flux_forward_joint(self._h, L.ptr(hs), I, L.ptr(il), I2, B, L.ptr(enc), T, L.ptr(pooled), L.ptr(t),
                                                  L.ptr(g), L.ptr(cos), L.ptr(sin), L.ptr(out), L.ptr(self._ws), self._ws.numel(),
                                                  L.stream_ptr(hs.device)))
        if return_dict:
            return {"sample": out}
        return (out,)


class FluxKontextSamplingEngine:
    """The FLUX edit loop (edit_ppo/pipeline.py:1009-1140 minus encoders / VAE): sigma schedule with the
    resolution-dependent shift, [latents | image_latents] joint input, FMPPOScheduler update."""

    def __init__(self, transformer, scheduler, guidance_scale=2.5):
        self.transformer, self.scheduler, self.guidance_scale = transformer, scheduler, float(guidance_scale)

    @torch.no_grad()
    def generate(self, latents, image_latents, prompt_embeds, pooled_prompt_embeds, latent_hw, image_hw=None,
                 num_inference_steps=8, record=False):
        """latents / image_latents: packed [B, L, 64]; latent_hw = (H/2, W/2) of the packed grid.
        record=True additionally returns the PPO trajectory records of edit_ppo/denoise_diffusion.py:152-172
        (conds{x, epsilon}, probs, actions, masks; steps i > 0 only)."""
        dev = latents.device
        B, Lq, _ = latents.shape
        sch = self.scheduler
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
        mu = calculate_shift(Lq, sch.config.get("base_image_seq_len", 256), sch.config.get("max_image_seq_len", 4096),
                             sch.config.get("base_shift", 0.5), sch.config.get("max_shift", 1.15))
        sch.set_timesteps(sigmas=sigmas, mu=mu, device=dev)
        sch.set_begin_index(0)
        ids = prepare_latent_image_ids(*latent_hw)
        if image_latents is not None:
            ids = np.concatenate([ids, prepare_latent_image_ids(*(image_hw or latent_hw), first=1.0)], 0)
        txt_ids = np.zeros((prompt_embeds.shape[1], 3), np.float32)
        guidance = torch.full([B], self.guidance_scale, device=dev, dtype=torch.float32)
        # timestep / 1000 in the model dtype like the reference (`t.expand(B).to(dtype)` then `/ 1000`, pipeline.py:1084-1089)
        sig = (sch.timesteps.to(latents.dtype) / 1000).to(torch.float32)
        x = latents
        rec = dict(x=[], epsilon=[], probs=[], actions=[], masks=[])
        prev_record = sch.record_conds
        sch.record_conds = bool(record)
        try:
            for i, t in enumerate(sch.timesteps):
                # [latents | image_latents] is read in place by the embedder; v holds the latent rows only
                v = self.transformer(x, sig[i:i + 1].expand(B), guidance=guidance, pooled_projections=pooled_prompt_embeds,
                                     encoder_hidden_states=prompt_embeds, txt_ids=txt_ids, img_ids=ids, image_latents=image_latents)[0]
                x, actions, probs, conds, masks = sch.step(v, t, x, return_dict=False)
                if record and i > 0:
                    rec["x"].append(conds["x"].unsqueeze(1)); rec["epsilon"].append(conds["epsilon"].unsqueeze(1))
                    rec["probs"].append(probs.unsqueeze(1)); rec["actions"].append(actions.unsqueeze(1)); rec["masks"].append(masks.unsqueeze(1))
        finally:
            sch.record_conds = prev_record
        if not record:
            return x
        cat = {k: torch.cat(v, dim=1) for k, v in rec.items()}
        return x, {"x": cat["x"], "epsilon": cat["epsilon"]}, cat["probs"], cat["actions"], cat["masks"]
