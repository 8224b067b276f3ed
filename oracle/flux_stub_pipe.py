"""Closed-form stand-in for the ``FluxKontextPipeline`` object that
``edit_ppo/denoise_diffusion.denoise_diffusion(scheduler, pipe, ...)`` drives (edit_ppo/denoise_diffusion.py:11-176).

TEST INFRASTRUCTURE ONLY.  It exists so that the IMPORTED reference function can run in the build container
(``oracle/make_golden.py flux_rollout`` -> tests/golden/flux_rollout.npz) and so that the product's mirror
(``consolver_amd/rollout_flux.denoise_diffusion``) can be driven through exactly the same object on the GPU:
every third-party network of the real pipeline (T5/CLIP, VAE, DiT) is replaced by a small closed-form function of
its inputs, so the only arithmetic under test is the rollout loop + ``FMPPOScheduler.step`` + the record layout.
Plain torch ops, device-agnostic (CPU for the fixture generator, CUDA in the -m gpu test).  Nothing here is a
restatement of reference code; the surface (method names / arguments) is the one the reference function calls.
"""
import types

import numpy as np
import torch


def velocity_np(h, sigma, guidance, pooled, enc, img_ids):
    """the stub DiT as numpy fp32 (used by the oracle-side rollout): h [B, S, C] -> v [B, S, C]"""
    h = np.asarray(h, np.float32)
    B = h.shape[0]
    m = h.mean(axis=1, keepdims=True, dtype=np.float32)
    ids = np.asarray(img_ids, np.float32)
    pos = (0.01 * np.sin(ids[:, 1] + 2.0 * ids[:, 2] + 3.0 * ids[:, 0])).astype(np.float32)[None, :, None]
    s = (0.1 * np.asarray(sigma, np.float32).reshape(B, 1, 1) + 0.01 * np.asarray(guidance, np.float32).reshape(B, 1, 1)
         + 0.05 * np.asarray(pooled, np.float32).mean(axis=1, dtype=np.float32).reshape(B, 1, 1)
         + 0.05 * np.asarray(enc, np.float32).mean(axis=(1, 2), dtype=np.float32).reshape(B, 1, 1)).astype(np.float32)
    return (0.9 * np.tanh(h) + 0.3 * m + pos + s).astype(np.float32)


class _Transformer:
    def __init__(self):
        self.config = types.SimpleNamespace(in_channels=64, guidance_embeds=True)
        self.calls = []          # (S, timestep) per call, for the record checks

    def __call__(self, hidden_states=None, timestep=None, guidance=None, pooled_projections=None, encoder_hidden_states=None,
                 txt_ids=None, img_ids=None, return_dict=False, **kw):
        h = hidden_states.float()
        B, S, _ = h.shape
        assert img_ids.shape == (S, 3) and txt_ids.shape == (encoder_hidden_states.shape[1], 3)
        self.calls.append((S, timestep.float().cpu().numpy().copy()))
        ids = img_ids.float()
        pos = (0.01 * torch.sin(ids[:, 1] + 2.0 * ids[:, 2] + 3.0 * ids[:, 0]))[None, :, None]
        s = (0.1 * timestep.float().view(B, 1, 1) + 0.01 * guidance.float().view(B, 1, 1)
             + 0.05 * pooled_projections.float().mean(dim=1).view(B, 1, 1)
             + 0.05 * encoder_hidden_states.float().mean(dim=(1, 2)).view(B, 1, 1))
        v = 0.9 * torch.tanh(h) + 0.3 * h.mean(dim=1, keepdim=True) + pos + s
        return (v.to(hidden_states.dtype),)


class _ImageProcessor:
    def preprocess(self, image):
        return image if isinstance(image, torch.Tensor) else torch.stack(list(image))

    def postprocess(self, images, output_type="pil"):
        return images               # the stub keeps tensors (the reference returns PIL images here)


class _Vae:
    def __init__(self):
        self.config = types.SimpleNamespace(scaling_factor=0.3611, shift_factor=0.1159)

    @staticmethod
    def _mix(n_out, n_in, device):
        i = torch.arange(n_out, dtype=torch.float32, device=device)[:, None]
        j = torch.arange(n_in, dtype=torch.float32, device=device)[None, :]
        return torch.sin(0.7 * i + 1.3 * j + 0.2) / n_in ** 0.5

    def encode_mode(self, image):
        """[B, 3, 8h, 8w] -> [B, 16, h, w]: 8x8 block mean + a fixed channel mix"""
        B, C, H, W = image.shape
        x = image.float().view(B, C, H // 8, 8, W // 8, 8).mean(dim=(3, 5))
        return torch.einsum("oc,bchw->bohw", self._mix(16, C, image.device), x)

    def decode(self, latents, return_dict=False):
        x = torch.einsum("oc,bchw->bohw", self._mix(3, latents.shape[1], latents.device), latents.float())
        x = torch.tanh(x).repeat_interleave(8, dim=2).repeat_interleave(8, dim=3)
        return (x.to(latents.dtype),)


class StubKontextPipe:
    vae_scale_factor = 8

    def __init__(self, txt_len=16, joint_dim=32, pooled_dim=24):
        self.transformer = _Transformer()
        self.image_processor = _ImageProcessor()
        self.vae = _Vae()
        self.txt_len, self.joint_dim, self.pooled_dim = txt_len, joint_dim, pooled_dim

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1, max_sequence_length=512, **kw):
        B = len(prompt)
        code = torch.tensor([[float((len(p) * 7 + 3 * j) % 13) for j in range(self.txt_len)] for p in prompt], device=device)
        f = torch.arange(1, self.joint_dim + 1, dtype=torch.float32, device=device)
        embeds = torch.sin(code[..., None] * f * 0.37).to(torch.bfloat16)
        pooled = torch.cos(code[:, :1] * torch.arange(1, self.pooled_dim + 1, dtype=torch.float32, device=device) * 0.21).to(torch.bfloat16)
        text_ids = torch.zeros(self.txt_len, 3, device=device, dtype=torch.bfloat16)
        assert embeds.shape[0] == B
        return embeds, pooled, text_ids

    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        # the reference passes height // 8 of ITS hard-coded 1024 (edit_ppo/denoise_diffusion.py:51); the stub packs by the
        # tensor's own shape so that fixtures can be small
        B, C, H, W = latents.shape
        x = latents.view(B, C, H // 2, 2, W // 2, 2).permute(0, 2, 4, 1, 3, 5)
        return x.reshape(B, (H // 2) * (W // 2), C * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        B, L, ch = latents.shape
        side = int(round(L ** 0.5))
        x = latents.view(B, side, side, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(B, ch // 4, 2 * side, 2 * side)

    @staticmethod
    def _ids(h, w, first, device, dtype):
        ids = torch.zeros(h, w, 3, device=device, dtype=torch.float32)
        ids[..., 0] = first
        ids[..., 1] += torch.arange(h, device=device, dtype=torch.float32)[:, None]
        ids[..., 2] += torch.arange(w, device=device, dtype=torch.float32)[None, :]
        return ids.reshape(h * w, 3).to(dtype)

    def prepare_latents(self, image=None, batch_size=1, num_channels_latents=16, height=None, width=None, dtype=None, device=None,
                        generator=None, latents=None):
        enc = (self.vae.encode_mode(image) - self.vae.config.shift_factor) * self.vae.config.scaling_factor
        enc = enc.to(dtype)
        image_latents = self._pack_latents(enc, batch_size, num_channels_latents, enc.shape[2], enc.shape[3])
        side = int(round(latents.shape[1] ** 0.5))
        latent_ids = self._ids(side, side, 0.0, device, dtype)
        image_ids = self._ids(enc.shape[2] // 2, enc.shape[3] // 2, 1.0, device, dtype)
        return latents, image_latents, latent_ids, image_ids
