"""``ConsistencySolverPipeline``: the call surface the reference's drivers use on ``StableDiffusionPipeline``
(gen_ppo.py:289-312: ``pipeline(prompt=..., num_inference_steps=..., generator=..., guidance_scale=..., height=..., width=...).images``;
readme.md:65-77), assembled from the HIP components: CLIP text encoder -> 8-step PPOScheduler loop around the UNet (CFG dual
batch, fused solver update) -> VAE decoder.  ``pipe.scheduler``, ``pipe.unet``, ``pipe.vae``, ``pipe.text_encoder`` and
``pipe.tokenizer`` are plain attributes like on the diffusers object, so ``pipe.scheduler.factor_net.load_state_dict(...)`` works.

Without a tokenizer (its vocabulary files are assets) pass ``prompt_embeds`` / ``negative_prompt_embeds`` instead of ``prompt``
(diffusers' pipeline accepts the same keyword arguments).
"""
import types

import torch

from .engine import SDSamplingEngine
from .text_encoder import encode_prompts


class ConsistencySolverPipeline:
    def __init__(self, unet, scheduler, vae, text_encoder=None, tokenizer=None):
        self.unet, self.scheduler, self.vae, self.text_encoder, self.tokenizer = unet, scheduler, vae, text_encoder, tokenizer
        self.vae_scale_factor = 8
        self._engine = None

    @property
    def device(self):
        return self.unet.device

    def to(self, *_args, **_kw):           # the HIP modules are created on their device; kept for call compatibility
        return self

    def enable_vae_slicing(self):          # gen_ppo.py:199: decode one image at a time
        self._decode_batch = 1

    def _eng(self, guidance_scale):
        if self._engine is None or self._engine.guidance_scale != float(guidance_scale) or self._engine.scheduler is not self.scheduler:
            self._engine = SDSamplingEngine(self.unet, self.scheduler, guidance_scale=guidance_scale, vae=self.vae)
        return self._engine

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_inference_steps=50, guidance_scale=7.5, negative_prompt=None,
                 generator=None, latents=None, prompt_embeds=None, negative_prompt_embeds=None, output_type="pil", return_dict=True,
                 **_ignored):
        S = self.unet.config["sample_size"]
        if (height is not None and height != 8 * S) or (width is not None and width != 8 * S):
            raise ValueError(f"this pipeline instance is built for {8 * S} x {8 * S} images (UNet sample_size {S})")
        dev = self.device
        if prompt_embeds is None:
            if prompt is None:
                raise ValueError("pass `prompt` or `prompt_embeds`")
            if self.text_encoder is None or self.tokenizer is None:
                raise RuntimeError("a text prompt needs `text_encoder` and `tokenizer`; without them pass prompt_embeds / negative_prompt_embeds")
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            neg = "" if negative_prompt is None else negative_prompt
            prompt_embeds, negative_prompt_embeds = encode_prompts(self.text_encoder, self.tokenizer, prompts, dev, negative_prompt=neg)
        B = prompt_embeds.shape[0]
        if guidance_scale > 1.0 and negative_prompt_embeds is None:
            raise ValueError("guidance_scale > 1 needs negative_prompt_embeds (or a tokenizer / text_encoder to encode \"\")")
        if latents is None:
            shape = (B, self.unet.config["in_channels"], S, S)
            gen_dev = generator.device if generator is not None else dev
            latents = torch.randn(shape, generator=generator, device=gen_dev, dtype=torch.float16).to(dev)      # pipeline.prepare_latents
        eng = self._eng(guidance_scale)
        if output_type == "latent":
            out = eng.generate(prompt_embeds.to(dev), negative_prompt_embeds.to(dev) if negative_prompt_embeds is not None else None,
                               latents=latents, num_inference_steps=num_inference_steps).clone()
        else:
            out = eng.generate(prompt_embeds.to(dev), negative_prompt_embeds.to(dev) if negative_prompt_embeds is not None else None,
                               latents=latents, num_inference_steps=num_inference_steps, output_type="pt",
                               decode_batch_size=getattr(self, "_decode_batch", None))
            if output_type == "pil":
                from PIL import Image
                from .evaluation import tensor_to_uint8_hwc
                out = [Image.fromarray(tensor_to_uint8_hwc(img)) for img in out]
            elif output_type == "np":
                out = out.float().permute(0, 2, 3, 1).cpu().numpy()
            elif output_type != "pt":
                raise ValueError(f"unknown output_type {output_type!r}")
        if not return_dict:
            return (out, None)
        return types.SimpleNamespace(images=out, nsfw_content_detected=None)
