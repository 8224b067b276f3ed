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


class FluxKontextEditPipeline:
    """The ``FluxKontextPipeline`` call the FLUX driver uses (edit_ppo/generate_ours.py:87-93):
    ``pipe(image=ref_image, prompt=instruction, num_inference_steps=8, guidance_scale=2.5, generator=torch.manual_seed(0)).images[0]``
    on the HIP components: T5 + CLIP text encoders -> VAE encoder of the reference image -> FMPPOScheduler loop around the DiT ->
    VAE decoder.  Token ids (``input_ids_t5`` / ``input_ids_clip``) or precomputed ``prompt_embeds`` / ``pooled_prompt_embeds`` replace
    ``prompt`` when the tokenizers' asset files are not at hand."""

    def __init__(self, transformer, scheduler, vae, text_encoder=None, text_encoder_2=None, tokenizer=None, tokenizer_2=None, guidance_scale=2.5):
        from .flux import FluxKontextSamplingEngine
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self.text_encoder, self.text_encoder_2, self.tokenizer, self.tokenizer_2 = text_encoder, text_encoder_2, tokenizer, tokenizer_2
        self._engine = FluxKontextSamplingEngine(transformer, scheduler, guidance_scale=guidance_scale)

    vae_scale_factor = 8

    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt=1, max_sequence_length=512,
                      input_ids_t5=None, input_ids_clip=None, **_ignored):
        """edit_ppo/pipeline.py:279-345 -> (prompt_embeds = T5 last_hidden_state [B, 512, 4096], pooled_prompt_embeds = CLIP
        pooler_output [B, 768], text_ids zeros [512, 3]), the 3-tuple edit_ppo/denoise_diffusion.py:36-42 unpacks.  Token ids
        (``input_ids_t5`` / ``input_ids_clip``) replace ``prompt`` when the tokenizers' asset files are not at hand."""
        dev = self.vae.device
        if input_ids_t5 is None or input_ids_clip is None:
            if prompt is None or self.tokenizer is None or self.tokenizer_2 is None:
                raise RuntimeError("a text prompt needs both tokenizers; without them pass token ids or prompt_embeds / pooled_prompt_embeds")
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            input_ids_clip = self.tokenizer(prompts, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                                            return_tensors="pt").input_ids
            input_ids_t5 = self.tokenizer_2(prompts, padding="max_length", max_length=max_sequence_length, truncation=True,
                                            return_tensors="pt").input_ids
        if self.text_encoder is None or self.text_encoder_2 is None:
            raise RuntimeError("text encoders are not attached to this pipeline")
        pooled = self.text_encoder(input_ids_clip.to(dev)).pooler_output
        embeds = self.text_encoder_2(input_ids_t5.to(dev))[0]
        text_ids = torch.zeros(embeds.shape[1], 3, device=dev, dtype=embeds.dtype)
        return embeds, pooled, text_ids

    # ---- the FluxKontextPipeline members edit_ppo/denoise_diffusion.py drives (:45-66, :163-166) --------------------------------
    @property
    def image_processor(self):
        pipe = self

        class _Proc:
            def preprocess(self, image):
                """PIL image(s) -> [B, 3, 8S, 8S] in [-1, 1] (resized to the VAE's configured size); tensors pass through"""
                if isinstance(image, torch.Tensor):
                    return image
                import numpy as np
                from PIL import Image
                S = pipe.vae.config.sample_size
                ims = image if isinstance(image, (list, tuple)) else [image]
                arr = [np.asarray(im.convert("RGB").resize((8 * S, 8 * S), Image.LANCZOS), dtype=np.uint8) for im in ims]
                return torch.stack([torch.from_numpy(a.copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1 for a in arr])

            def postprocess(self, images, output_type="pil"):
                """decoder output in [-1, 1] -> (x / 2 + 0.5).clamp(0, 1) -> PIL images ("pil") or the tensor ("pt")"""
                x = (images.float() / 2 + 0.5).clamp(0, 1)
                if output_type == "pt":
                    return x
                from PIL import Image
                from .evaluation import tensor_to_uint8_hwc
                return [Image.fromarray(tensor_to_uint8_hwc(img)) for img in x]
        return _Proc()

    @staticmethod
    def _pack_latents(latents, batch_size=None, num_channels_latents=None, height=None, width=None):
        from .flux import pack_latents
        return pack_latents(latents)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor=8):
        from .flux import unpack_latents
        return unpack_latents(latents, height, width, vae_scale_factor)

    def prepare_latents(self, image=None, batch_size=1, num_channels_latents=16, height=None, width=None, dtype=None, device=None,
                        generator=None, latents=None):
        """edit_ppo/pipeline.py:625-712: (latents, image_latents = packed VAE-encoded reference image, latent_ids, image_ids);
        ``latents`` (packed noise) is taken as given, like the rollout passes it (edit_ppo/denoise_diffusion.py:52-62)."""
        from .flux import pack_latents, prepare_latent_image_ids
        from .vae import encode_image_latents
        dev = device or self.vae.device
        S = self.vae.config.sample_size
        image_latents = image_ids = None
        if image is not None:
            img = image.to(dev, torch.float16)
            if img.shape[0] == 1 and batch_size > 1:
                img = img.expand(batch_size, -1, -1, -1)
            image_latents = pack_latents(encode_image_latents(self.vae, img.contiguous())).to(dtype or torch.bfloat16)
            image_ids = torch.from_numpy(prepare_latent_image_ids(S // 2, S // 2, first=1.0)).to(dev)
        if latents is None:
            noise = torch.randn(batch_size, self.vae.config.latent_channels, S, S, generator=generator,
                                device=generator.device if generator is not None else dev).to(dev)
            latents = pack_latents(noise).to(dtype or torch.bfloat16)
        latent_ids = torch.from_numpy(prepare_latent_image_ids(S // 2, S // 2)).to(dev)
        return latents, image_latents, latent_ids, image_ids

    @torch.no_grad()
    def __call__(self, image=None, prompt=None, num_inference_steps=28, guidance_scale=None, generator=None, prompt_embeds=None,
                 pooled_prompt_embeds=None, input_ids_t5=None, input_ids_clip=None, output_type="pil", return_dict=True, **_ignored):
        from .flux import pack_latents
        from .vae import encode_image_latents, flux_decode_latents
        dev, S = self.vae.device, self.vae.config.sample_size
        if guidance_scale is not None:
            self._engine.guidance_scale = float(guidance_scale)
        if prompt_embeds is None or pooled_prompt_embeds is None:
            prompt_embeds, pooled_prompt_embeds, _ = self.encode_prompt(prompt, input_ids_t5=input_ids_t5, input_ids_clip=input_ids_clip)
        B = prompt_embeds.shape[0]
        image_latents = None
        if image is not None:
            if not isinstance(image, torch.Tensor):                               # PIL image -> [1, 3, 8S, 8S] in [-1, 1]
                import numpy as np
                from PIL import Image
                arr = np.asarray(image.convert("RGB").resize((8 * S, 8 * S), Image.LANCZOS), dtype=np.uint8)
                image = (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0 * 2 - 1)[None]
            image = image.to(dev, torch.float16)
            if image.shape[0] == 1 and B > 1:
                image = image.expand(B, -1, -1, -1)
            image_latents = pack_latents(encode_image_latents(self.vae, image.contiguous())).to(torch.bfloat16)
        noise = torch.randn(B, self.vae.config.latent_channels, S, S, generator=generator,
                            device=generator.device if generator is not None else dev).to(dev)
        out = self._engine.generate(pack_latents(noise).to(torch.bfloat16), image_latents, prompt_embeds.to(dev, torch.bfloat16),
                                    pooled_prompt_embeds.to(dev, torch.bfloat16), latent_hw=(S // 2, S // 2), num_inference_steps=num_inference_steps)
        if output_type == "latent":
            images = out
        else:
            images = flux_decode_latents(self.vae, out.to(torch.float16), height=8 * S, width=8 * S)
            if output_type == "pil":
                from PIL import Image
                from .evaluation import tensor_to_uint8_hwc
                images = [Image.fromarray(tensor_to_uint8_hwc(img)) for img in images]
            elif output_type != "pt":
                raise ValueError(f"unknown output_type {output_type!r}")
        return types.SimpleNamespace(images=images) if return_dict else (images,)
