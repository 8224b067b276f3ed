"""FLUX-Kontext PPO rollout -- mirror of ``edit_ppo/denoise_diffusion.denoise_diffusion``
(edit_ppo/denoise_diffusion.py:11-176): same arguments, same return value

    (latents_output, pred_images, conds_{x, epsilon}, probs_, actions_, masks_)      (:170-174)
    (latents_output, pred_images)                            with use_naive_scheduler   (:175-176)

with trajectory records for steps ``i > 0`` only (:152-157).  ``pipe`` is driven through the same surface the
reference uses (``encode_prompt``, ``image_processor``, ``_pack_latents``, ``prepare_latents``, ``transformer``,
``_unpack_latents``, ``vae``): ``consolver_amd.pipeline.FluxKontextEditPipeline`` provides it on the HIP
components, and any object with that surface works.

Differences in mechanism, not in results:
* when ``pipe.transformer`` is the HIP DiT the joint input ``cat([latents, image_latents], 1)`` (:102) is never
  materialised (the embedder reads both buffers, ``image_latents=``) and the prediction comes back for the latent
  rows only (the reference slices ``noise_pred[:, :L]``, :145);
* height / width come from the noise tensor (``8 * noise.shape[-2:]``) instead of the hard-coded 1024 (:33-34) -- the same
  value for the reference's 128 x 128 latents;
* ``text`` may also be a dict ``{"prompt_embeds", "pooled_prompt_embeds"[, "text_ids"]}`` of precomputed embeddings
  (tokenizer assets are not part of this repo);
* ``gradient_checkpointing`` is rejected: the rollout is inference-only (``@torch.no_grad()``, :10).
"""
import numpy as np
import torch

from . import _lib as L
from .tables import calculate_shift


@torch.no_grad()
def denoise_diffusion(scheduler, pipe, noise, text, image, cfg=2.5, num_inference_steps=28, gradient_checkpointing=False,
                      use_naive_scheduler=False):
    if gradient_checkpointing:
        raise NotImplementedError("inference-only rollout: the denoiser is frozen")
    L.require_cuda(noise, "noise")
    device, dtype = noise.device, noise.dtype
    guidance_scale = cfg
    height, width = noise.shape[-2] * pipe.vae_scale_factor, noise.shape[-1] * pipe.vae_scale_factor

    if isinstance(text, dict):
        prompt_embeds, pooled_prompt_embeds = text["prompt_embeds"].to(device), text["pooled_prompt_embeds"].to(device)
        text_ids = text.get("text_ids")
        if text_ids is None:
            text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device, dtype=dtype)
        batch_size = prompt_embeds.shape[0]
    else:
        if isinstance(text, str):
            text = [text]
        batch_size = len(text)
        prompt_embeds, pooled_prompt_embeds, text_ids = pipe.encode_prompt(prompt=text, prompt_2=None, device=device,
                                                                           num_images_per_prompt=1, max_sequence_length=512)

    image = pipe.image_processor.preprocess(image).to(device=device, dtype=dtype)

    num_channels_latents = pipe.transformer.config.in_channels // 4
    noise = pipe._pack_latents(noise, batch_size, num_channels_latents, height // 8, width // 8)
    initial_latents, image_latents, latent_ids, image_ids = pipe.prepare_latents(
        image=image, batch_size=batch_size, num_channels_latents=num_channels_latents, height=height, width=width, dtype=dtype,
        device=device, generator=None, latents=noise)
    if image_ids is not None:
        latent_ids = torch.cat([latent_ids, image_ids], dim=0)

    guidance = None
    if pipe.transformer.config.guidance_embeds:
        guidance = torch.full([batch_size], guidance_scale, device=device, dtype=torch.float32)

    # sigma schedule with the resolution-dependent shift (:74-91; retrieve_timesteps == scheduler.set_timesteps(sigmas=, mu=))
    sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
    image_seq_len = initial_latents.shape[1]
    cfgget = scheduler.config.get
    mu = calculate_shift(image_seq_len, cfgget("base_image_seq_len", 256), cfgget("max_image_seq_len", 4096),
                         cfgget("base_shift", 0.5), cfgget("max_shift", 1.15))
    scheduler.set_timesteps(sigmas=sigmas, mu=mu, device=device)
    timesteps = scheduler.timesteps

    dit = pipe.transformer
    native = getattr(dit, "is_consolver_hip", False)
    # trajectory records of the steps i > 0 (:152-157), one list per field
    rec = {"x": [], "epsilon": [], "probs": [], "actions": [], "masks": []}
    record_prev = getattr(scheduler, "record_conds", None)
    if record_prev is not None and not use_naive_scheduler:
        scheduler.record_conds = True           # conds['epsilon'] is materialised only for the rollout
    # t.expand(B).to(dtype) / 1000 for every step at once: the per-step scalar stays a device slice (no host sync)
    ts_model = timesteps.to(dtype) / 1000
    latents = initial_latents
    n_lat_tokens = latents.size(1)
    common = dict(guidance=guidance, pooled_projections=pooled_prompt_embeds, encoder_hidden_states=prompt_embeds, txt_ids=text_ids,
                  img_ids=latent_ids, return_dict=False)
    try:
        for i, t in enumerate(timesteps):
            t_in = ts_model[i].expand(batch_size)
            if native and image_latents is not None:
                # [latents | image_latents] read in place; the prediction comes back for the latent rows only
                velocity = dit(hidden_states=latents, timestep=t_in, image_latents=image_latents, **common)[0]
            elif image_latents is not None:
                velocity = dit(hidden_states=torch.cat([latents, image_latents], dim=1), timestep=t_in, **common)[0][:, :n_lat_tokens]
            else:
                velocity = dit(hidden_states=latents, timestep=t_in, **common)[0]
            stepped = scheduler.step(velocity, t, latents, return_dict=False)
            latents = stepped[0]
            if not use_naive_scheduler and i > 0:
                _, actions, probs, conds, masks = stepped
                for key, val in (("x", conds["x"]), ("epsilon", conds["epsilon"]), ("probs", probs), ("actions", actions), ("masks", masks)):
                    rec[key].append(val.unsqueeze(1))
    finally:
        if record_prev is not None:
            scheduler.record_conds = record_prev
    latents_output = latents

    # decode (:163-166)
    lat = pipe._unpack_latents(latents, height, width, pipe.vae_scale_factor)
    lat = (lat / pipe.vae.config.scaling_factor) + pipe.vae.config.shift_factor
    pred_images = pipe.vae.decode(lat, return_dict=False)[0]
    pred_images = pipe.image_processor.postprocess(pred_images, output_type="pil")

    if use_naive_scheduler:
        return latents_output, pred_images
    stacked = {key: torch.cat(vals, dim=1) for key, vals in rec.items()}
    return (latents_output, pred_images, {"x": stacked["x"], "epsilon": stacked["epsilon"]}, stacked["probs"], stacked["actions"],
            stacked["masks"])
