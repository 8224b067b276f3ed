"""PPO rollout loop (training trajectories) -- mirror of ``denoise_ppo.denoise_diffusion``
(denoise_ppo.py:6-120): same arguments, same 6-tuple
``(latents, conds{x, epsilon}, probs, actions, masks, prompt_embeds)`` with records for steps
``i > 0`` only (:105-111).

Differences in mechanism, not in results: the CFG dual batch is not materialised by
``torch.cat`` when the denoiser is the HIP UNet (it reads latent ``b % B``), and the CFG
combine ``u + g (c - u)`` (:96-100) is fused into the solver-update kernel.

``identical_inputs=True`` (extension): the caller states that all B rows of ``noise`` / ``text`` /
prompt embeddings are copies of ONE sample -- what the trainer feeds this function
(``repeat_random_sample``, data_processing.py:65-83, train_ppo.py:336).  The B trajectories then only
diverge through their sampled solver coefficients, which first enter the update at step 1 (one history
entry at step 0: eps_eff = eps, scheduler_ppo.py:263-265; scale actions, if any, act from step 0).  The
denoiser inputs of step 0 (and of step 1 when ``scaler_dim == 0``) are therefore the same for every row:
they are evaluated for one row and broadcast.  Every returned tensor keeps its full-batch shape.
"""
import torch

from . import _lib as L


def denoise_diffusion(text_encoder, scheduler, unet, noise, text, tokenizer, cfg=3, num_inference_steps=50,
                      gradient_checkpointing=False, prompt_embeds=None, negative_prompt_embeds=None, identical_inputs=False):
    if gradient_checkpointing:
        raise NotImplementedError("inference-only rollout: the denoiser is frozen (train_ppo.py:148-154)")
    if isinstance(text, str):
        text = [text]
    batch_size = len(text)
    L.require_cuda(noise, "noise")
    device = noise.device

    if prompt_embeds is None:
        ids = tokenizer(text, padding="max_length", max_length=tokenizer.model_max_length, truncation=True,
                        return_tensors="pt").input_ids.to(device)
        prompt_embeds = text_encoder(ids)[0]
    prompt_embeds_txt = prompt_embeds
    do_cfg = cfg > 1.0
    if do_cfg:
        if negative_prompt_embeds is None:
            ids = tokenizer([""] * batch_size, padding="max_length", max_length=tokenizer.model_max_length,
                            truncation=True, return_tensors="pt").input_ids.to(device)
            negative_prompt_embeds = text_encoder(ids)[0]
        prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])

    latents = noise.clone()
    scheduler.set_timesteps(num_inference_steps, device=device)
    record_prev = scheduler.record_conds
    scheduler.record_conds = True
    native = getattr(unet, "is_consolver_hip", False)
    rec = dict(x=[], epsilon=[], probs=[], actions=[], masks=[])
    try:
        shared_steps = 0
        if identical_inputs and native and batch_size > 1:
            # the caller's statement is checked once per rollout (one small device->host read): a batch that is NOT B copies of one sample
            # would otherwise get steps 0-1 computed from row 0 and broadcast, silently
            same = bool(torch.equal(noise[:1].expand_as(noise), noise)) and bool(torch.equal(prompt_embeds[:1].expand(batch_size, -1, -1), prompt_embeds[:batch_size]))
            if do_cfg:
                same = same and bool(torch.equal(prompt_embeds[batch_size:batch_size + 1].expand(batch_size, -1, -1), prompt_embeds[batch_size:]))
            if not same:
                raise ValueError("identical_inputs=True, but the rows of noise / prompt embeddings are not copies of one sample")
            shared_steps = 2 if scheduler.config.scaler_dim == 0 else 1
            pe1 = (torch.cat([prompt_embeds[:1], prompt_embeds[batch_size:batch_size + 1]]) if do_cfg else prompt_embeds[:1]).contiguous()
        for i, t in enumerate(scheduler.timesteps):
            if native and i < shared_steps:
                # every row carries the same latents: one row through the denoiser, broadcast to the batch
                e1 = unet(latents[:1], t, encoder_hidden_states=pe1, return_dict=False, dup=2 if do_cfg else 1, reuse_kv=(i > 0))[0]
                noise_pred = (torch.cat([e1[:1].expand(batch_size, -1, -1, -1), e1[1:].expand(batch_size, -1, -1, -1)]) if do_cfg
                              else e1.expand(batch_size, -1, -1, -1)).contiguous()
            elif native:
                # K/V of the prompt: computed at the first full-batch step of THIS rollout, reused afterwards
                noise_pred = unet(latents, t, encoder_hidden_states=prompt_embeds, return_dict=False,
                                  dup=2 if do_cfg else 1, reuse_kv=(i > shared_steps))[0]
            else:
                lat_in = torch.cat([latents] * 2) if do_cfg else latents
                lat_in = scheduler.scale_model_input(lat_in, t)
                noise_pred = unet(lat_in, t, encoder_hidden_states=prompt_embeds, return_dict=False)[0]
            if do_cfg:
                u, c = noise_pred[:batch_size], noise_pred[batch_size:]
                # CFG combine fused into the update kernel (use_conv: into the cosine-feature kernel, cs_cosine_features_cfg)
                out = scheduler.step(c, t, latents, return_dict=False, eps_uncond=u, guidance_scale=cfg)
            else:
                out = scheduler.step(noise_pred, t, latents, return_dict=False)
            latents, actions, probs, conds, masks = out
            if i > 0:
                rec["x"].append(conds["x"].unsqueeze(1))
                rec["epsilon"].append(conds["epsilon"].unsqueeze(1))
                rec["probs"].append(probs.unsqueeze(1))
                rec["actions"].append(actions.unsqueeze(1))
                rec["masks"].append(masks.unsqueeze(1))
    finally:
        scheduler.record_conds = record_prev
    # a step driven with a CUDA timestep that was not the grid entry the scheduler assumed must not reach the reward / update
    # silently: one device->host read per rollout (the records below are about to be consumed on the host side anyway)
    scheduler.verify_timesteps()
    cat = {k: torch.cat(v, dim=1) for k, v in rec.items()}
    return latents, {"x": cat["x"], "epsilon": cat["epsilon"]}, cat["probs"], cat["actions"], cat["masks"], prompt_embeds_txt

