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

``solver_state_dtype`` (extension, default ``torch.float32``): the dtype the LATENTS are carried in between the steps of the loop.
The reference hands ``scheduler.step`` whatever the previous step returned: an fp16 tensor after step 1 and -- because the fp32
policy's ``[B,1,1,1]`` actions promote the update (scheduler_ppo.py:263-272) -- fp32 tensors afterwards.  Here the
state is fp32 from step 0 on: the denoiser reads its fp16 rounding (``HipUNet2DConditionModel`` casts its input), the update kernel
reads and writes the unrounded state (CsStepArgs::x_is_f32).  An fp16 state rounds the latents once per step (2.8e-4 relative L2
each, adding in quadrature): 12 steps ended at 1.10e-3 of the fp32 oracle, above north_star's 1e-3 gate.  The
RETURNED latents are cast back to ``noise.dtype`` -- the tensor the trainer decodes (train_ppo.py:359-365) -- and every record
(``conds`` / ``probs`` / ``actions`` / ``masks``) has the dtype it has in the reference.  ``solver_state_dtype=None`` carries the
state in ``noise.dtype`` (the arithmetic class of an all-fp16 loop).  Round 6: with the fp32 state the native denoiser also hands its OUTPUT over in fp32
(``unet(..., out_dtype=torch.float32)``: conv_out's accumulator unrounded, and the output head's two-plane form) -- the update and the policy's features read it, the
``conds`` records are its rounding to ``noise.dtype``, the dtype they have in the reference.
"""
import torch

from . import _lib as L


def denoise_diffusion(text_encoder, scheduler, unet, noise, text, tokenizer, cfg=3, num_inference_steps=50,
                      gradient_checkpointing=False, prompt_embeds=None, negative_prompt_embeds=None, identical_inputs=False,
                      solver_state_dtype=torch.float32):
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

    native = getattr(unet, "is_consolver_hip", False)
    if solver_state_dtype not in (None, torch.float32):
        raise ValueError("solver_state_dtype must be torch.float32 (default) or None (= noise.dtype)")
    # the solver state: fp32 between the steps (see the module docstring); a foreign denoiser gets the tensor in its own dtype below
    latents = noise.to(solver_state_dtype) if (solver_state_dtype is not None and native and noise.dtype != solver_state_dtype) else noise.clone()
    # the denoiser's output in fp32 next to an fp32 state (module docstring); records keep the model dtype
    eps_dtype = torch.float32 if (native and latents.dtype == torch.float32 and noise.dtype != torch.float32) else None
    rec_dtype = noise.dtype
    scheduler.set_timesteps(num_inference_steps, device=device)
    record_prev = scheduler.record_conds
    scheduler.record_conds = True
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
                e1 = unet(latents[:1], t, encoder_hidden_states=pe1, return_dict=False, dup=2 if do_cfg else 1, reuse_kv=(i > 0), out_dtype=eps_dtype)[0]
                noise_pred = (torch.cat([e1[:1].expand(batch_size, -1, -1, -1), e1[1:].expand(batch_size, -1, -1, -1)]) if do_cfg
                              else e1.expand(batch_size, -1, -1, -1)).contiguous()
            elif native:
                # K/V of the prompt: computed at the first full-batch step of THIS rollout, reused afterwards
                noise_pred = unet(latents, t, encoder_hidden_states=prompt_embeds, return_dict=False,
                                  dup=2 if do_cfg else 1, reuse_kv=(i > shared_steps), out_dtype=eps_dtype)[0]
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
                rec["x"].append(conds["x"].to(rec_dtype).unsqueeze(1))
                rec["epsilon"].append(conds["epsilon"].to(rec_dtype).unsqueeze(1))
                rec["probs"].append(probs.unsqueeze(1))
                rec["actions"].append(actions.unsqueeze(1))
                rec["masks"].append(masks.unsqueeze(1))
    finally:
        scheduler.record_conds = record_prev
    # a step driven with a CUDA timestep that was not the grid entry the scheduler assumed must not reach the reward / update
    # silently: one device->host read per rollout (the records below are about to be consumed on the host side anyway)
    scheduler.verify_timesteps()
    cat = {k: torch.cat(v, dim=1) for k, v in rec.items()}
    return latents.to(noise.dtype), {"x": cat["x"], "epsilon": cat["epsilon"]}, cat["probs"], cat["actions"], cat["masks"], prompt_embeds_txt

