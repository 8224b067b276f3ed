"""PPO rollout consumer (config 5): decode -> reward -> advantages -> flattened PPO batch, and the loss value.

Mirror of the body of the training loop in train_ppo.py:352-427 up to (not including) the optimiser
step, with the reference's names:

* ``calculate_reward(reward_type, reward_model, reward_model_processor, model_pred, target, device)``
  (edit_ppo/reward_model.py:138-161).  Only the arithmetic-only rewards are on the path
  (SURVEY 8 a21): ``"image_psnr"`` (:484-509).  The backbone rewards (depth / inception / clip / ...)
  are third-party networks and out of scope; they raise.
* ``compute_advantages`` = train_ppo.py:376-390, ``ppo_loss`` = :408-421 (value), ``PolicyTrainer`` = the
  optimisation step :404-437 (gradients, clip_grad_norm_, AdamW) and the checkpoint format.
* ``collect_rollout`` = :352-403 for one batch of teacher pairs.

Everything runs in the HIP library (cs_image_psnr / cs_ppo_advantages / cs_ppo_loss); there is no
CPU fallback.
"""
import ctypes as C

import torch

from . import _lib as L
from .rollout import denoise_diffusion
from .vae import decode_latents

_DT = {torch.float32: 0, torch.float16: 1}


def _psnr(pred, target, clamp_hi):
    L.require_cuda(pred, "model_pred")
    L.require_cuda(target, "target")
    if pred.shape != target.shape:
        raise ValueError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)} (the bilinear-resize branch, "
                         "reward_model.py:492-494, is not on the hot path)")
    if pred.dtype != target.dtype:
        target = target.to(pred.dtype)
    if pred.dtype not in _DT:
        pred, target = pred.float(), target.float()
    pred, target = pred.contiguous(), target.contiguous()
    B = pred.shape[0]
    out = torch.empty(B, 1, dtype=torch.float32, device=pred.device)
    if B == 0:
        return out
    elems = pred[0].numel()
    lib = L.lib()
    ws = torch.empty(int(lib.cs_psnr_workspace_bytes(B)), dtype=torch.uint8, device=pred.device)
    L.check(lib.cs_image_psnr(L.ptr(pred), L.ptr(target), B, elems, _DT[pred.dtype], float(clamp_hi), L.ptr(out), L.ptr(ws), ws.numel(),
                              L.stream_ptr(pred.device)))
    return out


def calculate_image_psnr_reward(reward_model_processor, model_pred, target, device=None):
    """edit_ppo/reward_model.py:484-509: [B,3,H,W] in [0,1] x2 -> PSNR [B,1] clamped to [0, 100]."""
    return _psnr(model_pred, target, 100.0)


def depth_psnr_tail(pred_depth, target_depth):
    """edit_ppo/reward_model.py:404-422: PSNR of normalised depth maps [B,H,W], clamp(min=0) only -> [B,1]."""
    return _psnr(pred_depth, target_depth, 0.0)


def calculate_reward(reward_type, reward_model, reward_model_processor, model_pred, target, device=None):
    """edit_ppo/reward_model.py:138-161.  decode_latents already maps to [0, 1], so the clamp at :141-142 is a no-op here."""
    if reward_type == "image_psnr":
        return calculate_image_psnr_reward(reward_model_processor, model_pred, target, device)
    if reward_type in ("depth", "inception", "segmentation", "clip", "llava", "qwen_vl", "dino"):
        raise NotImplementedError(f"reward_type '{reward_type}' needs a third-party backbone network (out of scope, SURVEY 8 a21)")
    raise ValueError(f"Unknown reward_type: {reward_type}")


def compute_advantages(rewards, masks, num_inference_steps):
    """train_ppo.py:376-390: rewards [B,1] -> advantages [B (n-1), A] (normalised x10, tiled over the recorded steps, masked)."""
    L.require_cuda(rewards, "rewards")
    r = rewards.reshape(-1).to(torch.float32).contiguous()
    B, steps = r.shape[0], num_inference_steps - 1
    m = masks.reshape(B * steps, -1).to(torch.float32).contiguous()
    out = torch.empty_like(m)
    L.check(L.lib().cs_ppo_advantages(L.ptr(r), B, steps, m.shape[1], L.ptr(m), L.ptr(out), L.stream_ptr(r.device)))
    return out


def ppo_loss(curr_probs, old_probs, entropy, advantages, clip_range=0.2, entropy_coef=0.01):
    """train_ppo.py:408-421 (value only): clipped surrogate on the joint distribution - entropy bonus -> 0-d fp32 tensor."""
    L.require_cuda(curr_probs, "curr_probs")
    c, o, e, a = (t.to(torch.float32).contiguous() for t in (curr_probs, old_probs, entropy, advantages))
    R, A = c.shape
    if a.shape != (R, A):
        a = a.expand(R, A).contiguous()
    out = torch.empty(1, dtype=torch.float32, device=c.device)
    L.check(L.lib().cs_ppo_loss(L.ptr(c), L.ptr(o), L.ptr(e), L.ptr(a), R, A, float(clip_range), float(entropy_coef), L.ptr(out),
                                L.stream_ptr(c.device)))
    return out[0]


def collect_rollout(text_encoder, noise_scheduler, unet, vae, noise, text, tokenizer, target_latents, cfg=3.0, num_inference_steps=8,
                    reward_type="image_psnr", reward_model=None, reward_model_processor=None, decode_batch_size=8,
                    prompt_embeds=None, negative_prompt_embeds=None, identical_inputs=False):
    """train_ppo.py:352-403 for one batch: rollout, decode pred and teacher latents, reward, advantages, and the
    records flattened to [B (n-1), ...].  Returns a dict(conds, actions, probs, masks, advantages, rewards, model_pred).
    ``identical_inputs=True``: the B rows are copies of one (prompt, noise, teacher latent) sample (``repeat_random_sample``):
    the rollout shares the denoiser calls whose inputs cannot differ yet, and the teacher image is decoded once."""
    n = num_inference_steps
    model_pred, conds, probs, actions, masks, _ = denoise_diffusion(
        text_encoder, noise_scheduler, unet, noise, text, tokenizer, cfg=float(cfg), num_inference_steps=n,
        prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds, identical_inputs=identical_inputs)
    model_pred_decoded = decode_latents(vae, model_pred, batch_size=decode_batch_size)
    if identical_inputs and target_latents.shape[0] > 1:
        target_decoded = decode_latents(vae, target_latents[:1], batch_size=1).expand(target_latents.shape[0], -1, -1, -1).contiguous()
    else:
        target_decoded = decode_latents(vae, target_latents, batch_size=decode_batch_size)
    rewards = calculate_reward(reward_type, reward_model, reward_model_processor, model_pred_decoded, target_decoded, noise.device)
    B = model_pred.shape[0]
    flat = lambda v: v.reshape(v.shape[0] * (n - 1), *v.shape[2:])
    conds = {k: flat(v) for k, v in conds.items()}
    actions, probs, masks = (t.reshape(B * (n - 1), -1) for t in (actions, probs, masks))
    advantages = compute_advantages(rewards, masks, n)
    return dict(conds=conds, actions=actions, probs=probs, masks=masks, advantages=advantages, rewards=rewards, model_pred=model_pred)


class PolicyTrainer:
    """The optimisation step of train_ppo.py:404-437 on the HIP library: gradient of the clipped-surrogate + entropy loss
    with respect to ``factor_net``'s parameters (hand-derived backward, cs_ppo_policy_grads), ``clip_grad_norm_`` and
    ``torch.optim.AdamW`` semantics (cs_clip_grad_norm / cs_adamw_step), parameters updated in place.

    ``factor_net`` must hold fp32 parameters on the GPU (the reference trains the policy in fp32, train_ppo.py:205-229).
    Checkpoints use the reference's format: ``<dir>/checkpoint-<step>/model.ckpt`` = ``torch.save(factor_net.state_dict())``
    including the ``action_values`` buffer (train_ppo.py:174-178)."""

    def __init__(self, factor_net, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0,
                 clip_range=0.2, entropy_coef=0.01):
        self.net = factor_net
        self.lr, self.betas, self.weight_decay, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(weight_decay), float(eps)
        self.max_grad_norm, self.clip_range, self.entropy_coef = float(max_grad_norm), float(clip_range), float(entropy_coef)
        self.params = [factor_net.mlp[0].weight, factor_net.mlp[0].bias, factor_net.mlp[2].weight, factor_net.mlp[2].bias,
                       factor_net.mlp[4].weight, factor_net.mlp[4].bias]
        for p in self.params:
            L.require_cuda(p, "factor_net parameter (call .to('cuda') first)")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise TypeError("PolicyTrainer needs contiguous fp32 parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        if n != int(L.lib().cs_policy_param_count(C.byref(factor_net._net_struct()))):
            raise RuntimeError("parameter count mismatch between the module and the library")
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.grads)
        self.exp_avg_sq = torch.zeros_like(self.grads)
        self.step_count = 0
        self._ws = None
        self._out = torch.zeros(2, dtype=torch.float32, device=dev)       # loss, grad norm

    def grad_views(self):
        """per-parameter views of the packed gradient vector, state-dict order."""
        out, o = {}, 0
        for name, p in zip(("mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias", "mlp.4.weight", "mlp.4.bias"), self.params):
            out[name] = self.grads[o:o + p.numel()].view_as(p)
            o += p.numel()
        return out

    def compute_grads(self, conds, actions, old_probs, advantages):
        """-> loss (0-d tensor); gradients are left in ``self.grads`` (un-clipped)."""
        net = self.net
        x = L.require_cuda(conds["x"], "conds['x']").to(torch.float32).contiguous()
        R = x.shape[0]
        cosf = None
        if net.use_conv:
            eps = conds["epsilon"]
            cosf = net.cosine_features([eps[:, k] for k in range(net.order_dim)], net.order_dim)
        a, o, adv = (t.to(torch.float32).reshape(R, -1).contiguous() for t in (actions, old_probs, advantages))
        if adv.shape[1] != a.shape[1]:
            adv = adv.expand(R, a.shape[1]).contiguous()
        st = net._net_struct()
        lib = L.lib()
        need = int(lib.cs_policy_workspace_bytes(C.byref(st), R))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        L.check(lib.cs_ppo_policy_grads(C.byref(st), L.ptr(x), L.ptr(cosf), L.ptr(a), L.ptr(net._weights32()[6]), L.ptr(o), L.ptr(adv), R,
                                        self.clip_range, self.entropy_coef, L.ptr(self.grads), L.ptr(self._out), L.ptr(self._ws),
                                        self._ws.numel(), L.stream_ptr(x.device)))
        return self._out[0]

    def step(self, conds, actions, old_probs, advantages, dist=None):
        """one PPO epoch iteration (train_ppo.py:408-437) -> (loss, total grad norm before clipping), both 0-d tensors.
        ``dist``: a torch.distributed module whose default group spans the data-parallel ranks; the local gradients are then
        averaged over the ranks before clipping, which is what DDP does for the reference (one ~300 KB all-reduce over RCCL)."""
        loss = self.compute_grads(conds, actions, old_probs, advantages).clone()
        if dist is not None:
            from .launch import average_gradients
            average_gradients(dist, self.grads)
        lib, st = L.lib(), L.stream_ptr(self.grads.device)
        L.check(lib.cs_clip_grad_norm(L.ptr(self.grads), self.grads.numel(), self.max_grad_norm, L.ptr(self._out[1:]), st))
        self.step_count += 1
        o = 0
        for p in self.params:
            n = p.numel()
            L.check(lib.cs_adamw_step(L.ptr(p.data), L.ptr(self.grads[o:]), L.ptr(self.exp_avg[o:]), L.ptr(self.exp_avg_sq[o:]), n,
                                      self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.weight_decay, st))
            o += n
        return loss, self._out[1].clone()

    # -- checkpoint format of the reference (train_ppo.py:174-186) ---------------------------------------------
    def save_checkpoint(self, output_dir, global_step):
        import os
        d = os.path.join(output_dir, f"checkpoint-{global_step}")
        os.makedirs(d, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.net.state_dict().items()}, os.path.join(d, "model.ckpt"))
        return d

    def load_checkpoint(self, input_dir):
        import os
        sd = torch.load(os.path.join(input_dir, "model.ckpt"), map_location="cpu")
        with torch.no_grad():
            for k, v in self.net.state_dict().items():
                v.copy_(sd[k].to(v.device, v.dtype))
        return self


def train_iteration(trainer, text_encoder, noise_scheduler, unet, vae, batch, tokenizer, cfg=3.0, num_inference_steps=None, ppo_epochs=4,
                    reward_type="image_psnr", dist=None, prompt_embeds=None, negative_prompt_embeds=None, rng=None, share_identical=True):
    """One iteration of the training loop body (train_ppo.py:322-437): ``repeat_random_sample`` -> random step count in [2, 15]
    (:345) -> rollout, decode, reward, advantages (``collect_rollout``) -> ``ppo_epochs`` optimisation steps on the collected
    batch.  ``batch`` = (text list, noise [B,4,64,64], teacher latents [B,4,64,64]) already on the GPU.
    Returns dict(loss, norm, reward, num_inference_steps)."""
    import random
    from .ppo_data import repeat_random_sample
    rng = rng or random
    text, noise, tch, i = repeat_random_sample(batch, return_index=True)
    B = noise.shape[0]
    if prompt_embeds is not None and prompt_embeds.shape[0] == B:       # cached embeddings follow the item that was picked
        prompt_embeds = prompt_embeds[i:i + 1].expand(B, -1, -1).contiguous()
        if negative_prompt_embeds is not None:
            negative_prompt_embeds = negative_prompt_embeds[i:i + 1].expand(B, -1, -1).contiguous()
    n = num_inference_steps or rng.choice(list(range(2, 16)))
    # the B rows are copies of one sample by construction (data_processing.py:65-83): share what cannot differ
    roll = collect_rollout(text_encoder, noise_scheduler, unet, vae, noise, text, tokenizer, tch, cfg=cfg, num_inference_steps=n,
                           reward_type=reward_type, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                           identical_inputs=share_identical)
    loss = norm = None
    for _ in range(ppo_epochs):
        loss, norm = trainer.step(roll["conds"], roll["actions"], roll["probs"], roll["advantages"], dist=dist)
    return dict(loss=loss, norm=norm, reward=roll["rewards"].mean(), num_inference_steps=n)
