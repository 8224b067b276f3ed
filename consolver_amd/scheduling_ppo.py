"""PPOScheduler -- the SD-side ConsistencySolver scheduler, HIP-backed.

Drop-in for the reference's ``PPOScheduler`` (scheduler_ppo.py:48-361): same
constructor arguments, ``set_timesteps`` / ``timesteps`` / ``init_noise_sigma`` /
``scale_model_input`` / ``step(...) -> (prev_sample, actions, probs, conds, masks)`` /
``add_noise`` / ``config`` / ``_compatibles`` / ``order`` and the ``factor_net``
attribute users reach into (``load_state_dict``, ``.to``).

Per ``step`` the device work is: [cosine features] -> policy MLP+softmax ->
categorical draw/gather -> ONE fused kernel (coefficient fix-up, linear
multistep combine over the cached eps history, optional CFG combine and scales,
DDIM update).  The host only does integer timestep arithmetic and table
look-ups -- no device->host synchronisation (the reference syncs three times
per step: scheduler_ppo.py:207,243,288).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import tables
from ._scheduler_base import (ConfigMixin, KARRAS_COMPATIBLES, SchedulerMixin, SolverConfig,  # noqa: F401
                              register_to_config)
from .factor_net import FactorNetPPO


class SolverOutput(dict):
    """what ``return_dict=True`` returns (the reference passes extra fields to a
    single-field diffusers dataclass and would raise, scheduler_ppo.py:299)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class HistoryMixin:
    """eps history + per-step policy evaluation shared by the SD and FLUX schedulers."""

    record_conds = False     # materialise conds['epsilon'] (needed only by the PPO rollout)
    verbose = False          # print coefficients like the reference (forces a host sync)
    prev_sample_dtype = None  # PPOScheduler.step: None = prev_sample has the sample's dtype; torch.float32 = promote a 16-bit sample (the reference with an fp32 policy net)

    def _policy(self, cond_row_f32, B, device, cfg=None, newest=None):
        """-> probs3 [B,A,K], actions [B,A], action_probs [B,A], idx.  ``cfg`` / ``newest``: use_conv under fused CFG -- the newest
        history slot (``self.ets[-1]``, the buffer the combined eps is written to) is read through ``newest`` (the text branch)."""
        net = self.factor_net.module if hasattr(self.factor_net, "module") else self.factor_net
        hist = self.ets[::-1]
        if newest is not None:
            hist = [newest] + hist[1:]
        x = cond_row_f32
        probs3 = net.probs_from(x, hist if net.use_conv else None, len(hist), batch=B, cfg=cfg)
        actions, aprobs, idx = net.draw(probs3)
        return probs3, actions, aprobs, idx

    def _masks(self, B, A, m, device):
        """[B, A] 0 / 1 mask of the action dims a step with m history entries uses.  A function of (B, A, m, order) only: computed once per shape and device
        (cs_step_masks) and handed out again -- callers treat step outputs as read-only (the rollout stacks them)."""
        cache = self.__dict__.setdefault("_mask_cache", {})
        key = (B, A, m, self.config.order_dim, str(device))
        masks = cache.get(key)
        if masks is None:
            masks = torch.empty(B, A, dtype=torch.float32, device=device)
            L.check(L.lib().cs_step_masks(B, A, m, self.config.order_dim, L.ptr(masks), L.stream_ptr(device)))
            if not torch.cuda.is_current_stream_capturing():      # (a tensor created under capture belongs to the graph's pool)
                cache[key] = masks
        return masks

    def _stack(self, B, elems, like):
        order = self.config.order_dim
        hist = [h.contiguous() for h in self.ets[::-1]]
        out = torch.empty((B, order) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
        arr = (C.c_void_p * L.CS_MAX_ORDER)(*[h.data_ptr() for h in hist])
        L.check(L.lib().cs_stack_history(arr, len(hist), order, B, elems, L.dtype_code(like.dtype), L.ptr(out),
                                         L.stream_ptr(like.device)))
        return out

    def _fill_step_args(self, a, sample, eps_text, eps_uncond, guidance, actions, out, eps_out, out_dtype, out_lp=None):
        B = sample.shape[0]
        hist = self.ets[::-1]  # newest first, hist[0] is the eps just pushed
        a.x, a.eps_text = sample.data_ptr(), eps_text.data_ptr()
        a.eps_uncond = eps_uncond.data_ptr() if eps_uncond is not None else None
        a.guidance = float(guidance)
        for k in range(L.CS_MAX_ORDER):
            a.hist[k] = hist[k + 1].data_ptr() if k + 1 < len(hist) else None
        a.m, a.order_dim, a.scaler_dim = len(hist), self.config.order_dim, self.config.scaler_dim
        a.actions, a.actions_stride = actions.data_ptr(), actions.shape[1]
        a.B, a.elems = B, sample.numel() // max(B, 1)
        # io_dtype is the model output's (eps, history, eps_out); an fp32 sample next to a 16-bit model output is passed as such (x_is_f32): the update
        # reads it unrounded, which is what torch's promotion does with an fp32 `sample` (scheduler_ppo.py:306-332, scheduler_fmppo.py:354)
        a.io_dtype, a.out_dtype = L.dtype_code(eps_text.dtype), L.dtype_code(out_dtype)
        a.x_is_f32 = int(sample.dtype == torch.float32 and eps_text.dtype != torch.float32)
        a.x_out = out.data_ptr()
        a.eps_out = eps_out.data_ptr() if eps_out is not None else None
        if out_lp is not None:
            # the model-dtype copy of an fp32 prev_sample, written by the same kernel pass (CsStepArgs::x_out_lp): what the denoiser reads at the next step
            if out_lp.dtype not in (torch.float16, torch.bfloat16) or out.dtype != torch.float32 or out_lp.shape != out.shape or not out_lp.is_contiguous():
                raise ValueError("step(out_lp=...) is the contiguous fp16 / bf16 copy of an fp32 prev_sample of the same shape")
            a.x_out_lp, a.lp_dtype = out_lp.data_ptr(), L.dtype_code(out_lp.dtype)


class PPOScheduler(HistoryMixin, SchedulerMixin, ConfigMixin):
    """``class PPOScheduler(SchedulerMixin, ConfigMixin)`` of scheduler_ppo.py:48: diffusers' mixins when diffusers is
    importable, the stand-alone ones of ``_scheduler_base`` otherwise -- either way ``config`` (backed by
    ``_internal_dict``, which ``StableDiffusionPipeline.__init__`` rewrites when ``steps_offset != 1``),
    ``save_config`` / ``from_config`` / ``save_pretrained`` / ``from_pretrained`` / ``compatibles``."""
    _compatibles = list(KARRAS_COMPATIBLES)
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, prediction_type="epsilon", timestep_spacing="leading", steps_offset=0,
                 order_dim=4, scaler_dim=2, use_conv=False, ppo_type="discrete", factor_net_kwargs=None):
        if not (1 < order_dim <= L.CS_MAX_ORDER):
            # order_dim=1 raises IndexError inside the reference (scheduler_ppo.py:166 on an empty list)
            raise ValueError(f"order_dim must be in [2, {L.CS_MAX_ORDER}]")
        self._betas = tables.make_betas(beta_schedule, beta_start, beta_end, num_train_timesteps, trained_betas)
        self._ac = tables.alphas_cumprod(self._betas)
        self.betas = torch.from_numpy(self._betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.from_numpy(self._ac)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self._timesteps = np.arange(0, num_train_timesteps)[::-1].copy()
        self.timesteps = torch.from_numpy(self._timesteps)
        self.ets = []
        self._grid_index = {int(v): i for i, v in enumerate(self._timesteps)}
        self._step_counter = None
        self._t_mismatch = None
        kw = dict(factor_net_kwargs) if factor_net_kwargs is not None else {}
        kw["order_dim"], kw["scaler_dim"], kw["use_conv"] = order_dim, scaler_dim, use_conv
        kw.setdefault("embedding_dim", 32)
        kw.setdefault("hidden_dim", 256)
        if ppo_type != "discrete":
            # the reference imports a module that does not exist upstream (scheduler_ppo.py:23,138-139)
            raise NotImplementedError("only ppo_type='discrete' exists in the reference")
        kw.setdefault("num_actions", 161)
        self.factor_net = FactorNetPPO(**kw)
        self._cond_table = None

    # ------------------------------------------------------------------ protocol
    def set_timesteps(self, num_inference_steps, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError(f"`num_inference_steps` ({num_inference_steps}) cannot be larger than "
                             f"`num_train_timesteps` ({T}).")
        ts = tables.sd_timestep_grid(num_inference_steps, T, self.config.timestep_spacing, self.config.steps_offset)
        if ts.min() < 0 or ts.max() >= T:
            # e.g. trailing with n=61 yields 62 entries ending in -1 in the reference (it would wrap around)
            raise ValueError(f"timestep grid for n={num_inference_steps} leaves [0, {T}): {ts.min()}..{ts.max()}")
        if self._t_mismatch is not None and not torch.cuda.is_current_stream_capturing():
            self.verify_timesteps()
        self.num_inference_steps = num_inference_steps
        self._timesteps = ts
        self._grid_index = {int(v): i for i, v in enumerate(ts)}
        self._step_counter = None
        same = (self._cond_table is not None and self._cond_table[3] == num_inference_steps and device is not None
                and self.timesteps.device == torch.device(device))
        if not same:
            self.timesteps = torch.from_numpy(ts).to(device)
            self._cond_table = None            # (kept across calls with the same n/device: no H2D, graph-capture safe)
        self.ets = []

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    def _resolve_timestep(self, timestep):
        """integer value of ``timestep`` without a device->host sync per step.

        Host ints / CPU tensors are read directly.  A CUDA element of ``self.timesteps`` (what
        ``for t in scheduler.timesteps`` yields) is identified by its address.  Any other CUDA tensor
        (``t.clone()``, ``t + 0``, a caller's own grid) costs ONE ``.item()`` for the first such step after
        ``set_timesteps`` -- that anchors a host-side step counter on the grid -- and later steps take the
        next grid entry from the counter, verified on the device without a sync: the accumulated mismatch flag is
        read ONCE when the trajectory ends (the step that consumes the last grid entry, see ``step``), by
        ``verify_timesteps()`` (the rollout and the engine call it), and by the next ``set_timesteps`` -- a step
        driven with a different value raises there, before its results are consumed.  The reference itself syncs
        on every step (CPU table indexed by the CUDA timestep, scheduler_ppo.py:309-312)."""
        grid = self._timesteps
        if not isinstance(timestep, torch.Tensor):
            v = int(timestep)
        elif not timestep.is_cuda:
            v = int(timestep)
        else:
            ts = self.timesteps
            v = None
            if ts.is_cuda and ts.device == timestep.device and timestep.dtype == ts.dtype:
                off = timestep.data_ptr() - ts.data_ptr()
                if 0 <= off < ts.numel() * ts.element_size() and off % ts.element_size() == 0:
                    v = int(grid[off // ts.element_size()])
            if v is None:
                i = self._step_counter
                if i is None or i >= len(grid):
                    v = int(timestep.item())                     # anchor (first foreign tensor) or past the grid
                else:
                    v = int(grid[i])
                    bad = (timestep.reshape(-1)[:1] != v).to(torch.int32)
                    self._t_mismatch = bad if self._t_mismatch is None else self._t_mismatch + bad
        i = self._grid_index.get(v)
        self._step_counter = None if i is None else i + 1
        return v

    def verify_timesteps(self):
        """raise if a step whose timestep was taken from the host-side counter was driven with another value
        (one device->host read; called by ``set_timesteps`` outside stream capture, or by the user)."""
        m, self._t_mismatch = self._t_mismatch, None
        if m is not None and int(m.item()) != 0:
            raise RuntimeError("PPOScheduler.step was called with CUDA timesteps that are not consecutive entries of "
                               "scheduler.timesteps; pass elements of scheduler.timesteps or host integers")

    def _cond_row(self, t, prev_t, dtype, device):
        """[1,2] fp32 device row (t, prev_t) rounded through the model dtype like
        ``torch.tensor([[t, prev_t]], dtype=model_output.dtype)`` (scheduler_ppo.py:207)."""
        if (self._cond_table is None or self._cond_table[0].device != device or self._cond_table[2] != dtype
                or self._cond_table[3] != self.num_inference_steps):
            n = self.num_inference_steps
            host = np.stack([self._timesteps, self._timesteps - self.config.num_train_timesteps // n], 1)
            dev = torch.from_numpy(host.astype(np.float32)).to(device).to(dtype).to(torch.float32)
            self._cond_table = (dev, {int(v): i for i, v in enumerate(self._timesteps)}, dtype, n)
        dev, index, _, _ = self._cond_table
        i = index.get(int(t))
        if i is None:
            return torch.tensor([[t, prev_t]], dtype=dtype).to(torch.float32).to(device)
        return dev[i:i + 1]

    def _cond_rows(self, t, cond_row, B, dtype):
        """conds['x'] = the (t, prev_t) row in the model dtype, repeated for the batch (scheduler_ppo.py:207): a function of (t, B, dtype), cached per grid entry
        like the masks (two tiny launches per step otherwise); read-only by convention"""
        cache = self.__dict__.setdefault("_cond_rows_cache", {})
        key = (int(t), B, dtype, str(cond_row.device), self.num_inference_steps)
        rows = cache.get(key)
        if rows is None:
            rows = cond_row.to(dtype).repeat(B, 1)
            if not torch.cuda.is_current_stream_capturing():
                if len(cache) > 256:
                    cache.clear()
                cache[key] = rows
        return rows

    def _ddim_scalars(self, t, prev_t):
        a_t = self._ac[t]
        a_p = self._ac[prev_t] if prev_t >= 0 else self._ac[0]
        one = np.float32(1.0)
        return (float(np.sqrt(a_t)), float(np.sqrt(one - a_t)), float(np.sqrt(a_p)), float(np.sqrt(one - a_p)))

    def step(self, model_output, timestep, sample, return_dict=True, *, eps_uncond=None, guidance_scale=1.0,
             eps_out=None, out=None, out_lp=None):
        """scheduler_ppo.py:178-299.  Extension (keyword-only): pass the two CFG branches as
        ``model_output`` (text) + ``eps_uncond`` and the combine u + g (c - u) is fused into the
        update kernel; ``eps_out`` receives the combined eps (the history entry); ``out_lp`` (with an fp32 sample) receives the
        model-dtype rounding of ``prev_sample`` in the same kernel pass (the denoiser's input at the next step)."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None'. Call 'set_timesteps' first.")
        if self.config.prediction_type not in ("epsilon", "v_prediction"):
            raise ValueError(f"Unsupported prediction_type: {self.config.prediction_type}")
        if self.config.scaler_dim > 2:
            raise AssertionError("not implemented")     # scheduler_ppo.py:280
        L.require_cuda(model_output, "model_output")
        L.require_cuda(sample, "sample")
        model_output = model_output.contiguous()
        sample = sample.contiguous()
        # dtype of the latents: an fp32 `sample` stays fp32 through the update (torch promotion; the reference's latents ARE fp32 from step 2 on when the policy
        # net is fp32) -- the native engine keeps its solver state that way: the per-step fp16 rounding of the latents (2.8e-4 relative L2 per step,
        # accumulating in quadrature) was what made the 8-step drift grow and the 12-step trajectory miss the 1e-3 gate.  A 16-bit sample
        # keeps its dtype unless `prev_sample_dtype = torch.float32` asks for the promotion.
        if sample.dtype != model_output.dtype and sample.dtype != torch.float32:
            sample = sample.to(model_output.dtype)
        if self.prev_sample_dtype is not None and sample.dtype != self.prev_sample_dtype:
            sample = sample.to(self.prev_sample_dtype)
        dev = model_output.device
        t = self._resolve_timestep(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        B = model_output.shape[0]

        if eps_uncond is not None:
            eps_uncond = L.require_cuda(eps_uncond, "eps_uncond").contiguous()
            if eps_out is None:
                eps_out = torch.empty_like(model_output)
            current = eps_out
        else:
            current = model_output
        self.ets.append(current)
        self.ets = self.ets[-self.config.order_dim:]
        m = len(self.ets)

        net = self.factor_net.module if hasattr(self.factor_net, "module") else self.factor_net
        cond_row = self._cond_row(t, prev_t, model_output.dtype, dev)
        if net.use_conv and eps_uncond is not None:
            # the policy's cosine features need the combined eps (scheduler_ppo.py:207-240 after denoise_ppo.py:96-100): the
            # feature kernel forms u + g (c - u) on the fly, leaves it in eps_out, and the update kernel takes it as combined
            probs3, actions, aprobs, _ = self._policy(cond_row, B, dev, cfg=(eps_uncond, guidance_scale, eps_out), newest=model_output)
            model_output, eps_uncond, eps_out = eps_out, None, None
        else:
            probs3, actions, aprobs, _ = self._policy(cond_row, B, dev)
        masks = self._masks(B, net.action_dims, m, dev)

        prev = out if out is not None else torch.empty_like(sample)
        if prev.dtype != sample.dtype:
            raise ValueError(f"step(out=...) must have the sample's dtype {sample.dtype}, got {prev.dtype}")
        a = L.CsStepArgs()
        self._fill_step_args(a, sample, model_output, eps_uncond, guidance_scale, actions, prev, eps_out, sample.dtype, out_lp)
        a.sqrt_at, a.sqrt_1mat, a.sqrt_ap, a.sqrt_1map = self._ddim_scalars(t, prev_t)
        a.v_prediction = int(self.config.prediction_type == "v_prediction")
        L.check(L.lib().cs_lms_ddim_step(C.byref(a), L.stream_ptr(dev)))
        if (self._t_mismatch is not None and self._step_counter is not None and self._step_counter >= len(self._timesteps)
                and not torch.cuda.is_current_stream_capturing()):
            self.verify_timesteps()       # end of the trajectory: one device->host read, before the caller consumes the result

        conds = {"x": self._cond_rows(t, cond_row, B, model_output.dtype),
                 "epsilon": self._stack(B, a.elems, current) if (self.record_conds or net.use_conv) else None}
        if self.verbose:
            print(f"T={t} -> {prev_t} | actions: {actions[0].tolist()} | Prob: {aprobs[0].tolist()}")
        if not return_dict:
            return (prev, actions, aprobs, conds, masks)
        return SolverOutput(prev_sample=prev, actions=actions, probs=aprobs, conds=conds, masks=masks)

    def add_noise(self, original_samples, noise, timesteps):
        """scheduler_ppo.py:336-358 (forward process; not on the sampling path)."""
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while len(sa.shape) < len(original_samples.shape):
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original_samples + sb * noise
