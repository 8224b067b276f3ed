"""Baseline solvers for A/B against the learned ConsistencySolver.

* ``DDIMBaselineScheduler`` -- eta = 0 DDIM on the SAME tables / timestep grids / ``prev_t = t - T // n`` rule as
  ``PPOScheduler`` (scheduler_ppo.py:203,306-332), i.e. the learned solver with every coefficient at its default
  (order 1: eps_eff = eps_t).  It runs the same fused update kernel (cs_lms_ddim_step with a history of one).
* ``FlowMatchEulerBaselineScheduler`` -- the ``type == "euler"`` branch of edit_ppo/scheduler_fm.py:405-410
  (``x' = x + (sigma_next - sigma) v``) on ``FMPPOScheduler``'s sigma schedule (cs_lms_euler_step, history of one).

Both keep the scheduler protocol (``set_timesteps``, ``timesteps``, ``step(...)[0]``, ``init_noise_sigma``,
``scale_model_input``) and have no policy network: ``step`` returns ``(prev_sample,)`` / an object with ``prev_sample``.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .scheduling_fmppo import FMPPOScheduler
from .scheduling_ppo import PPOScheduler, SolverOutput


class DDIMBaselineScheduler(PPOScheduler):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear", trained_betas=None,
                 prediction_type="epsilon", timestep_spacing="leading", steps_offset=0):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, prediction_type, timestep_spacing,
                         steps_offset, order_dim=2, scaler_dim=0, factor_net_kwargs=dict(hidden_dim=4, num_actions=3))
        self.factor_net = None                      # no policy
        self._zero_actions = None

    def step(self, model_output, timestep, sample, return_dict=True, *, eps_uncond=None, guidance_scale=1.0, out=None, out_lp=None, **_ignored):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None'. Call 'set_timesteps' first.")
        L.require_cuda(model_output, "model_output")
        L.require_cuda(sample, "sample")
        model_output, sample = model_output.contiguous(), sample.contiguous()
        # same dtype rule as PPOScheduler.step: an fp32 sample next to a 16-bit model output stays fp32 through the update (CsStepArgs::x_is_f32, the
        # engine's default solver state); any other mismatch is cast to the model output's dtype
        if sample.dtype != model_output.dtype and sample.dtype != torch.float32:
            sample = sample.to(model_output.dtype)
        if self.prev_sample_dtype is not None and sample.dtype != self.prev_sample_dtype:
            sample = sample.to(self.prev_sample_dtype)
        dev, B = model_output.device, model_output.shape[0]
        t = self._resolve_timestep(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        if eps_uncond is not None:
            eps_uncond = L.require_cuda(eps_uncond, "eps_uncond").contiguous()
        self.ets = [model_output]                   # history of one: eps_eff = eps_t
        if self._zero_actions is None or self._zero_actions.shape[0] < B or self._zero_actions.device != dev:
            self._zero_actions = torch.zeros(max(B, 1), 1, dtype=torch.float32, device=dev)
        prev = out if out is not None else torch.empty_like(sample)
        if prev.dtype != sample.dtype:
            raise ValueError(f"step(out=...) must have the sample's dtype {sample.dtype}, got {prev.dtype}")
        eps_out = torch.empty_like(model_output) if eps_uncond is not None else None     # the kernel writes the combined eps
        a = L.CsStepArgs()
        self._fill_step_args(a, sample, model_output, eps_uncond, guidance_scale, self._zero_actions, prev, eps_out, sample.dtype, out_lp)
        a.sqrt_at, a.sqrt_1mat, a.sqrt_ap, a.sqrt_1map = self._ddim_scalars(t, prev_t)
        a.v_prediction = int(self.config.prediction_type == "v_prediction")
        L.check(L.lib().cs_lms_ddim_step(C.byref(a), L.stream_ptr(dev)))
        return (prev,) if not return_dict else SolverOutput(prev_sample=prev)


class FlowMatchEulerBaselineScheduler(FMPPOScheduler):
    def __init__(self, **kw):
        for k in ("order_dim", "scaler_dim", "mu_dim", "factor_net_kwargs"):
            kw.pop(k, None)
        super().__init__(order_dim=2, scaler_dim=0, mu_dim=0, factor_net_kwargs=dict(hidden_dim=4, num_actions=3), **kw)
        self.factor_net = None
        self._zero_actions = None

    def step(self, model_output, timestep, sample, return_dict=True, *, out=None, **_ignored):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None'. Call 'set_timesteps' first.")
        if self._step_index is None:
            self._init_step_index(timestep)
        L.require_cuda(model_output, "model_output")
        L.require_cuda(sample, "sample")
        model_output = model_output.contiguous()
        sample = sample.contiguous()
        if sample.dtype != model_output.dtype and sample.dtype != torch.float32:        # (an fp32 sample is consumed as fp32: scheduler_fm.py upcasts it, FMPPOScheduler.step does the same)
            sample = sample.to(model_output.dtype)
        dev, B = model_output.device, model_output.shape[0]
        i = self._step_index
        dt = np.float32(self._sigmas[i + 1] - self._sigmas[i])
        self.ets = [model_output]
        if self._zero_actions is None or self._zero_actions.shape[0] < B or self._zero_actions.device != dev:
            self._zero_actions = torch.zeros(max(B, 1), 1, dtype=torch.float32, device=dev)
        prev = out if out is not None else torch.empty_like(model_output)
        if prev.dtype != model_output.dtype:                                           # the Euler result is rounded to the model dtype (scheduler_fm.py:410), whatever the sample's
            raise ValueError(f"step(out=...) must have the model output's dtype {model_output.dtype}, got {prev.dtype}")
        a = L.CsStepArgs()
        self._fill_step_args(a, sample, model_output, None, 1.0, self._zero_actions, prev, None, model_output.dtype)
        a.dt = float(dt)
        L.check(L.lib().cs_lms_euler_step(C.byref(a), L.stream_ptr(dev)))
        self._step_index += 1
        return (prev,) if not return_dict else SolverOutput(prev_sample=prev)
