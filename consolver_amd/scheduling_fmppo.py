"""FMPPOScheduler -- the FLUX-side (flow matching) ConsistencySolver scheduler, HIP-backed.

Drop-in for the reference's ``FMPPOScheduler`` (edit_ppo/scheduler_fmppo.py:56-553):
``set_timesteps(num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None)``,
``set_begin_index``, ``step(model_output, timestep, sample, ..., return_dict)``,
``scale_noise``, ``sigmas`` / ``timesteps`` / ``step_index`` / ``begin_index`` / ``shift`` /
``config`` and ``from_pretrained(repo, subfolder=..., order_dim=..., ...)`` keyword overrides
(edit_ppo/generate_ours.py:127-134).

Per step: policy MLP+softmax(logits/0.01) -> draw/gather -> ONE fused kernel
``x' = x_fp32 + (sigma_{i+1} - sigma_i) * sum_k c_k v_{t-k}`` rounded once to the model dtype.
"""
import ctypes as C
import os
import warnings

import numpy as np
import torch

from . import _lib as L
from . import tables
from .factor_net import FluxFactorNetPPO
from ._scheduler_base import ConfigMixin, HAVE_DIFFUSERS, SchedulerMixin, register_to_config
from .scheduling_ppo import HistoryMixin, SolverOutput

# FLUX.1-Kontext-dev's published scheduler/scheduler_config.json: what from_pretrained falls back to offline
KONTEXT_SCHEDULER_CONFIG = dict(shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                                base_image_seq_len=256, max_image_seq_len=4096)


class FMPPOScheduler(HistoryMixin, SchedulerMixin, ConfigMixin):
    """``class FMPPOScheduler(SchedulerMixin, ConfigMixin)`` of edit_ppo/scheduler_fmppo.py:56 (mixins: see
    ``_scheduler_base``)."""
    _compatibles = []
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False, base_shift=0.5,
                 max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096, invert_sigmas=False,
                 shift_terminal=None, use_karras_sigmas=False, use_exponential_sigmas=False,
                 use_beta_sigmas=False, time_shift_type="exponential", stochastic_sampling=False,
                 order_dim=4, scaler_dim=2, mu_dim=1, use_conv=False, ppo_type="discrete",
                 factor_net_kwargs=None):
        if sum([use_beta_sigmas, use_exponential_sigmas, use_karras_sigmas]) > 1:
            raise ValueError("Only one of `use_beta_sigmas`, `use_exponential_sigmas`, `use_karras_sigmas` can be used.")
        if time_shift_type not in {"exponential", "linear"}:
            raise ValueError("`time_shift_type` must either be 'exponential' or 'linear'.")
        if not (1 < order_dim <= L.CS_MAX_ORDER):
            raise ValueError(f"order_dim must be in [2, {L.CS_MAX_ORDER}]")
        T = num_train_timesteps
        ts = np.linspace(1, T, T, dtype=np.float32)[::-1].copy()
        sig = (ts / np.float32(T)).astype(np.float32)
        if not use_dynamic_shifting:
            sig = (np.float32(shift) * sig / (np.float32(1) + np.float32(shift - 1) * sig)).astype(np.float32)
        self._sigmas = sig
        self.sigmas = torch.from_numpy(sig)
        self.timesteps = self.sigmas * T
        self.sigma_min, self.sigma_max = float(sig[-1]), float(sig[0])
        self._shift = shift
        self._step_index = None
        self._begin_index = None
        self.num_inference_steps = None
        self.ets = []
        kw = dict(factor_net_kwargs) if factor_net_kwargs is not None else {}
        kw["order_dim"], kw["scaler_dim"], kw["mu_dim"], kw["use_conv"] = order_dim, scaler_dim, mu_dim, use_conv
        kw.setdefault("embedding_dim", 32)
        kw.setdefault("hidden_dim", 256)
        if ppo_type != "discrete":
            raise AssertionError("only ppo_type='discrete' exists (scheduler_fmppo.py:169-170)")
        kw.setdefault("num_actions", 161)
        self.factor_net = FluxFactorNetPPO(**kw)
        self._cond_dev = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder=None, return_unused_kwargs=False, **kw):
        """``FMPPOScheduler.from_pretrained(repo, subfolder="scheduler", order_dim=..., ...)`` (edit_ppo/generate_ours.py:127-134).
        A local directory holding ``scheduler_config.json`` (or, with diffusers installed, anything its loader resolves) is
        loaded through the mixin; a hub id that cannot be resolved (no network) falls back to FLUX.1-Kontext's published
        scheduler config with the keyword overrides on top."""
        src = pretrained_model_name_or_path
        if isinstance(src, dict):
            return cls.from_config({**KONTEXT_SCHEDULER_CONFIG, **src}, return_unused_kwargs=return_unused_kwargs, **kw)
        local = src is not None and (os.path.exists(os.path.join(str(src), subfolder or "", cls.config_name)) or os.path.isfile(str(src)))
        if not local and src is not None and not HAVE_DIFFUSERS:
            # a filesystem-looking path that does not exist is a mistake, not a hub id: plausible-but-wrong shift parameters must not load silently
            looks_like_path = (os.path.isabs(str(src)) or str(src).startswith((".", "~")) or os.path.isdir(str(src)) or str(src).count("/") > 1
                               or str(src).endswith(".json"))
            if looks_like_path:
                raise EnvironmentError(f"{src!r}: no {cls.config_name} under {os.path.join(str(src), subfolder or '')!r}")
            warnings.warn(f"{src!r} is taken for a hub id (no network here): using the published FLUX.1-Kontext scheduler config")
        if local and os.path.isfile(str(src)):
            import json
            with open(str(src)) as f:
                return cls.from_config({**KONTEXT_SCHEDULER_CONFIG, **{k: v for k, v in json.load(f).items() if not k.startswith("_")}},
                                       return_unused_kwargs=return_unused_kwargs, **kw)
        if local or (HAVE_DIFFUSERS and src is not None):
            try:
                return super().from_pretrained(src, subfolder=subfolder, return_unused_kwargs=return_unused_kwargs, **kw)
            except (EnvironmentError, OSError, ValueError) as e:
                if local:
                    raise
                warnings.warn(f"{src!r} could not be resolved ({type(e).__name__}); using the published FLUX.1-Kontext scheduler config")
        return cls.from_config(dict(KONTEXT_SCHEDULER_CONFIG), return_unused_kwargs=return_unused_kwargs, **kw)

    # ------------------------------------------------------------------ properties
    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    @property
    def shift(self):
        return self._shift

    def set_begin_index(self, begin_index=0):
        self._begin_index = begin_index

    def set_shift(self, shift):
        self._shift = shift

    def _sigma_to_t(self, sigma):
        return sigma * self.config.num_train_timesteps

    def __len__(self):
        return self.config.num_train_timesteps

    # ------------------------------------------------------------------ protocol
    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, mu=None, timesteps=None):
        sig, ts = tables.flux_sigma_schedule(self.config, self._shift, self.sigma_min, self.sigma_max,
                                             num_inference_steps, sigmas, mu, timesteps)
        self.num_inference_steps = len(ts)
        self._sigmas, self._timesteps = sig, ts
        self.sigmas = torch.from_numpy(sig).to(device)
        self.timesteps = torch.from_numpy(ts).to(device)
        self._step_index = None
        self._begin_index = None
        self.ets = []
        self._cond_dev = None

    def index_for_timestep(self, timestep, schedule_timesteps=None):
        ts = self._timesteps if schedule_timesteps is None else np.asarray(schedule_timesteps.cpu())
        idx = np.nonzero(ts == np.float32(float(timestep)))[0]
        return int(idx[1 if len(idx) > 1 else 0])

    def _init_step_index(self, timestep):
        if self._begin_index is None:
            if isinstance(timestep, torch.Tensor) and timestep.is_cuda and self.timesteps.is_cuda:
                off = timestep.data_ptr() - self.timesteps.data_ptr()
                es = self.timesteps.element_size()
                if 0 <= off < self.timesteps.numel() * es and off % es == 0:
                    self._step_index = off // es
                    return
            self._step_index = self.index_for_timestep(timestep)
        else:
            self._step_index = self._begin_index

    def step(self, model_output, timestep, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0,
             generator=None, per_token_timesteps=None, return_dict=True, *, out=None):
        """edit_ppo/scheduler_fmppo.py:306-455."""
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None'. Call 'set_timesteps' first.")
        if isinstance(timestep, int) or isinstance(timestep, (torch.IntTensor, torch.LongTensor)) or (
                isinstance(timestep, torch.Tensor) and not timestep.is_floating_point()):
            raise ValueError("Passing integer indices as timesteps to `step()` is not supported. "
                             "Pass one of `scheduler.timesteps`.")
        if per_token_timesteps is not None:
            raise NotImplementedError("per_token_timesteps is not used by the reference drivers")
        if self.config.scaler_dim > 2:
            raise NotImplementedError("More than two scale parameters not supported.")
        if self._step_index is None:
            self._init_step_index(timestep)
        L.require_cuda(model_output, "model_output")
        L.require_cuda(sample, "sample")
        model_output = model_output.contiguous()
        # the reference upcasts the SAMPLE to fp32 (scheduler_fmppo.py:354) and rounds only the result (:435-436):
        # an fp32 sample is consumed as fp32 by the kernel; a sample already in the model dtype is exact either way
        sample = sample.contiguous()
        x_is_f32 = (sample.dtype == torch.float32 and model_output.dtype != torch.float32)
        if not x_is_f32 and sample.dtype != model_output.dtype:
            sample = sample.to(model_output.dtype)
        dev = model_output.device
        B = model_output.shape[0]
        self.ets.append(model_output)
        self.ets = self.ets[-self.config.order_dim:]
        m = len(self.ets)
        i = self._step_index
        cur, nxt = self._sigmas[i], self._sigmas[i + 1]
        dt = np.float32(nxt - cur)

        if self._cond_dev is None or self._cond_dev[0].device != dev or self._cond_dev[1] != model_output.dtype:
            host = np.stack([self._sigmas[:-1], self._sigmas[1:]], 1).astype(np.float32)
            # sigmas rounded through the model dtype (bf16) like scheduler_fmppo.py:383
            self._cond_dev = (torch.from_numpy(host).to(dev).to(model_output.dtype).to(torch.float32),
                              model_output.dtype)
        cond_row = self._cond_dev[0][i:i + 1]
        net = self.factor_net.module if hasattr(self.factor_net, "module") else self.factor_net
        probs3, actions, aprobs, _ = self._policy(cond_row, B, dev)
        masks = self._masks(B, net.action_dims, m, dev)

        prev = out if out is not None else torch.empty_like(model_output)
        a = L.CsStepArgs()
        self._fill_step_args(a, sample, model_output, None, 1.0, actions, prev, None, model_output.dtype)
        a.dt = float(dt)
        if x_is_f32:
            a.io_dtype, a.x_is_f32 = L.dtype_code(model_output.dtype), 1
        L.check(L.lib().cs_lms_euler_step(C.byref(a), L.stream_ptr(dev)))
        self._step_index += 1

        conds = {"x": cond_row.to(model_output.dtype).repeat(B, 1),
                 "epsilon": self._stack(B, a.elems, model_output) if (self.record_conds or net.use_conv) else None}
        if self.verbose:
            print(f"T={float(cur) * 1000:.2f} -> {float(nxt) * 1000:.2f} | actions: {actions[0].tolist()}")
        if not return_dict:
            return (prev, actions, aprobs, conds, masks)
        return SolverOutput(prev_sample=prev, actions=actions, probs=aprobs, conds=conds, masks=masks)

    def scale_noise(self, sample, timestep, noise=None):
        """scheduler_fmppo.py:457-484 (forward process; not on the sampling path)."""
        sigmas = self.sigmas.to(device=sample.device, dtype=sample.dtype)
        ts = self.timesteps.to(sample.device)
        timestep = timestep.to(sample.device)
        if self._begin_index is None:
            idx = [self.index_for_timestep(t, ts) for t in timestep]
        elif self._step_index is not None:
            idx = [self._step_index] * timestep.shape[0]
        else:
            idx = [self._begin_index] * timestep.shape[0]
        sigma = sigmas[idx].flatten()
        while len(sigma.shape) < len(sample.shape):
            sigma = sigma.unsqueeze(-1)
        return sigma * noise + (1.0 - sigma) * sample
