"""Host-side schedule tables (numpy; integer grids are bit-exact with the reference).

* beta / alphas_cumprod tables : scheduler_ppo.py:99-114
* SD timestep grids            : scheduler_ppo.py:142-163
* FLUX sigma schedule          : edit_ppo/scheduler_fmppo.py:171-245, edit_ppo/pipeline.py:119-129

These run once per ``set_timesteps`` on the host; the per-step device work is in
the HIP kernels.
"""
import math

import numpy as np

F32 = np.float32


def torch_linspace_f32(start, end, steps):
    """fp32 ``torch.linspace``: first half ``start + step*i``, second half
    ``end - step*(n-1-i)`` (ATen RangeFactories)."""
    s, e = F32(start), F32(end)
    if steps == 1:
        return np.array([s], dtype=F32)
    step = F32((e - s) / F32(steps - 1))
    idx = np.arange(steps)
    half = steps // 2
    out = np.empty(steps, F32)
    out[:half] = s + step * idx[:half].astype(F32)
    out[half:] = e - step * (steps - 1 - idx[half:]).astype(F32)
    return out


def make_betas(beta_schedule, beta_start, beta_end, T, trained_betas=None):
    if trained_betas is not None:
        return np.asarray(trained_betas, dtype=F32)
    if beta_schedule == "linear":
        return torch_linspace_f32(beta_start, beta_end, T)
    if beta_schedule == "scaled_linear":
        r = torch_linspace_f32(beta_start ** 0.5, beta_end ** 0.5, T)
        return (r * r).astype(F32)
    if beta_schedule == "squaredcos_cap_v2":
        def bar(t):
            return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.asarray([min(1 - bar((i + 1) / T) / bar(i / T), 0.999) for i in range(T)], dtype=F32)
    raise NotImplementedError(f"{beta_schedule} schedule not implemented.")


def alphas_cumprod(betas):
    return np.cumprod((F32(1.0) - betas).astype(F32), dtype=F32)


def sd_timestep_grid(n, T, spacing, steps_offset):
    if spacing == "linspace":
        return np.linspace(0, T - 1, n).round()[::-1].copy().astype(np.int64)
    if spacing == "leading":
        ts = (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64)
        return ts + steps_offset
    if spacing == "trailing":
        return np.round(np.arange(T, 0, -(T / n))).astype(np.int64) - 1
    raise ValueError(f"Unsupported timestep_spacing: {spacing}.")


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.15):
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * m + (base_shift - m * base_seq_len)


def flux_sigma_schedule(cfg, shift, sigma_min, sigma_max, num_inference_steps=None, sigmas=None, mu=None,
                        timesteps=None):
    """Returns (sigmas fp32 [n+1], timesteps fp32 [n]).  ``cfg`` is the scheduler config."""
    T = cfg.num_train_timesteps
    if cfg.use_dynamic_shifting and mu is None:
        raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
    if sigmas is not None and timesteps is not None and len(sigmas) != len(timesteps):
        raise ValueError("`sigmas` and `timesteps` should have the same length")
    if num_inference_steps is not None:
        if (sigmas is not None and len(sigmas) != num_inference_steps) or (
                timesteps is not None and len(timesteps) != num_inference_steps):
            raise ValueError("`sigmas` and `timesteps` should have the same length as num_inference_steps, "
                             "if `num_inference_steps` is provided")
    else:
        num_inference_steps = len(sigmas) if sigmas is not None else len(timesteps)
    provided = timesteps is not None
    if provided:
        timesteps = np.array(timesteps).astype(F32)
    if sigmas is None:
        if timesteps is None:
            timesteps = np.linspace(sigma_max * T, sigma_min * T, num_inference_steps)
        sig = timesteps / T
    else:
        sig = np.array(sigmas).astype(F32)
    if cfg.use_dynamic_shifting:
        if cfg.time_shift_type == "exponential":
            sig = math.exp(mu) / (math.exp(mu) + (1 / sig - 1) ** 1.0)
        else:
            sig = mu / (mu + (1 / sig - 1) ** 1.0)
    else:
        sig = shift * sig / (1 + (shift - 1) * sig)
    if cfg.shift_terminal:
        omz = 1 - sig
        sig = 1 - (omz / (omz[-1] / (1 - cfg.shift_terminal)))
    if cfg.use_karras_sigmas or cfg.use_exponential_sigmas or cfg.use_beta_sigmas:
        smin, smax = float(sig[-1]), float(sig[0])
        if cfg.use_karras_sigmas:
            ramp = np.linspace(0, 1, num_inference_steps)
            a, b = smin ** (1 / 7.0), smax ** (1 / 7.0)
            sig = (b + ramp * (a - b)) ** 7.0
        elif cfg.use_exponential_sigmas:
            sig = np.exp(np.linspace(math.log(smax), math.log(smin), num_inference_steps))
        else:
            import scipy.stats
            sig = np.array([smin + (p * (smax - smin)) for p in
                            [scipy.stats.beta.ppf(t, 0.6, 0.6) for t in 1 - np.linspace(0, 1, num_inference_steps)]])
    sig = np.asarray(sig).astype(F32)
    ts = timesteps.astype(F32) if provided else (sig * F32(T)).astype(F32)
    if cfg.invert_sigmas:
        sig = (F32(1.0) - sig).astype(F32)
        ts = (sig * F32(T)).astype(F32)
        sig = np.concatenate([sig, np.ones(1, F32)])
    else:
        sig = np.concatenate([sig, np.zeros(1, F32)])
    return sig, ts
