"""FactorNetPPO -- the ConsistencySolver policy network, HIP-backed.

Mirrors the reference interface (same constructor arguments, same state-dict
keys ``mlp.{0,2,4}.{weight,bias}`` + buffer ``action_values``, same methods
``sample_action`` / ``get_action_probs`` / ``forward``):

* SD variant   : factor_net_ppo.py:57-184
* FLUX variant : edit_ppo/factor_net_ppo.py:57-196 (``mu_dim``, softmax(logits/0.01),
  no /999 input normalisation, default (non-zero) init of the last layer)

The MLP + softmax, the cosine features, the categorical gather and the PPO
re-evaluation run as HIP kernels behind the C ABI (``cs_factor_probs``,
``cs_cosine_features``, ``cs_gather_actions`` / ``cs_sample_actions``,
``cs_action_probs``).  Tensors must live on the GPU; there is no CPU path.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from .tables import torch_linspace_f32


def _grid_sd(order_dim, scaler_dim, K):
    rows = []
    for i in range(order_dim + scaler_dim - 1):
        if i == 0:
            rows.append(torch_linspace_f32(0, 2, K))
        elif i == 1:
            rows.append(torch_linspace_f32(-2, 0, K))
        elif i < order_dim - 1:
            rows.append(torch_linspace_f32(-1, 1, K))
        else:
            rows.append(torch_linspace_f32(-0.05, 0.05, K))
    return np.stack(rows) if rows else np.zeros((0, K), np.float32)


def _grid_flux(order_dim, scaler_dim, mu_dim, K):
    rows = []
    mu_vals = np.concatenate([np.zeros(1, np.float32), torch_linspace_f32(0.5, 0.99, K - 1)])
    for i in range(order_dim + scaler_dim + mu_dim - 1):
        if i == 0:
            rows.append(torch_linspace_f32(0, 1, K))
        elif i == 1 and i < order_dim - 1:
            rows.append(torch_linspace_f32(-2, 0, K))
        elif i < order_dim - 1:
            rows.append(torch_linspace_f32(-1, 1, K))
        elif i < order_dim + scaler_dim - 1:
            rows.append(torch_linspace_f32(-0.05, 0.05, K))
        else:
            rows.append(mu_vals)
    return np.stack(rows) if rows else np.zeros((0, K), np.float32)


class FactorNetPPO(nn.Module):
    """SD-side policy net (factor_net_ppo.py:57).  ``embedding_dim``,
    ``input_channels`` and ``conv_out_channels`` are accepted and ignored exactly
    like the reference; ``use_conv=True`` means "append cosine-similarity
    features of the eps history" (factor_net_ppo.py:72-73,146-149)."""

    variant = "sd"
    input_scale = 1.0 / 999.0
    inv_temperature = 1.0

    def __init__(self, embedding_dim=1024, hidden_dim=256, num_actions=161, order_dim=4, scaler_dim=2,
                 use_conv=False, input_channels=4, conv_out_channels=8, **_ignored):
        super().__init__()
        self.num_actions = num_actions
        self.order_dim = order_dim
        self.scaler_dim = scaler_dim
        self.action_dims = self._action_dims()
        self.use_conv = use_conv
        mlp_in = 2 + ((order_dim - 1) if use_conv else 0)
        self.mlp = nn.Sequential(
            nn.Linear(mlp_in, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
            nn.Linear(hidden_dim, num_actions * self.action_dims))
        self._init_last_layer()
        self.register_buffer("action_values", torch.from_numpy(self._grid()))
        # replay hook: when set ([B, A] int64), sample_action gathers these indices instead of drawing
        self.forced_action_idx = None
        # "inverse_cdf" (default) = one torch.rand + the HIP inverse-CDF kernel (cs_sample_actions): 2 launches per step, graph
        # capturable, eager == graph; "multinomial" = torch.multinomial on the default generator, the reference's RNG consumer
        # (factor_net_ppo.py:161; ~15 small launches + a device assert per step).  Neither reproduces the reference's CUDA random
        # stream on ROCm -- identical-seed parity is defined on replayed indices (forced_action_idx), SURVEY 7.3.
        self.sampler = "inverse_cdf"
        self._w32 = None
        self._w32_key = None

    # -- variant hooks ---------------------------------------------------------
    def _action_dims(self):
        return self.order_dim + self.scaler_dim - 1

    def _grid(self):
        return _grid_sd(self.order_dim, self.scaler_dim, self.num_actions)

    def _init_last_layer(self):
        nn.init.zeros_(self.mlp[-1].bias)      # factor_net_ppo.py:82-83
        nn.init.zeros_(self.mlp[-1].weight)

    # -- weights as fp32 device arrays (cached until a parameter changes) --------
    def _weights32(self):
        ps = [self.mlp[0].weight, self.mlp[0].bias, self.mlp[2].weight, self.mlp[2].bias,
              self.mlp[4].weight, self.mlp[4].bias, self.action_values]
        key = tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) for p in ps)
        if key != self._w32_key:
            for p in ps:
                L.require_cuda(p, "factor_net parameter (call .to('cuda') first)")
            self._w32 = [p.detach().to(torch.float32).contiguous() for p in ps]
            self._w32_key = key
        return self._w32

    def _net_struct(self):
        w = self._weights32()
        return L.CsFactorNet(w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), w[3].data_ptr(),
                             w[4].data_ptr(), w[5].data_ptr(), self.mlp[0].in_features, self.mlp[0].out_features,
                             self.action_dims, self.num_actions, self.input_scale, self.inv_temperature)

    # -- kernels ---------------------------------------------------------------
    def cosine_features(self, hist, m=None, cfg=None):
        """hist: list of [B, ...] tensors newest first (len m) -> [B, order-1] fp32.

        ``cfg=(eps_uncond, guidance, eps_out)``: ``hist[0]`` is the TEXT branch of a CFG dual batch; the newest entry
        ``u + g (c - u)`` is formed inside the kernel (rounded as the update kernel rounds it) and written to ``eps_out``."""
        m = len(hist) if m is None else m
        e0 = L.require_cuda(hist[0], "epsilon")
        B = e0.shape[0]
        elems = e0.numel() // max(B, 1)
        out = torch.empty(B, self.order_dim - 1, dtype=torch.float32, device=e0.device)
        hs = [h.contiguous() for h in hist[:m]]
        arr = (C.c_void_p * L.CS_MAX_ORDER)(*[h.data_ptr() for h in hs])
        if cfg is None:
            L.check(L.lib().cs_cosine_features(arr, m, self.order_dim, B, elems, L.dtype_code(e0.dtype),
                                               L.ptr(out), L.stream_ptr(e0.device)))
        else:
            eu, g, eps_out = cfg
            L.require_cuda(eu, "eps_uncond"), L.require_cuda(eps_out, "eps_out")
            if not (eu.is_contiguous() and eps_out.is_contiguous()) or eu.dtype != e0.dtype or eps_out.dtype != e0.dtype:
                raise ValueError("eps_uncond / eps_out must be contiguous tensors of the model-output dtype")
            L.check(L.lib().cs_cosine_features_cfg(arr, m, self.order_dim, B, elems, L.dtype_code(e0.dtype), L.ptr(eu), float(g),
                                                   L.ptr(eps_out), L.ptr(out), L.stream_ptr(e0.device)))
        return out

    def probs_from(self, x, hist=None, m=None, batch=None, cfg=None):
        """x: [B, 2] or [1, 2] (broadcast) conditioning; hist: eps history newest first (use_conv);
        cfg: see ``cosine_features`` (use_conv under classifier-free guidance)."""
        x = L.require_cuda(x, "conds['x']").to(torch.float32).contiguous()
        B = batch if batch is not None else x.shape[0]
        stride = 0 if (x.shape[0] == 1 and B != 1) else x.shape[1]
        cosf = None
        if self.use_conv:
            if hist is None:
                raise ValueError("use_conv=True requires the epsilon history")
            cosf = self.cosine_features(hist, m, cfg)
        probs = torch.empty(B, self.action_dims, self.num_actions, dtype=torch.float32, device=x.device)
        net = self._net_struct()
        L.check(L.lib().cs_factor_probs(C.byref(net), L.ptr(x), stride, L.ptr(cosf), B, L.ptr(probs),
                                        L.stream_ptr(x.device)))
        return probs

    def forward_(self, x_dict):
        """factor_net_ppo.py:137-157 -> probs [B, A, K]."""
        eps = x_dict.get("epsilon", None)
        hist = None
        if self.use_conv:
            if eps is None:
                raise ValueError("use_conv=True requires x_dict['epsilon']")
            hist = [eps[:, k] for k in range(self.order_dim)]
        return self.probs_from(x_dict["x"], hist, self.order_dim if hist else None)

    def draw(self, probs):
        """probs [B,A,K] -> (actions [B,A], action_probs [B,A], idx [B,A])."""
        B, A, K = probs.shape
        av = self._weights32()[6]
        actions = torch.empty(B, A, dtype=torch.float32, device=probs.device)
        aprobs = torch.empty_like(actions)
        lib, st = L.lib(), L.stream_ptr(probs.device)
        if self.forced_action_idx is not None:
            forced = self.forced_action_idx
            if isinstance(forced, list):                      # a replay queue: one [B, A] index tensor per step
                if not forced:
                    raise RuntimeError("forced_action_idx queue is empty")
                forced = forced.pop(0)
            idx = forced.to(device=probs.device, dtype=torch.int64).reshape(B, A).contiguous()
            L.check(lib.cs_gather_actions(L.ptr(probs), L.ptr(idx), L.ptr(av), B, A, K, L.ptr(actions), L.ptr(aprobs), st))
        elif self.sampler == "multinomial":
            idx = torch.multinomial(probs.view(-1, K), num_samples=1).view(B, A)
            L.check(lib.cs_gather_actions(L.ptr(probs), L.ptr(idx), L.ptr(av), B, A, K, L.ptr(actions), L.ptr(aprobs), st))
        elif self.sampler == "inverse_cdf":
            u = torch.rand(B, A, dtype=torch.float32, device=probs.device)
            idx = torch.empty(B, A, dtype=torch.int64, device=probs.device)
            L.check(lib.cs_sample_actions(L.ptr(probs), L.ptr(u), L.ptr(av), B, A, K, L.ptr(idx), L.ptr(actions), L.ptr(aprobs), st))
        else:
            raise ValueError(f"unknown sampler {self.sampler!r}")
        return actions, aprobs, idx

    def sample_action(self, x_dict):
        """factor_net_ppo.py:159-168 -> (sampled_actions [B,A], action_probs [B,A])."""
        actions, aprobs, _ = self.draw(self.forward_(x_dict))
        return actions, aprobs

    def get_action_probs(self, x_dict, actions):
        """factor_net_ppo.py:170-184 -> (selected_probs [B,A], normalised entropy [B,A])."""
        probs = self.forward_(x_dict)
        B, A, K = probs.shape
        actions = L.require_cuda(actions.to(probs.device), "actions").to(torch.float32).contiguous()
        sel = torch.empty(B, A, dtype=torch.float32, device=probs.device)
        ent = torch.empty_like(sel)
        L.check(L.lib().cs_action_probs(L.ptr(probs), L.ptr(actions), L.ptr(self._weights32()[6]), B, A, K,
                                        L.ptr(sel), L.ptr(ent), L.stream_ptr(probs.device)))
        return sel, ent

    def forward(self, x_dict, actions=None):
        if actions is None:
            return self.sample_action(x_dict)
        return self.get_action_probs(x_dict, actions)


class FluxFactorNetPPO(FactorNetPPO):
    """FLUX-side policy net (edit_ppo/factor_net_ppo.py:57)."""

    variant = "flux"
    input_scale = 1.0       # normalize_input is the identity (:112-114)
    inv_temperature = 100.0  # softmax(logits / 0.01) (:168)

    def __init__(self, embedding_dim=1024, hidden_dim=256, num_actions=161, order_dim=4, scaler_dim=2, mu_dim=1,
                 use_conv=False, input_channels=4, conv_out_channels=8, **_ignored):
        self.mu_dim = mu_dim
        super().__init__(embedding_dim, hidden_dim, num_actions, order_dim, scaler_dim, use_conv)

    def _action_dims(self):
        return self.order_dim + self.scaler_dim + self.mu_dim - 1

    def _grid(self):
        return _grid_flux(self.order_dim, self.scaler_dim, self.mu_dim, self.num_actions)

    def _init_last_layer(self):
        pass  # default nn.Linear init (zero-init is commented out upstream, :87-88)
