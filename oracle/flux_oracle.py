"""CPU restatement (plain torch ops, fp32) of the FLUX DiT (``FluxTransformer2DModel``) forward.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the arithmetic lives in diffusers (git main, unpinned,
readme.md:234) which is not under /root/reference nor installed, and the reference holds no test or vector
for it.  This follows the public FLUX.1 architecture with the diffusers state-dict key
names, anchored on the call sites edit_ppo/pipeline.py:1087-1097 and edit_ppo/denoise_diffusion.py:135-144.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def sinusoid(t, dim=256):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], -1)          # flip_sin_to_cos=True


def rope_tables(ids, axes_dims, theta=10000.0):
    cos, sin = [], []
    ids = np.asarray(ids, np.float64)
    for i, d in enumerate(axes_dims):
        freqs = 1.0 / (theta ** (np.arange(0, d, 2, dtype=np.float64)[: d // 2] / d))
        ang = np.outer(ids[:, i], freqs)
        cos.append(np.cos(ang)); sin.append(np.sin(ang))
    cos = torch.from_numpy(np.concatenate(cos, 1)).float().repeat_interleave(2, dim=1)
    sin = torch.from_numpy(np.concatenate(sin, 1)).float().repeat_interleave(2, dim=1)
    return cos, sin                                              # [S, head_dim]


def apply_rope(x, cos, sin):                                    # x [B, H, S, D]; fp32 tables, result in x's dtype (as diffusers' apply_rotary_emb)
    xf = x.float()
    xr, xi = xf.reshape(*xf.shape[:-1], -1, 2).unbind(-1)
    rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (xf * cos[None, None] + rot * sin[None, None]).to(x.dtype)


class _Fp32View:
    """mapping view that converts a tensor to CPU fp32 when it is READ: lets the oracle walk a model whose fp32 weights do not fit in host memory
    (full-depth FLUX: 11.9 B parameters = 48 GB) -- the source mapping may hold them in 16 bits and / or on another device; one weight at a time is live."""

    def __init__(self, sd, device="cpu", dtype=torch.float32):
        self._sd, self._device, self._dtype = sd, device, dtype

    def __getitem__(self, k):
        return self._sd[k].detach().to(self._device, self._dtype)


class FluxOracle:
    def __init__(self, sd, config, lazy=False, device="cpu", dtype=torch.float32):
        """lazy=True: `sd` is read through on demand (see _Fp32View) instead of being converted to fp32 up front.
        device / dtype: the default (CPU, fp32) is the oracle.  Tests also run the SAME restatement as a plain torch bf16 graph on the GPU (device="cuda",
        dtype=torch.bfloat16: bf16 weights, bf16 activation storage between torch ops, the vendor kernels' fp32 accumulation; sinusoids, RoPE and the RMSNorm
        statistics in fp32 as diffusers computes them) -- the arithmetic class of the reference's own bf16 pipeline (edit_ppo/generate_ours.py:120-126) -- so that
        the HIP DiT's distance from the fp32 oracle can be read against what that class delivers (tests/test_flux_gpu.py)."""
        self.cfg = config
        self.device, self.dtype = torch.device(device), dtype
        self.sd = _Fp32View(sd, self.device, dtype) if lazy else {k: v.to(self.device, dtype) for k, v in sd.items()}
        self.D = config["num_heads"] * config["head_dim"]
        # precision emulation (tools/sim_precision_flux.py): rs rounds the residual STREAM after every update, rb the BRANCH tensors (modulated LayerNorm outputs,
        # q / k / v, attention output, MLP activations, the linear outputs that are added onto the stream) -- identity in the oracle
        self.rs = self.rb = (lambda t: t)
        # (round 6) rp rounds the softmax probabilities in front of the P V product, rh the two tensors of the output head (the modulated LayerNorm output that
        # feeds proj_out, and proj_out's output) -- the places where the HIP DiT stores the model dtype and `rb` does not reach
        self.rp = self.rh = (lambda t: t)

    def _in(self, x):
        return torch.as_tensor(x).to(self.device, self.dtype)

    def lin(self, x, p):
        return F.linear(x, self.sd[p + ".weight"], self.sd[p + ".bias"])

    def rms(self, x, p):
        xf = x.float()                                               # (diffusers' RMSNorm takes the variance in fp32)
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(x.dtype) * self.sd[p + ".weight"]

    def heads(self, x):
        B, S, _ = x.shape
        return x.view(B, S, self.cfg["num_heads"], self.cfg["head_dim"]).transpose(1, 2)

    def attn(self, q, k, v):
        if q.shape[2] * k.shape[2] * q.shape[1] > (1 << 28):        # long sequences: one head at a time (S = 8704: 303 MB of scores per head instead of 7.3 GB)
            a = torch.cat([self.rp(torch.softmax(q[:, h:h + 1] @ k[:, h:h + 1].transpose(-1, -2) * self.cfg["head_dim"] ** -0.5, dim=-1)) @ v[:, h:h + 1]
                           for h in range(q.shape[1])], dim=1)
        else:
            a = self.rp(torch.softmax(q @ k.transpose(-1, -2) * self.cfg["head_dim"] ** -0.5, dim=-1)) @ v
        return a.transpose(1, 2).reshape(a.shape[0], a.shape[2], self.D)

    @torch.no_grad()
    def __call__(self, hidden_states, timestep, guidance, pooled, enc, txt_ids, img_ids):
        cfg, D = self.cfg, self.D
        x = self.rs(self.lin(self._in(hidden_states), "x_embedder"))
        c = self.rs(self.lin(self._in(enc), "context_embedder"))
        tt = "time_text_embed."
        temb = self.lin(F.silu(self.lin(self._in(sinusoid(timestep.float().cpu() * 1000)), tt + "timestep_embedder.linear_1")), tt + "timestep_embedder.linear_2")
        if cfg["guidance_embeds"]:
            temb = temb + self.lin(F.silu(self.lin(self._in(sinusoid(guidance.float().cpu() * 1000)), tt + "guidance_embedder.linear_1")), tt + "guidance_embedder.linear_2")
        temb = temb + self.lin(F.silu(self.lin(self._in(pooled), tt + "text_embedder.linear_1")), tt + "text_embedder.linear_2")
        ids = np.concatenate([np.asarray(txt_ids, np.float32), np.asarray(img_ids, np.float32)], 0)
        cos, sin = (t.to(self.device) for t in rope_tables(ids, cfg["axes_dims_rope"]))
        T = c.shape[1]
        silu_t = F.silu(temb)
        ln = lambda h: F.layer_norm(h, (D,), eps=1e-6)
        for i in range(cfg["num_layers"]):
            b = f"transformer_blocks.{i}"
            m = self.lin(silu_t, b + ".norm1.linear")[:, None].chunk(6, dim=-1)
            mc = self.lin(silu_t, b + ".norm1_context.linear")[:, None].chunk(6, dim=-1)
            rs, rb = self.rs, self.rb
            nx = rb(ln(x) * (1 + m[1]) + m[0])
            nc = rb(ln(c) * (1 + mc[1]) + mc[0])
            q = self.rms(self.heads(rb(self.lin(nx, b + ".attn.to_q"))), b + ".attn.norm_q")
            k = self.rms(self.heads(rb(self.lin(nx, b + ".attn.to_k"))), b + ".attn.norm_k")
            v = self.heads(rb(self.lin(nx, b + ".attn.to_v")))
            cq = self.rms(self.heads(rb(self.lin(nc, b + ".attn.add_q_proj"))), b + ".attn.norm_added_q")
            ck = self.rms(self.heads(rb(self.lin(nc, b + ".attn.add_k_proj"))), b + ".attn.norm_added_k")
            cv = self.heads(rb(self.lin(nc, b + ".attn.add_v_proj")))
            q, k, v = torch.cat([cq, q], 2), torch.cat([ck, k], 2), torch.cat([cv, v], 2)
            a = rb(self.attn(rb(apply_rope(q, cos, sin)), rb(apply_rope(k, cos, sin)), v))
            ca, xa = a[:, :T], a[:, T:]
            x = rs(x + m[2] * self.lin(xa, b + ".attn.to_out.0"))
            c = rs(c + mc[2] * self.lin(ca, b + ".attn.to_add_out"))
            nx = rb(ln(x) * (1 + m[4]) + m[3])
            x = rs(x + m[5] * self.lin(rb(F.gelu(self.lin(nx, b + ".ff.net.0.proj"), approximate="tanh")), b + ".ff.net.2"))
            nc = rb(ln(c) * (1 + mc[4]) + mc[3])
            c = rs(c + mc[5] * self.lin(rb(F.gelu(self.lin(nc, b + ".ff_context.net.0.proj"), approximate="tanh")), b + ".ff_context.net.2"))
        h = torch.cat([c, x], dim=1)
        for i in range(cfg["num_single_layers"]):
            b = f"single_transformer_blocks.{i}"
            m = self.lin(silu_t, b + ".norm.linear")[:, None].chunk(3, dim=-1)
            rs, rb = self.rs, self.rb
            n = rb(ln(h) * (1 + m[1]) + m[0])
            mlp = rb(F.gelu(self.lin(n, b + ".proj_mlp"), approximate="tanh"))
            q = self.rms(self.heads(rb(self.lin(n, b + ".attn.to_q"))), b + ".attn.norm_q")
            k = self.rms(self.heads(rb(self.lin(n, b + ".attn.to_k"))), b + ".attn.norm_k")
            v = self.heads(rb(self.lin(n, b + ".attn.to_v")))
            a = rb(self.attn(rb(apply_rope(q, cos, sin)), rb(apply_rope(k, cos, sin)), v))
            h = rs(h + m[2] * self.lin(torch.cat([a, mlp], dim=2), b + ".proj_out"))
        x = h[:, T:]
        sc, sh = self.lin(silu_t, "norm_out.linear")[:, None].chunk(2, dim=-1)
        x = self.rh(ln(x) * (1 + sc) + sh)
        return self.rh(self.lin(x, "proj_out"))
