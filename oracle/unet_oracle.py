"""CPU restatement (plain torch ops, fp32) of the SD1.5 ``UNet2DConditionModel`` forward.

TEST INFRASTRUCTURE ONLY (imported by tests/, smoke() and bench.py's cpu_baseline leg).

PARITY UNPINNED: the denoiser's arithmetic lives in the un-vendored third-party package
diffusers==0.26.3 (env.yaml:52) which is absent from /root/reference and from this image, and the
reference has no tests or golden vectors for it.  This file restates the public
SD1.5 architecture (block widths 320/640/1280/1280, 2 layers per block, 8 heads,
cross dim 768, GroupNorm(32) eps 1e-5 in resnets / 1e-6 in transformers, GEGLU feed-forward,
sinusoidal timestep embedding with flip_sin_to_cos and shift 0, nearest x2 upsample, stride-2
downsample conv, skip concat) with the diffusers state-dict key names, anchored on the reference's
call sites (denoise_ppo.py:89-94, gen_pretrain/pipeline.py:1058-1066).  It is the checker for the
HIP UNet kernels, not a pinned copy of diffusers.
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)   # flip_sin_to_cos=True


class UNetOracle:
    def __init__(self, sd, config, round_weights_to_f16=True, device="cpu", dtype=torch.float32):
        """device / dtype: the default (CPU, fp32) is the oracle.  Tests also instantiate the SAME restatement as a
        plain torch fp16 graph on the GPU (device="cuda", dtype=torch.float16: fp16 weights, fp16 activation storage,
        the vendor libraries' fp32 accumulation) -- the arithmetic class of the reference's own fp16 pipeline
        (gen_ppo.py:193-195) -- to separate "intrinsic to fp16" from "our kernels" (tests/test_parity_e2e_gpu.py)."""
        self.cfg = config
        self.device, self.dtype = torch.device(device), dtype
        self.sd = {k: (v.half().float() if round_weights_to_f16 else v.float()).to(self.device, dtype) for k, v in sd.items()}
        # op hooks: tests may replace ONE op class of the torch graph by the HIP kernel of that class to attribute the
        # end-to-end error to kernel classes (tests/test_parity_e2e_gpu.py); the defaults are the plain torch ops.
        self.ops = dict(conv3=None, conv1=None, linear=None, geglu=None, sdpa=None, group_norm=None, layer_norm=None)

    def _conv(self, x, key, stride=1, padding=0):
        w, b = self.sd[key + ".weight"], self.sd.get(key + ".bias")
        hook = self.ops["conv3" if w.shape[-1] == 3 else "conv1"]
        if hook is not None:
            return hook(x, w, b, stride)
        return F.conv2d(x, w, b, stride=stride, padding=padding)

    def _linear(self, x, key, bias=True):
        w, b = self.sd[key + ".weight"], (self.sd[key + ".bias"] if bias else None)
        if self.ops["linear"] is not None and x.dim() == 3:
            return self.ops["linear"](x, w, b)
        return F.linear(x, w, b)

    def _ln(self, x, key):
        w, b = self.sd[key + ".weight"], self.sd[key + ".bias"]
        if self.ops["layer_norm"] is not None:
            return self.ops["layer_norm"](x, w, b)
        return F.layer_norm(x, (x.shape[-1],), w, b)

    def _gn(self, x, p, eps, silu):
        if self.ops["group_norm"] is not None:
            return self.ops["group_norm"](x, self.sd[p + ".weight"], self.sd[p + ".bias"], self.cfg["norm_num_groups"], eps, silu)
        y = F.group_norm(x, self.cfg["norm_num_groups"], self.sd[p + ".weight"], self.sd[p + ".bias"], eps)
        return F.silu(y) if silu else y

    def _resnet(self, x, temb_silu, p):
        sd = self.sd
        h = self._gn(x, p + ".norm1", 1e-5, True)
        h = self._conv(h, p + ".conv1", padding=1)
        t = F.linear(temb_silu, sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
        h = h + t[:, :, None, None]
        h = self._gn(h, p + ".norm2", 1e-5, True)
        h = self._conv(h, p + ".conv2", padding=1)
        if (p + ".conv_shortcut.weight") in sd:
            x = self._conv(x, p + ".conv_shortcut")
        return x + h

    def _attn(self, x, ctx, p):
        sd, H = self.sd, self.cfg["num_heads"]
        q = self._linear(x, p + ".to_q", bias=False)
        k = self._linear(ctx, p + ".to_k", bias=False)
        v = self._linear(ctx, p + ".to_v", bias=False)
        B, N, C = q.shape
        dh = C // H
        if self.ops["sdpa"] is not None:
            a = self.ops["sdpa"](q, k, v, H)
        else:
            q = q.view(B, N, H, dh).transpose(1, 2)
            k = k.view(B, -1, H, dh).transpose(1, 2)
            v = v.view(B, -1, H, dh).transpose(1, 2)
            a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v
            a = a.transpose(1, 2).reshape(B, N, C)
        return self._linear(a, p + ".to_out.0")

    def _xformer(self, x, ctx, p):
        sd = self.sd
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, p + ".norm", 1e-6, False)
        h = self._conv(h, p + ".proj_in")
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        t = p + ".transformer_blocks.0"
        n = self._ln(h, t + ".norm1")
        h = h + self._attn(n, n, t + ".attn1")
        n = self._ln(h, t + ".norm2")
        h = h + self._attn(n, ctx, t + ".attn2")
        n = self._ln(h, t + ".norm3")
        if self.ops["geglu"] is not None:
            ff = self.ops["geglu"](n, sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
        else:
            pr = F.linear(n, sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
            val, gate = pr.chunk(2, dim=-1)
            ff = val * F.gelu(gate)
        h = h + self._linear(ff, t + ".ff.net.2")
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self._conv(h, p + ".proj_out")
        return h + res

    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states, round_activations_to_f16=False):
        cfg, sd = self.cfg, self.sd
        x = sample.to(self.device, self.dtype)
        ctx = encoder_hidden_states.to(self.device, self.dtype)
        B = x.shape[0]
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1).to(self.device)
        if t.numel() == 1:
            t = t.expand(B)
        c0 = cfg["block_out_channels"][0]
        emb = timestep_embedding(t, c0).to(self.dtype)      # sinusoid in fp32, cast to the model dtype like diffusers
        emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
        emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
        ts = F.silu(emb)
        h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)      # 4 input channels: not a GEMM-class layer
        skips = [h]
        nres = cfg["layers_per_block"]
        for i in range(4):
            b = f"down_blocks.{i}"
            for j in range(nres):
                h = self._resnet(h, ts, f"{b}.resnets.{j}")
                if cfg["down_has_attn"][i]:
                    h = self._xformer(h, ctx, f"{b}.attentions.{j}")
                skips.append(h)
            if i < 3:
                h = self._conv(h, f"{b}.downsamplers.0.conv", stride=2, padding=1)
                skips.append(h)
        h = self._resnet(h, ts, "mid_block.resnets.0")
        h = self._xformer(h, ctx, "mid_block.attentions.0")
        h = self._resnet(h, ts, "mid_block.resnets.1")
        for i in range(4):
            b = f"up_blocks.{i}"
            for j in range(nres + 1):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self._resnet(h, ts, f"{b}.resnets.{j}")
                if cfg["up_has_attn"][i]:
                    h = self._xformer(h, ctx, f"{b}.attentions.{j}")
            if i < 3:
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = self._conv(h, f"{b}.upsamplers.0.conv", padding=1)
        h = self._gn(h, "conv_norm_out", 1e-5, True)
        return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
