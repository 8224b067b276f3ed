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
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)   # flip_sin_to_cos=True


class UNetOracle:
    def __init__(self, sd, config, round_weights_to_f16=True):
        self.cfg = config
        self.sd = {k: (v.half().float() if round_weights_to_f16 else v.float()) for k, v in sd.items()}

    def _gn(self, x, p, eps, silu):
        y = F.group_norm(x, self.cfg["norm_num_groups"], self.sd[p + ".weight"], self.sd[p + ".bias"], eps)
        return F.silu(y) if silu else y

    def _resnet(self, x, temb_silu, p):
        sd = self.sd
        h = self._gn(x, p + ".norm1", 1e-5, True)
        h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
        t = F.linear(temb_silu, sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
        h = h + t[:, :, None, None]
        h = self._gn(h, p + ".norm2", 1e-5, True)
        h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
        if (p + ".conv_shortcut.weight") in sd:
            x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
        return x + h

    def _attn(self, x, ctx, p):
        sd, H = self.sd, self.cfg["num_heads"]
        q = F.linear(x, sd[p + ".to_q.weight"])
        k = F.linear(ctx, sd[p + ".to_k.weight"])
        v = F.linear(ctx, sd[p + ".to_v.weight"])
        B, N, C = q.shape
        dh = C // H
        q = q.view(B, N, H, dh).transpose(1, 2)
        k = k.view(B, -1, H, dh).transpose(1, 2)
        v = v.view(B, -1, H, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, N, C)
        return F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])

    def _xformer(self, x, ctx, p):
        sd = self.sd
        B, C, H, W = x.shape
        res = x
        h = self._gn(x, p + ".norm", 1e-6, False)
        h = F.conv2d(h, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        t = p + ".transformer_blocks.0"
        n = F.layer_norm(h, (C,), sd[t + ".norm1.weight"], sd[t + ".norm1.bias"])
        h = h + self._attn(n, n, t + ".attn1")
        n = F.layer_norm(h, (C,), sd[t + ".norm2.weight"], sd[t + ".norm2.bias"])
        h = h + self._attn(n, ctx, t + ".attn2")
        n = F.layer_norm(h, (C,), sd[t + ".norm3.weight"], sd[t + ".norm3.bias"])
        pr = F.linear(n, sd[t + ".ff.net.0.proj.weight"], sd[t + ".ff.net.0.proj.bias"])
        val, gate = pr.chunk(2, dim=-1)
        h = h + F.linear(val * F.gelu(gate), sd[t + ".ff.net.2.weight"], sd[t + ".ff.net.2.bias"])
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
        return h + res

    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states, round_activations_to_f16=False):
        cfg, sd = self.cfg, self.sd
        x = sample.float()
        ctx = encoder_hidden_states.float()
        B = x.shape[0]
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        c0 = cfg["block_out_channels"][0]
        emb = timestep_embedding(t, c0)
        emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
        emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
        ts = F.silu(emb)
        h = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
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
                h = F.conv2d(h, sd[f"{b}.downsamplers.0.conv.weight"], sd[f"{b}.downsamplers.0.conv.bias"], stride=2, padding=1)
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
                h = F.conv2d(h, sd[f"{b}.upsamplers.0.conv.weight"], sd[f"{b}.upsamplers.0.conv.bias"], padding=1)
        h = self._gn(h, "conv_norm_out", 1e-5, True)
        return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
