"""CPU emulation of WHERE the HIP UNet executor rounds to fp16 (tools only; imports the oracle like a test does).

The fp32 oracle graph (oracle/unet_oracle.py) is re-walked with explicit rounding points that mirror
consolver_amd/csrc/unet.cpp: every tensor the executor stores in fp16 is rounded (x.half().float()), everything
it keeps in fp32 (accumulators, the fp32 epilogue patch) is not.  Switches select which STORES are fp16:

    stream   the residual stream (resnet outputs, transformer hidden, proj_out / down / upsample outputs, conv_in)
    raw      the fp16 operand a GEMM reads when it consumes the stream directly (shortcut 1x1, down / upsample conv,
             proj_out); only meaningful when stream is fp32 (otherwise the stream is already rounded)
    norm     GroupNorm / LayerNorm outputs (always the fp16 MFMA operand)
    branch   intra-branch tensors (conv1 output, qkv, attention output, GEGLU output)

Run:  python tools/sim_precision.py [t]      -> relative L2 of eps vs the fp32 oracle for a few configurations.
It answers "what does an fp32 residual stream buy" before any kernel is written.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.unet_oracle import UNetOracle, timestep_embedding  # noqa: E402
from consolver_amd.synth import synthetic_prompt_embeds, synthetic_unet_state_dict  # noqa: E402
from consolver_amd.unet import SD15_CONFIG  # noqa: E402


def r16(x):
    return x.half().float()


class Emu(UNetOracle):
    def __init__(self, sd, cfg, stream=True, raw=True, norm=True, branch=True, raw_sc=None, raw_po=None, raw_ud=None, ln_fold=False, head=True):
        super().__init__(sd, cfg)
        self.head = head          # False: conv_out reads the unrounded output of the final GroupNorm + SiLU (an operand kept as hi + lo)
        ident = lambda x: x
        self.rs = r16 if stream else ident
        self.rr = r16 if raw else ident
        self.rn = r16 if norm else ident
        self.rb = r16 if branch else ident
        # per consumer kind (None: follow `raw`): shortcut 1x1, proj_out, down / upsample conv
        self.rr_sc = self.rr if raw_sc is None else (r16 if raw_sc else ident)
        self.rr_po = self.rr if raw_po is None else (r16 if raw_po else ident)
        self.rr_ud = self.rr if raw_ud is None else (r16 if raw_ud else ident)
        self.ln_fold = ln_fold

    def _ln_linear(self, h, ln_key, lin_keys, bias=True):
        """LayerNorm -> Linear(s) as the executor would run it.  Unfolded: fp16(LN(h)) . W.  Folded (ln_fold): the GEMM multiplies the RAW fp16
        hidden state with W' = fp16(gamma * W) and the epilogue applies rstd (acc - mean s) + b' with s = rowsum(W'), b' = W beta + b in fp32."""
        sd = self.sd
        g, be = sd[ln_key + ".weight"], sd[ln_key + ".bias"]
        outs = []
        if not self.ln_fold:
            n = self.rn(self._ln(h, ln_key))
            for k in lin_keys:
                outs.append(F.linear(n, sd[k + ".weight"], sd.get(k + ".bias") if bias else None))
            return outs
        mean = h.mean(-1, keepdim=True)
        rstd = torch.rsqrt(h.var(-1, unbiased=False, keepdim=True) + 1e-5)
        hr = r16(h)
        for k in lin_keys:
            w = sd[k + ".weight"]
            wp = r16(w * g[None, :])
            s_ = wp.sum(1)
            bp = w @ be + (sd[k + ".bias"] if (bias and (k + ".bias") in sd) else 0.0)
            outs.append((hr @ wp.t() - mean * s_[None, None, :]) * rstd + bp)
        return outs

    def _resnet(self, x, temb_silu, p):
        sd = self.sd
        h = self.rn(self._gn(x, p + ".norm1", 1e-5, True))
        h = self._conv(h, p + ".conv1", padding=1)
        t = F.linear(temb_silu, sd[p + ".time_emb_proj.weight"], sd[p + ".time_emb_proj.bias"])
        h = self.rb(h + r16(t)[:, :, None, None])
        h = self.rn(self._gn(h, p + ".norm2", 1e-5, True))
        h = self._conv(h, p + ".conv2", padding=1)
        if (p + ".conv_shortcut.weight") in sd:
            x = self.rs(self._conv(self.rr_sc(x), p + ".conv_shortcut"))
        return self.rs(x + h)

    def _sdpa(self, q, k, v):
        H = self.cfg["num_heads"]
        B, N, C = q.shape
        dh = C // H
        q = q.view(B, N, H, dh).transpose(1, 2)
        k = k.view(B, -1, H, dh).transpose(1, 2)
        v = v.view(B, -1, H, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v
        return self.rb(a.transpose(1, 2).reshape(B, N, C))

    def _xformer(self, x, ctx, p):
        sd = self.sd
        B, C, H, W = x.shape
        res = x
        h = self.rn(self._gn(x, p + ".norm", 1e-6, False))
        h = self.rs(self._conv(h, p + ".proj_in"))
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        t = p + ".transformer_blocks.0"
        q, k, v = self._ln_linear(h, t + ".norm1", [t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"], bias=False)
        a = self._sdpa(self.rb(q), self.rb(k), self.rb(v))
        h = self.rs(h + self._linear(a, t + ".attn1.to_out.0"))
        (q,) = self._ln_linear(h, t + ".norm2", [t + ".attn2.to_q"], bias=False)
        cx = r16(ctx)
        k = self.rb(self._linear(cx, t + ".attn2.to_k", bias=False)); v = self.rb(self._linear(cx, t + ".attn2.to_v", bias=False))
        a = self._sdpa(self.rb(q), k, v)
        h = self.rs(h + self._linear(a, t + ".attn2.to_out.0"))
        (pr,) = self._ln_linear(h, t + ".norm3", [t + ".ff.net.0.proj"])
        val, gate = pr.chunk(2, dim=-1)
        ff = self.rb(val * F.gelu(gate))
        h = h + self._linear(ff, t + ".ff.net.2")
        # the hidden after the feed-forward is consumed by proj_out only: it is the GEMM's fp16 operand in every mode
        h = self.rr_po(self.rs(h))
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self._conv(h, p + ".proj_out")
        return self.rs(h + res)

    @torch.no_grad()
    def __call__(self, sample, timestep, encoder_hidden_states):
        cfg, sd = self.cfg, self.sd
        x = r16(sample.float())
        ctx = r16(encoder_hidden_states.float())
        B = x.shape[0]
        t = torch.as_tensor(timestep, dtype=torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(B)
        c0 = cfg["block_out_channels"][0]
        emb = timestep_embedding(t, c0)
        emb = F.linear(emb, sd["time_embedding.linear_1.weight"], sd["time_embedding.linear_1.bias"])
        emb = F.linear(F.silu(emb), sd["time_embedding.linear_2.weight"], sd["time_embedding.linear_2.bias"])
        ts = r16(F.silu(emb))
        h = self.rs(F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1))
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
                h = self.rs(self._conv(self.rr_ud(h), f"{b}.downsamplers.0.conv", stride=2, padding=1))
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
                h = F.interpolate(self.rr_ud(h), scale_factor=2.0, mode="nearest")
                h = self.rs(self._conv(h, f"{b}.upsamplers.0.conv", padding=1))
        h = self._gn(h, "conv_norm_out", 1e-5, True)
        if self.head:
            h = self.rn(h)
        return F.conv2d(h, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def manifest_from_oracle_keys(cfg):
    """the executor's manifest needs the .so; the same names / shapes come from a dry walk of the topology"""
    import ctypes as C
    from consolver_amd import _lib as L
    from consolver_amd.unet import HipUNet2DConditionModel
    u = HipUNet2DConditionModel.__new__(HipUNet2DConditionModel)
    c = L.CsUNetConfig()
    c.in_channels, c.out_channels = cfg["in_channels"], cfg["out_channels"]
    for i in range(4):
        c.block_out_channels[i] = cfg["block_out_channels"][i]
        c.down_has_attn[i] = cfg["down_has_attn"][i]
        c.up_has_attn[i] = cfg["up_has_attn"][i]
    c.layers_per_block, c.num_heads = cfg["layers_per_block"], cfg["num_heads"]
    c.cross_attention_dim, c.norm_num_groups = cfg["cross_attention_dim"], cfg["norm_num_groups"]
    c.sample_size, c.ctx_len = cfg["sample_size"], cfg["ctx_len"]
    h = C.c_void_p()
    L.check(L.lib().cs_unet_create(C.byref(c), C.byref(h)))
    u._h = h
    m = u.manifest()
    return m


if __name__ == "__main__":
    t = int(sys.argv[1]) if len(sys.argv) > 1 else 499
    torch.set_num_threads(os.cpu_count())
    cfg = dict(SD15_CONFIG)
    sd = synthetic_unet_state_dict(manifest_from_oracle_keys(cfg), seed=7)
    g = torch.Generator().manual_seed(5)
    lat = torch.randn(1, 4, 64, 64, generator=g).half().float()
    ctx = synthetic_prompt_embeds(2, seed=13 + t).half().float()
    x2 = torch.cat([lat] * 2)
    t0 = time.time()
    want = UNetOracle(sd, cfg)(x2, t, ctx)
    print(f"oracle forward {time.time() - t0:.1f} s", flush=True)
    base = dict(stream=False, norm=True, branch=True)
    configs = [
        ("fp16 executor as built (every store fp16)", dict(stream=True, raw=True, norm=True, branch=True)),
        ("fp32-class residual stream, every raw GEMM operand fp16 (hi plane only)", dict(base, raw=True)),
        ("  + proj_out reads hi + lo", dict(base, raw=True, raw_po=False)),
        ("  + shortcut 1x1 reads hi + lo", dict(base, raw=True, raw_sc=False)),
        ("  + down / upsample conv reads hi + lo", dict(base, raw=True, raw_ud=False)),
        ("  + proj_out and shortcut read hi + lo", dict(base, raw=True, raw_po=False, raw_sc=False)),
        ("fp32-class residual stream, every raw operand hi + lo", dict(base, raw=False)),
        ("  shortcut hi + lo, LayerNorm folded into the consumer GEMM", dict(base, raw=True, raw_sc=False, ln_fold=True)),
        ("  shortcut + proj_out hi + lo, LayerNorm folded", dict(base, raw=True, raw_sc=False, raw_po=False, ln_fold=True)),
        ("fp16 executor as built + LayerNorm folded", dict(stream=True, raw=True, norm=True, branch=True, ln_fold=True)),
        ("only the residual stream fp16", dict(stream=True, raw=False, norm=False, branch=False)),
        ("only norm outputs fp16", dict(stream=False, raw=False, norm=True, branch=False)),
        ("only branch tensors fp16", dict(stream=False, raw=False, norm=False, branch=True)),
        ("only raw operands fp16", dict(stream=False, raw=True, norm=False, branch=False)),
    ]
    if os.environ.get("SIM_ONLY"):
        keep = [int(k) for k in os.environ["SIM_ONLY"].split(",")]
        configs = [configs[k] for k in keep]
    for name, kw in configs:
        e = rel_l2(Emu(sd, cfg, **kw)(x2, t, ctx), want)
        print(f"  {e:.3e}  {name}", flush=True)
