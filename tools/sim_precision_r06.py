"""Round-6 precision sizing on the CPU (tools only), on top of tools/sim_precision.py's emulation of the executor's rounding points:

  (a) the transformer blocks' hidden state h stored as hi (fp16) + lo (8-bit e5m2 = the top byte of the fp16 lo plane: 14 significant bits) instead of hi + lo (fp16, 22 bits);
  (b) the same for EVERY residual-stream tensor;
  (c) each of the 13 resnet shortcut 1x1 convs reading the hi plane only instead of hi + lo (x2_split_a bit 0 per layer): error it adds, next to the k-loop time it costs.

    python tools/sim_precision_r06.py [t]
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from sim_precision import Emu, manifest_from_oracle_keys, r16, rel_l2      # noqa: E402
from oracle.unet_oracle import UNetOracle                                    # noqa: E402
from consolver_amd.synth import synthetic_prompt_embeds, synthetic_unet_state_dict  # noqa: E402
from consolver_amd.unet import SD15_CONFIG                                   # noqa: E402


def hi_lo8(x):
    """value stored as fp16 hi + e5m2 lo (what cs lo8 planes hold)"""
    hi = x.half().float()
    return hi + (x - hi).to(torch.float8_e5m2).float()


class Emu6(Emu):
    """Emu with (1) a separate store function for the transformer hidden state h, (2) a per-layer choice of the shortcut's operand"""

    def __init__(self, sd, cfg, h_store=None, sc_hi_only=(), **kw):
        super().__init__(sd, cfg, **kw)
        self.h_store = h_store
        self.sc_hi_only = set(sc_hi_only)

    def _resnet(self, x, temb_silu, p):
        keep = self.rr_sc
        if p in self.sc_hi_only:
            self.rr_sc = r16
        try:
            return super()._resnet(x, temb_silu, p)
        finally:
            self.rr_sc = keep

    def _xformer(self, x, ctx, p):
        if self.h_store is None:
            return super()._xformer(x, ctx, p)
        # h's stores (proj_in, to_out, cross-attention to_out) go through h_store; the block's output (proj_out + residual) stays a stream store (self.rs)
        sd = self.sd
        B, C, H, W = x.shape
        res = x
        hs = self.h_store
        h = self.rn(self._gn(x, p + ".norm", 1e-6, False))
        h = hs(self._conv(h, p + ".proj_in"))
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        t = p + ".transformer_blocks.0"
        q, k, v = self._ln_linear(h, t + ".norm1", [t + ".attn1.to_q", t + ".attn1.to_k", t + ".attn1.to_v"], bias=False)
        a = self._sdpa(self.rb(q), self.rb(k), self.rb(v))
        h = hs(h + self._linear(a, t + ".attn1.to_out.0"))
        (q,) = self._ln_linear(h, t + ".norm2", [t + ".attn2.to_q"], bias=False)
        cx = r16(ctx)
        k = self.rb(self._linear(cx, t + ".attn2.to_k", bias=False)); v = self.rb(self._linear(cx, t + ".attn2.to_v", bias=False))
        a = self._sdpa(self.rb(q), k, v)
        h = hs(h + self._linear(a, t + ".attn2.to_out.0"))
        (pr,) = self._ln_linear(h, t + ".norm3", [t + ".ff.net.0.proj"])
        val, gate = pr.chunk(2, dim=-1)
        ff = self.rb(val * F.gelu(gate))
        h = h + self._linear(ff, t + ".ff.net.2")
        h = self.rr_po(self.rs(h))
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = self._conv(h, p + ".proj_out")
        return self.rs(h + res)


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
    print(f"t = {t}; oracle forward {time.time() - t0:.1f} s", flush=True)
    built = dict(stream=False, raw=True, norm=True, branch=True, raw_sc=False, ln_fold=True)      # the f16x2 executor as built (round 5)
    shortcuts = [k[:-len(".conv_shortcut.weight")] for k in sd if k.endswith(".conv_shortcut.weight")]
    runs = [("f16x2 executor as built: stream hi + lo (fp16), shortcut reads hi + lo, LayerNorm folded", {}),
            ("  h inside the transformer blocks as hi + lo8 (e5m2 lo: 14 bits)", dict(h_store=hi_lo8)),
            ("  EVERY stream tensor as hi + lo8", dict(h_store=hi_lo8, _all8=True)),
            ("  h inside the transformer blocks as ONE fp16 plane", dict(h_store=r16)),
            ("  no shortcut reads hi + lo (x2_split_a = 0)", dict(sc_hi_only=shortcuts))]
    runs += [(f"  shortcut of {s} reads the hi plane only", dict(sc_hi_only=[s])) for s in shortcuts]
    if os.environ.get("SIM_ONLY"):
        keep = [int(k) for k in os.environ["SIM_ONLY"].split(",")]
        runs = [runs[k] for k in keep]
    for name, kw in runs:
        kw = dict(kw)
        all8 = kw.pop("_all8", False)
        e = Emu6(sd, cfg, **built, **kw)
        if all8:
            e.rs = hi_lo8
        print(f"  {rel_l2(e(x2, t, ctx), want):.4e}  {name}", flush=True)
