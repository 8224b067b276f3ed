"""CPU restatement (plain torch ops, fp32) of the SD1.5 ``AutoencoderKL`` decoder and ``decode_latents``.

TEST INFRASTRUCTURE ONLY (imported by tests/, smoke() and bench.py's cpu_baseline leg).

PARITY UNPINNED for the decoder network: its arithmetic lives in the un-vendored third-party package
diffusers==0.26.3 (env.yaml:52), absent from /root/reference and from this image, and the reference
holds no tests or golden vectors for it.  This file restates the public SD1.5 VAE decoder
(post_quant_conv 1x1; conv_in; mid block = resnet, single-head attention over GroupNorm'd tokens,
resnet; four up blocks of three resnets with nearest-x2 upsample + conv after the first three;
GroupNorm(32, eps 1e-6) + SiLU; conv_out; block widths 128/256/512/512) with the diffusers
state-dict key names.  ``decode_latents`` (the chunking, the 1/scaling_factor, the
(x / 2 + 0.5).clamp(0, 1)) is the reference's own code and follows utils.py:6-34 exactly.
"""
import torch
import torch.nn.functional as F

SD15_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, sample_size=64, scaling_factor=0.18215, shift_factor=0.0, use_post_quant_conv=True,
                       use_quant_conv=True, with_encoder=False)


def vae_manifest(cfg):
    """[(name, shape)] of the decoder-side tensors, diffusers naming."""
    out = []
    L, top = cfg["latent_channels"], cfg["block_out_channels"][3]

    def res(p, cin, cout):
        out.extend([(p + ".norm1.weight", (cin,)), (p + ".norm1.bias", (cin,)), (p + ".conv1.weight", (cout, cin, 3, 3)),
                    (p + ".conv1.bias", (cout,)), (p + ".norm2.weight", (cout,)), (p + ".norm2.bias", (cout,)),
                    (p + ".conv2.weight", (cout, cout, 3, 3)), (p + ".conv2.bias", (cout,))])
        if cin != cout:
            out.extend([(p + ".conv_shortcut.weight", (cout, cin, 1, 1)), (p + ".conv_shortcut.bias", (cout,))])

    if cfg.get("with_encoder", False):
        c0 = cfg["block_out_channels"][0]
        out.extend([("encoder.conv_in.weight", (c0, cfg["out_channels"], 3, 3)), ("encoder.conv_in.bias", (c0,))])
        prev = c0
        for i in range(4):
            ch = cfg["block_out_channels"][i]
            for j in range(cfg["layers_per_block"]):
                res(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
            if i < 3:
                out.extend([(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (ch, ch, 3, 3)),
                            (f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (ch,))])
            prev = ch
        res("encoder.mid_block.resnets.0", top, top)
        a = "encoder.mid_block.attentions.0"
        out.extend([(a + ".group_norm.weight", (top,)), (a + ".group_norm.bias", (top,))])
        for q in (".to_q", ".to_k", ".to_v", ".to_out.0"):
            out.extend([(a + q + ".weight", (top, top)), (a + q + ".bias", (top,))])
        res("encoder.mid_block.resnets.1", top, top)
        out.extend([("encoder.conv_norm_out.weight", (top,)), ("encoder.conv_norm_out.bias", (top,)),
                    ("encoder.conv_out.weight", (2 * L, top, 3, 3)), ("encoder.conv_out.bias", (2 * L,))])
        if cfg.get("use_quant_conv", True):
            out.extend([("quant_conv.weight", (2 * L, 2 * L, 1, 1)), ("quant_conv.bias", (2 * L,))])
    if cfg.get("use_post_quant_conv", True):
        out.extend([("post_quant_conv.weight", (L, L, 1, 1)), ("post_quant_conv.bias", (L,))])
    out.extend([("decoder.conv_in.weight", (top, L, 3, 3)), ("decoder.conv_in.bias", (top,))])
    res("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    out.extend([(a + ".group_norm.weight", (top,)), (a + ".group_norm.bias", (top,))])
    for q in (".to_q", ".to_k", ".to_v", ".to_out.0"):
        out.extend([(a + q + ".weight", (top, top)), (a + q + ".bias", (top,))])
    res("decoder.mid_block.resnets.1", top, top)
    prev = top
    for i in range(4):
        ch = cfg["block_out_channels"][3 - i]
        for j in range(cfg["layers_per_block"] + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
        if i < 3:
            out.extend([(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (ch, ch, 3, 3)),
                        (f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (ch,))])
        prev = ch
    c0 = cfg["block_out_channels"][0]
    out.extend([("decoder.conv_norm_out.weight", (c0,)), ("decoder.conv_norm_out.bias", (c0,)),
                ("decoder.conv_out.weight", (cfg["out_channels"], c0, 3, 3)), ("decoder.conv_out.bias", (cfg["out_channels"],))])
    return out


class _Cfg:
    def __init__(self, d):
        self.__dict__.update(d)


class VaeOracle:
    """``vae.decode(z, return_dict=False)[0]`` in fp32; ``config.scaling_factor`` like the diffusers model."""

    def __init__(self, sd, config=None, round_weights_to_f16=True):
        self.cfg = dict(SD15_VAE_CONFIG)
        self.cfg.update(config or {})
        self.config = _Cfg(self.cfg)
        self.sd = {k: (v.half().float() if round_weights_to_f16 else v.float()) for k, v in sd.items()}

    def _gn(self, x, p, silu):
        y = F.group_norm(x, self.cfg["norm_num_groups"], self.sd[p + ".weight"], self.sd[p + ".bias"], 1e-6)
        return F.silu(y) if silu else y

    def _resnet(self, x, p):
        sd = self.sd
        h = self._gn(x, p + ".norm1", True)
        h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
        h = self._gn(h, p + ".norm2", True)
        h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
        if (p + ".conv_shortcut.weight") in sd:
            x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
        return x + h

    def _attn(self, x, p):
        sd = self.sd
        B, C, H, W = x.shape
        h = self._gn(x, p + ".group_norm", False).view(B, C, H * W).transpose(1, 2)      # [B, HW, C]
        q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
        k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
        v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
        a = torch.softmax(q @ k.transpose(1, 2) * C ** -0.5, dim=-1) @ v                 # one head of dim C
        a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
        return x + a.transpose(1, 2).reshape(B, C, H, W)

    @torch.no_grad()
    def encode_mode(self, x):
        """mode (= mean) of ``vae.encode(x).latent_dist``: conv_in, four down blocks (two resnets each; pad (0,1,0,1) + 3x3
        stride-2 conv after the first three), mid block, GroupNorm + SiLU, conv_out -> 2L moments [-> quant_conv] -> first L."""
        sd = self.sd
        h = F.conv2d(x.float(), sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
        for i in range(4):
            for j in range(self.cfg["layers_per_block"]):
                h = self._resnet(h, f"encoder.down_blocks.{i}.resnets.{j}")
            if i < 3:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
        h = self._resnet(h, "encoder.mid_block.resnets.0")
        h = self._attn(h, "encoder.mid_block.attentions.0")
        h = self._resnet(h, "encoder.mid_block.resnets.1")
        h = self._gn(h, "encoder.conv_norm_out", True)
        m = F.conv2d(h, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
        if "quant_conv.weight" in sd:
            m = F.conv2d(m, sd["quant_conv.weight"], sd["quant_conv.bias"])
        return m[:, :self.cfg["latent_channels"]]

    @torch.no_grad()
    def decode(self, z, return_dict=False):
        sd = self.sd
        z = z.float()
        if "post_quant_conv.weight" in sd:
            z = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
        x = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
        x = self._resnet(x, "decoder.mid_block.resnets.0")
        x = self._attn(x, "decoder.mid_block.attentions.0")
        x = self._resnet(x, "decoder.mid_block.resnets.1")
        for i in range(4):
            for j in range(self.cfg["layers_per_block"] + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}")
            if i < 3:
                x = F.interpolate(x, scale_factor=2.0, mode="nearest")
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        x = self._gn(x, "decoder.conv_norm_out", True)
        x = F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)
        return (x,)


def decode_latents(vae, latents, batch_size=1):
    """utils.py:6-34: scale by 1/scaling_factor, decode in chunks, map to [0, 1], concatenate."""
    latents = 1 / vae.config.scaling_factor * latents
    images = []
    for s in range(0, latents.shape[0], batch_size):
        img = vae.decode(latents[s:min(s + batch_size, latents.shape[0])], return_dict=False)[0]
        images.append((img / 2 + 0.5).clamp(0, 1))
    return torch.cat(images, dim=0)


def flux_decode_latents(vae, latents_unpacked):
    """edit_ppo/utils.py:22-25 after the unpack: latents / scaling_factor + shift_factor -> decode -> [0, 1]."""
    z = latents_unpacked / vae.config.scaling_factor + vae.config.shift_factor
    return (vae.decode(z, return_dict=False)[0] / 2 + 0.5).clamp(0, 1)


def encode_image_latents(vae, image):
    """edit_ppo/pipeline.py:613-623: (argmax of the posterior - shift_factor) * scaling_factor."""
    return (vae.encode_mode(image) - vae.config.shift_factor) * vae.config.scaling_factor
