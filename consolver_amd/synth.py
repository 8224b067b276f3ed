"""Seeded synthetic weights of the exact SD1.5 shapes (no checkpoints exist offline).

conv / linear weights ~ N(0, 1/fan_in), norm gamma = 1 + small noise, beta small noise, biases small:
activations stay O(1) through the GroupNorms so fp16 neither overflows nor underflows.
"""
import torch


def synthetic_unet_state_dict(manifest, seed=20251226, device="cpu"):
    """``device``: where the values are drawn (the CPU and the GPU generators give different streams for one seed: the tests and their committed numbers use the
    CPU stream, bench.py draws on each rank's own GPU)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in manifest:
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            w = torch.randn(shape, generator=g, device=device) * (1.0 / fan_in) ** 0.5
        elif ("norm" in name) and name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        else:
            w = 0.05 * torch.randn(shape, generator=g, device=device)
        sd[name] = w
    return sd


def synthetic_vae_state_dict(manifest, seed=20251227):
    """Same recipe for the AutoencoderKL decoder; conv_out is scaled so images land inside [-1, 1] with some clipping."""
    return synthetic_unet_state_dict(manifest, seed)


def synthetic_clip_state_dict(manifest, seed=20251228):
    """CLIP text tower: embeddings ~ N(0, 0.02^2)-like scale kept O(1) after the first LayerNorm, linears ~ N(0, 1/fan_in)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in manifest:
        if "embedding" in name:
            w = 0.5 * torch.randn(shape, generator=g)
        elif name.endswith(".weight") and len(shape) == 2:
            w = torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
        elif "layer_norm" in name and name.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            w = 0.05 * torch.randn(shape, generator=g)
        sd[name] = w
    return sd


def synthetic_prompt_embeds(batch, ctx_len=77, dim=768, seed=1001):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, ctx_len, dim, generator=g)
    return torch.nn.functional.layer_norm(x, (dim,))


def synthetic_flux_state_dict(manifest, seed=20251226, device="cpu", dtype=torch.float32):
    """seeded FLUX-DiT weights: linear ~ N(0, 1/fan_in), norm weights ~ 1, modulation linears small so that
    gates stay O(0.1-1) (an untrained adaLN-Zero would gate every block off)."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in manifest:
        if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
            w = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif name.endswith(".weight"):
            w = torch.randn(shape, generator=g, device=device) * (1.0 / shape[1]) ** 0.5
            if ".norm" in name and name.endswith("linear.weight"):
                w = w * 0.5
        else:
            w = 0.05 * torch.randn(shape, generator=g, device=device)
            if ".norm" in name and name.endswith("linear.bias"):
                w = w + 0.3
        sd[name] = w.to(dtype)
    return sd
