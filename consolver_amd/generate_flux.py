"""Sharded FLUX-Kontext edit driver: JSONL instructions + reference images -> edited images.

Mirror of ``edit_ppo/generate_ours.py`` (:30-189) on the native engine: the same JSONL entry fields (``key``, ``category``,
``file_name``, ``instruction``), the same output layout ``OUTPUT_DIR/<category>/<key>/{ref_image.jpg, instruction.txt,
edited_image.jpg}``, ceil-sized chunks per GPU (:176-177), one process per GPU, generator seed 0 per entry (:92).

Per entry (edit_ppo/pipeline.py:613-623, 1009-1150): reference image -> VAE encoder (posterior mode, shift, scale) -> packed
image latents; seeded noise -> packed latents; 8-step FMPPOScheduler loop around the DiT; unpack -> VAE decoder -> pixels.
The text side (T5-XXL + CLIP embeddings of the instruction) needs third-party weights and tokenizers: it is read from an
embedding cache (``save_instruction_cache``): ``{key}.prompt_embeds`` [512, 4096] and ``{key}.pooled`` [768] in one safetensors file.
"""
import json
import os
import re
import shutil
from math import ceil

import numpy as np
import torch

from . import launch
from .flux import pack_latents
from .vae import encode_image_latents, flux_decode_latents


def sanitize_folder_name(name):
    """generate_ours.py:28-32: every non-alphanumeric character of the stripped category becomes "_"; empty -> "Unknown"."""
    if not name:
        return "Unknown"
    return re.sub(r"[^a-zA-Z0-9]", "_", name.strip())


def ensure_unique_path(path):
    """generate_ours.py:40-48: append _1, _2, ... when the file exists."""
    if not os.path.exists(path):
        return path
    base, ext = os.path.splitext(path)
    i = 1
    while os.path.exists(f"{base}_{i}{ext}"):
        i += 1
    return f"{base}_{i}{ext}"


def load_jsonl(path):
    """generate_ours.py:152-163: invalid lines are skipped."""
    data = []
    with open(path) as f:
        for line in f:
            try:
                data.append(json.loads(line.strip()))
            except json.JSONDecodeError:
                continue
    return data


def chunk_entries(data, num_gpus):
    """generate_ours.py:176-177."""
    chunk_size = ceil(len(data) / num_gpus) if data else 0
    return [data[i:i + chunk_size] for i in range(0, len(data), chunk_size)] if chunk_size else []


def save_instruction_cache(path, embeds):
    """embeds: {key: (prompt_embeds [T, 4096], pooled [768])}"""
    from safetensors.torch import save_file
    t = {}
    for k, (pe, pooled) in embeds.items():
        t[f"{k}.prompt_embeds"] = pe.detach().to("cpu", torch.bfloat16).contiguous()
        t[f"{k}.pooled"] = pooled.detach().to("cpu", torch.bfloat16).contiguous()
    save_file(t, path, metadata={"format": "consolver_amd.instruction_cache.v1"})


def load_instruction_embeds(cache, key, device):
    return cache.get_tensor(f"{key}.prompt_embeds").to(device)[None], cache.get_tensor(f"{key}.pooled").to(device)[None]


def preprocess_image(path, size):
    """reference image -> [1, 3, size, size] in [-1, 1] (pipeline image_processor: resize + normalise)."""
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size, size), Image.LANCZOS)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return (x * 2 - 1)[None]


def process_instruction(entry, engine, vae, cache, image_dir, output_dir, device, num_inference_steps=8, seed=0):
    """generate_ours.py:50-104 for one entry; returns the edited-image path (or None when the reference image is missing)."""
    from PIL import Image
    key, category, file_name, instruction = entry["key"], entry["category"], entry["file_name"], entry["instruction"]
    src = os.path.join(image_dir, os.path.basename(file_name))
    if not os.path.exists(src):
        return None
    sub = os.path.join(output_dir, sanitize_folder_name(category), key)
    os.makedirs(sub, exist_ok=True)
    shutil.copy(src, ensure_unique_path(os.path.join(sub, "ref_image.jpg")))
    with open(ensure_unique_path(os.path.join(sub, "instruction.txt")), "w") as f:
        f.write(instruction)
    S = vae.config.sample_size                               # latent side; image side = 8 S
    image = preprocess_image(src, 8 * S).to(device, torch.float16)
    image_latents = pack_latents(encode_image_latents(vae, image)).to(torch.bfloat16)
    gen = torch.Generator().manual_seed(seed)                # generator=torch.manual_seed(0), generate_ours.py:92
    noise = torch.randn(1, vae.config.latent_channels, S, S, generator=gen).to(device)
    latents = pack_latents(noise).to(torch.bfloat16)
    pe, pooled = load_instruction_embeds(cache, key, device)
    out = engine.generate(latents, image_latents, pe, pooled, latent_hw=(S // 2, S // 2), num_inference_steps=num_inference_steps)
    img = flux_decode_latents(vae, out.to(torch.float16), height=8 * S, width=8 * S)[0]
    arr = (img.float().clamp(0, 1).permute(1, 2, 0) * 255.0).round().to(torch.uint8).cpu().numpy()
    dst = ensure_unique_path(os.path.join(sub, "edited_image.jpg"))
    Image.fromarray(arr).save(dst)
    return dst


def worker(entries, engine, vae, cache_path, image_dir, output_dir, device, num_inference_steps=8):
    """generate_ours.py:107-148 with the models already built for this process."""
    from safetensors import safe_open
    done = 0
    with safe_open(cache_path, framework="pt", device="cpu") as cache:
        for entry in entries:
            done += process_instruction(entry, engine, vae, cache, image_dir, output_dir, device, num_inference_steps) is not None
    return done


def shard_for_rank(data, world, rank):
    chunks = chunk_entries(data, world)
    return chunks[rank] if rank < len(chunks) else []


__all__ = ["sanitize_folder_name", "ensure_unique_path", "load_jsonl", "chunk_entries", "shard_for_rank", "save_instruction_cache",
           "process_instruction", "worker", "launch"]
