"""Evaluation harness: image writers of the batch generator and the paired-directory scorer.

Mirrors, by name and output schema:
* gen_ppo.py:205-212,318-325 -- ``{device_id}_{index:08d}.png`` + ``.txt`` (prompt) per generated image;
* compute_reward.py:52-78 ``find_image_pairs``, :81-95 ``load_image_tensor``, :332-365 ``calculate_statistics``,
  :447-462 the results JSON (``statistics`` / ``raw_scores`` / ``config``).
Only the arithmetic-only reward (``image_psnr``, edit_ppo/reward_model.py:484-509) is scored, on the GPU (cs_image_psnr);
the backbone rewards are third-party networks (out of scope).  PNG encode/decode is PIL (host I/O).
"""
import json
import os
from pathlib import Path

import numpy as np
import torch

from . import ppo


def generation_paths(out_dir, device_id, index):
    stem = os.path.join(out_dir, f"{device_id}_{index:08d}")       # gen_ppo.py:321-323
    return stem + ".png", stem + ".txt"


def tensor_to_uint8_hwc(image):
    """[3,H,W] in [0,1] (decode_latents output) -> uint8 [H,W,3], rounding like ``pipe.image_processor.postprocess``
    / ``numpy_to_pil``: (x * 255).round()."""
    x = image.detach().float().clamp(0, 1).cpu()
    return (x.permute(1, 2, 0) * 255.0).round().to(torch.uint8).numpy()


def save_generation(out_dir, device_id, index, image, prompt):
    """one generated image + its prompt, the batch generator's naming (gen_ppo.py:318-325)."""
    from PIL import Image
    os.makedirs(out_dir, exist_ok=True)
    png, txt = generation_paths(out_dir, device_id, index)
    Image.fromarray(tensor_to_uint8_hwc(image)).save(png)
    with open(txt, "w") as f:
        f.write(prompt)
    return png, txt


def find_image_pairs(root_dir1, root_dir2, *args):
    """compute_reward.py:52-78: PNG files with the same relative path under both roots."""
    root1, root2 = Path(root_dir1), Path(root_dir2)
    pairs = []
    for p in sorted(root1.rglob("*.png")):
        q = root2 / p.relative_to(root1)
        if q.exists():
            pairs.append((str(p), str(q)))
    return pairs


def load_image_tensor(image_path, device):
    """compute_reward.py:81-95: RGB, [3,H,W] float32 in [0,1] (ToTensor = uint8 / 255)."""
    from PIL import Image
    arr = np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8)
    return (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255.0).to(device)


def calculate_statistics(results):
    """compute_reward.py:332-365 (population std, like np.std)."""
    stats = {}
    for reward_type, scores in results.items():
        if len(scores) > 0:
            a = np.array(scores)
            stats[reward_type] = {"mean": float(np.mean(a)), "std": float(np.std(a)), "min": float(np.min(a)), "max": float(np.max(a)),
                                  "median": float(np.median(a)), "count": len(scores)}
        else:
            stats[reward_type] = {"mean": 0.0, "std": 0.0, "min": 0.0, "max": 0.0, "median": 0.0, "count": 0}
    return stats


def score_image_pairs(image_pairs, reward_types=("image_psnr",), batch_size=16, device="cuda:0"):
    """-> {reward_type: [score per pair]} (compute_reward.py:98-330 reduced to the arithmetic-only reward)."""
    results = {}
    for rt in reward_types:
        scores = []
        for s in range(0, len(image_pairs), batch_size):
            chunk = image_pairs[s:s + batch_size]
            a = torch.stack([load_image_tensor(p, device) for p, _ in chunk])
            b = torch.stack([load_image_tensor(q, device) for _, q in chunk])
            scores.extend(float(v) for v in ppo.calculate_reward(rt, None, None, a, b, device).flatten().cpu())
        results[rt] = scores
    return results


def write_results(output_path, results, config):
    """compute_reward.py:447-462."""
    data = {"statistics": calculate_statistics(results), "raw_scores": results, "config": config}
    with open(output_path, "w", encoding="utf-8") as f:
        json.dump(data, f, indent=2, ensure_ascii=False)
    return data
