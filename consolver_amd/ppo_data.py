"""Teacher-pair data of the PPO trainer: on-disk format and batch collation.

Format written by gen_pretrain/generate_data.py:180-213 and read by data_processing.py:10-63, one sample per id
``{device_id}_{index:08d}``:  ``{id}.txt`` (prompt), ``{id}.png`` (decoded teacher image, not needed by the reward path:
rewards are computed on the decoded teacher LATENT, train_ppo.py:366-373), ``noise_{id}.pth`` and ``latent_{id}.pth``
(``torch.save`` of ``[4, 64, 64]`` tensors: initial noise and the teacher's final latent).
Host-side logic only (file I/O); the tensors go to the GPU in the caller.
"""
import os
import random

import torch


def teacher_pair_id(device_id, index):
    return f"{device_id}_{index:08d}"          # generate_data.py:189-191


def save_teacher_pair(out_dir, sample_id, prompt, noise, latent):
    """generate_data.py:189-213 (without the PNG)."""
    if torch.isnan(latent).any():
        raise ValueError("teacher latent contains NaN")         # generate_data.py:209
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{sample_id}.txt"), "w") as f:
        f.write(prompt)
    torch.save(noise.detach().cpu().clone(), os.path.join(out_dir, f"noise_{sample_id}.pth"))
    torch.save(latent.detach().cpu().clone(), os.path.join(out_dir, f"latent_{sample_id}.pth"))


class TeacherPairDataset(torch.utils.data.Dataset):
    """data_processing.py:10-63: ids are the ``*.txt`` files of the directory; ``__getitem__`` -> (text, noise, latent).
    A sample whose files are missing or whose latent holds NaN is replaced by a random other one, like the reference
    (:40-61); ``strict=True`` raises instead."""

    def __init__(self, img_dir, strict=False):
        self.img_dir = img_dir
        self.strict = strict
        self.ids = sorted(f[:-4] for f in os.listdir(img_dir) if f.endswith(".txt"))

    def __len__(self):
        return len(self.ids)

    def _load(self, idx):
        sid = self.ids[idx].strip()
        with open(os.path.join(self.img_dir, sid + ".txt")) as f:
            text = f.read().strip()
        noise = torch.load(os.path.join(self.img_dir, f"noise_{sid}.pth"), map_location="cpu")
        latent = torch.load(os.path.join(self.img_dir, f"latent_{sid}.pth"), map_location="cpu")
        if torch.isnan(latent).any():
            raise FileNotFoundError(sid)
        return text, noise, latent

    def __getitem__(self, idx):
        for _ in range(1000):
            try:
                return self._load(idx)
            except (OSError, RuntimeError, FileNotFoundError):
                if self.strict:
                    raise
                idx = random.randint(0, len(self.ids) - 1)
        raise RuntimeError("no loadable teacher pair found")


def collate_teacher_pairs(samples):
    text, noise, latent = zip(*samples)
    return list(text), torch.stack(noise), torch.stack(latent)


def repeat_random_sample(batch, index=None, return_index=False):
    """data_processing.py:65-83: one random sample of the batch repeated batch-size times (the trainer rolls out B
    trajectories of the SAME prompt/noise so that the advantage normalisation compares policies, not prompts).
    ``index``: use this item instead of drawing one; ``return_index=True``: also return the item that was used, for callers that
    hold per-item side data (cached prompt embeddings) -- no hidden state is kept between calls."""
    text, noise, tch = batch
    B = noise.shape[0]
    i = random.randint(0, B - 1) if index is None else int(index)
    out = ([text[i]] * B, noise[i:i + 1].repeat(B, *[1] * (noise.dim() - 1)), tch[i:i + 1].repeat(B, *[1] * (tch.dim() - 1)))
    return out + (i,) if return_index else out
