"""One-process-per-GPU sharding of batched prompt generation (no data-path collective).

Mirrors the reference drivers' partition rules and naming so that outputs land where the
reference would put them:

* SD  : ``gen_ppo.generate_imgs`` (gen_ppo.py:349-357): contiguous blocks of ``len // P`` prompts,
  the last rank takes the remainder; batches of ``batch_size`` inside a shard; per-batch latent seed
  ``seed + batch_idx`` -- identical on every rank (gen_ppo.py:258-260); files ``{rank}_{idx:08d}``
  (gen_ppo.py:319-330).
* FLUX: ``edit_ppo/generate_ours.main`` (:176-177): ceil-sized chunks.

The reference runs 8 pipelines from 8 Python threads of one process (gen_ppo.py:446, one GIL);
here each GPU has its own process and RCCL is used only for the start barrier, the
max-over-ranks elapsed time and an all-gather of per-rank counts / latent checksums.
"""
import os

import torch


def shard_bounds(n_items, world, rank):
    per = n_items // world
    start = rank * per
    end = n_items if rank == world - 1 else (rank + 1) * per
    return start, end


def shard_bounds_ceil(n_items, world, rank):
    per = (n_items + world - 1) // world
    return min(rank * per, n_items), min((rank + 1) * per, n_items)


def batch_plan(n_items, world, rank, batch_size, seed=0, rule="floor"):
    """Yields (batch_idx, [global indices], latent_seed, [output stems]) for this rank."""
    lo, hi = (shard_bounds if rule == "floor" else shard_bounds_ceil)(n_items, world, rank)
    idx = list(range(lo, hi))
    for b, s in enumerate(range(0, len(idx), batch_size)):
        chunk = idx[s:s + batch_size]
        stems = [f"{rank}_{s + j:08d}" for j in range(len(chunk))]
        yield b, chunk, seed + b, stems


def init_distributed(backend=None):
    """(rank, world, local_rank, dist-or-None).  backend: 'nccl' (= RCCL over xGMI) on GPUs, 'gloo' on CPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local, dist


def reduce_max_seconds(dist, seconds, device="cpu"):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_report(dist, count, checksum, device="cpu"):
    """all-gather (count, checksum) per rank -> list of tuples (8 x 16 B on a full node)."""
    if dist is None:
        return [(int(count), float(checksum))]
    mine = torch.tensor([float(count), float(checksum)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [(int(o[0].item()), float(o[1].item())) for o in out]


def gather_rank_records(dist, values, device="cpu"):
    """all-gather one equal-length float64 vector per rank -> list (indexed by rank) of lists.  bench.py's N > 1 `per_rank` record: rank, local
    device, images, elapsed, shard bounds, latent checksum -- one small RCCL all-gather after the timed region."""
    mine = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist is None:
        return [mine.tolist()]
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [o.tolist() for o in out]


def average_gradients(dist, grads):
    """What DistributedDataParallel does to the policy's gradients in the PPO trainer (train_ppo.py:262-266, 8 ranks x 75 k
    parameters = one ~300 KB all-reduce): sum over ranks / world size, in place on the packed gradient vector."""
    if dist is None:
        return grads
    dist.all_reduce(grads, op=dist.ReduceOp.SUM)
    grads.div_(dist.get_world_size())
    return grads
